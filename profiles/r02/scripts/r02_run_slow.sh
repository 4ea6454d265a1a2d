#!/bin/bash
set -o pipefail
O=gpurun_out
TALC_LIB=$PWD/talc_amd/_build/libtalc_hip_prof.so TALC_PROF_PRINT=1 TALC_PROF_SLOW=1 timeout -k 10 200 python bench.py --steps 1 --warmup 0 --no-cpu --no-h2h --no-paralog > $O/slow.json 2> $O/slow.err || exit 1
grep "slow\]" $O/slow.err | awk '{print $0, "end", $9+$12}' | sort -t"|" -k1 | sort -k17 -g -r | head -40
grep "slow\]" $O/slow.err | wc -l
grep "utilis" $O/slow.err
