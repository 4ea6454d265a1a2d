#!/bin/bash
set -o pipefail
O=gpurun_out
TALC_LIB=$PWD/talc_amd/_build/libtalc_hip_prof.so TALC_PROF_PRINT=1 timeout -k 10 200 python bench.py --steps 1 --warmup 1 --no-cpu --no-h2h --no-paralog > $O/r02w_prof.json 2> $O/r02w_prof.err || exit 1
grep "prof\]" $O/r02w_prof.err | tail -44
