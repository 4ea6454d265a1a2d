#!/bin/bash
set -o pipefail
O=gpurun_out
TALC_LIB=$PWD/talc_amd/_build/libtalc_hip_prof.so TALC_PROF_PRINT=1 TALC_PROF_SLOW=1 timeout -k 10 500 python bench.py --config 5 --steps 1 --warmup 1 --no-cpu --no-h2h --no-paralog > $O/c5prof.json 2> $O/c5prof.err || { tail -5 $O/c5prof.err; exit 1; }
grep "prof\]" $O/c5prof.err | tail -41
grep "slow\]" $O/c5prof.err | tail -26
python -c "import json; d=json.load(open('$O/c5prof.json')); print(d['value'], d['ms_per_step'], d['kernels_ms'])"
