#!/bin/bash
set -o pipefail
O=gpurun_out
python bench.py --steps 5 --warmup 2 --no-cpu --no-h2h --no-paralog > $O/r02l_bench_c2.json 2> $O/r02l_bench_c2.err || exit 1
grep -h "warmup 1" $O/r02l_bench_c2.err
TALC_LIB=$PWD/talc_amd/_build/libtalc_hip_prof.so TALC_PROF_PRINT=1 python bench.py --steps 1 --warmup 1 --no-cpu --no-h2h --no-paralog > $O/r02l_prof.json 2> $O/r02l_prof.err || exit 1
grep "prof\]" $O/r02l_prof.err | tail -9
python bench.py --config 5 --steps 2 --warmup 1 --no-cpu --no-h2h --no-paralog > $O/r02l_bench_c5.json 2> $O/r02l_bench_c5.err || exit 1
grep -h "warmup 0" $O/r02l_bench_c5.err
