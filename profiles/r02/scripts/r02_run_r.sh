#!/bin/bash
set -o pipefail
O=gpurun_out
for w in 4 6; do TALC_LIB=$PWD/talc_amd/_build/libtalc_hip_w$w.so python bench.py --steps 6 --warmup 2 --no-cpu --no-h2h --no-paralog > $O/r02r_bench_c2_w$w.json 2> $O/r02r_bench_c2_w$w.err || exit 1; echo "w$w"; grep -h "warmup 1" $O/r02r_bench_c2_w$w.err; done
python bench.py --steps 6 --warmup 2 --no-cpu --no-h2h --no-paralog > $O/r02r_bench_c2.json 2> $O/r02r_bench_c2.err || exit 1
grep -h "warmup 1" $O/r02r_bench_c2.err
