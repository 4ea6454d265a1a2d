#!/bin/bash
set -o pipefail
O=gpurun_out
for km in 1000000 8000000; do python bench.py --kmers $km --steps 3 --warmup 1 --no-cpu --no-h2h > $O/r02c_bench_k$km.json 2> $O/r02c_bench_k$km.err || exit 1; done
grep -h "warmup 0\|setup" $O/r02c_bench_k*.err
