#!/bin/bash
# same-box A/B of the shipped defaults: the build before the edge tasks (libtalc_hip_base.so) against this one, on config 2, the
# branching workload and a config-5-like workload (K = 31, mixed lengths)
O=gpurun_out
mkdir -p $O
: > $O/r04_tasks_ab.txt
for rep in 1 2; do
TALC_LIB=talc_amd/_build/libtalc_hip_base.so timeout -k 10 300 python3 tools/search_bench.py --reps 7 >> $O/r04_tasks_ab.txt 2>&1 &&
timeout -k 10 300 python3 tools/search_bench.py --reps 7 >> $O/r04_tasks_ab.txt 2>&1 || exit 1
done
TALC_LIB=talc_amd/_build/libtalc_hip_base.so timeout -k 10 400 python3 tools/search_bench.py --reps 3 --no-paralog --k 31 --mixed --kmers 100000000 >> $O/r04_tasks_ab.txt 2>&1 &&
timeout -k 10 400 python3 tools/search_bench.py --reps 3 --no-paralog --k 31 --mixed --kmers 100000000 >> $O/r04_tasks_ab.txt 2>&1
grep -v "^+" $O/r04_tasks_ab.txt
