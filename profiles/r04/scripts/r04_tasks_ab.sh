#!/bin/bash
# same-box A/B: the committed build (libtalc_hip_base.so) / this build with the edge tasks / without the first-round rule / without the tasks
O=gpurun_out
mkdir -p $O
: > $O/r04_tasks_ab.txt
for rep in 1 2; do
TALC_LIB=talc_amd/_build/libtalc_hip_base.so timeout -k 10 300 python3 tools/search_bench.py --reps 3 >> $O/r04_tasks_ab.txt 2>&1 &&
timeout -k 10 300 python3 tools/search_bench.py --reps 3 >> $O/r04_tasks_ab.txt 2>&1 &&
TALC_EDGE_TASK_HEAVY=4095 timeout -k 10 300 python3 tools/search_bench.py --reps 3 >> $O/r04_tasks_ab.txt 2>&1 &&
TALC_NO_EDGE_TASKS=1 timeout -k 10 300 python3 tools/search_bench.py --reps 3 >> $O/r04_tasks_ab.txt 2>&1 || exit 1
done
grep -v "^+" $O/r04_tasks_ab.txt
