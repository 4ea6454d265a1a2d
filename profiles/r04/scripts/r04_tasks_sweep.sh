#!/bin/bash
# the edge tasks' thresholds and the queue key's gap scale on config 2 and the branching workload (same box)
O=gpurun_out
mkdir -p $O
: > $O/r04_tasks_sweep.txt
run() { echo "== $*" >> $O/r04_tasks_sweep.txt; env "$@" timeout -k 10 300 python3 tools/search_bench.py --reps 4 >> $O/r04_tasks_sweep.txt 2>&1 || exit 1; }
run TALC_LIB=talc_amd/_build/libtalc_hip_base.so
run TALC_NO_EDGE_TASKS=1 TALC_ORDER_GAP_SCALE=256
run TALC_EDGE_TASK_MIN=150
run TALC_EDGE_TASK_MIN=300
run TALC_EDGE_TASK_MIN=450
run TALC_EDGE_TASK_MIN=450 TALC_EDGE_TASK_HEAVY=350
run TALC_EDGE_TASK_MIN=150 TALC_ORDER_GAP_SCALE=256
run TALC_EDGE_TASK_MIN=150 TALC_ORDER_GAP_SCALE=512
run TALC_EDGE_TASK_MIN=150 TALC_ORDER_GAP_SCALE=1024
run TALC_EDGE_TASK_MIN=150 TALC_ORDER_GAP_SCALE=2304
grep -v "^+" $O/r04_tasks_sweep.txt
