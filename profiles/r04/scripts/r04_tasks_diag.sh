#!/bin/bash
# edge tasks: for how many rounds of the queue a long border is published from the start of its search; the key's gap scale
O=gpurun_out
mkdir -p $O
: > $O/r04_tasks_diag.txt
run() { echo "== $*" >> $O/r04_tasks_diag.txt; env "$@" timeout -k 10 300 python3 tools/search_bench.py --reps 7 >> $O/r04_tasks_diag.txt 2>&1 || exit 1; }
run TALC_NO_EDGE_TASKS=1 TALC_ORDER_GAP_SCALE=256
run TALC_EDGE_TASK_ROUNDS=1
run TALC_EDGE_TASK_ROUNDS=65535
run TALC_EDGE_TASK_ROUNDS=65535 TALC_EDGE_TASK_HEAVY=300
run TALC_EDGE_TASK_ROUNDS=65535 TALC_ORDER_GAP_SCALE=256
run TALC_EDGE_TASK_ROUNDS=65535 TALC_ORDER_GAP_SCALE=512
grep -v "^+" $O/r04_tasks_diag.txt
