#!/bin/bash
# the -m gpu suite with the edge tasks forced on and every edge published (the suite's own fixtures through the task path), then as shipped
O=gpurun_out
mkdir -p $O
TALC_EDGE_TASKS=1 TALC_EDGE_TASK_MIN=0 TALC_EDGE_TASK_HEAVY=0 timeout -k 10 700 python3 -m pytest tests -x -q -m gpu > $O/r04_suite_tasks_forced.txt 2>&1 || { tail -30 $O/r04_suite_tasks_forced.txt; exit 1; }
tail -3 $O/r04_suite_tasks_forced.txt
