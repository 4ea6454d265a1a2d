#!/bin/bash
# first runs of the edge tasks: the new parity test, the parity suite, then same-box A/B of config 2 and the branching workload
O=gpurun_out
mkdir -p $O
timeout -k 10 300 python3 -m pytest tests/test_gpu_stress.py -x -q -m gpu -k "edge_anchors" > $O/r04_tasks_test.txt 2>&1 || { tail -30 $O/r04_tasks_test.txt; exit 1; }
tail -3 $O/r04_tasks_test.txt
timeout -k 10 400 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/r04_tasks_parity.txt 2>&1 || { tail -30 $O/r04_tasks_parity.txt; exit 1; }
tail -3 $O/r04_tasks_parity.txt
timeout -k 10 300 python3 tools/search_bench.py --reps 3 > $O/r04_tasks_ab.txt 2>&1 &&
TALC_NO_EDGE_TASKS=1 timeout -k 10 300 python3 tools/search_bench.py --reps 3 >> $O/r04_tasks_ab.txt 2>&1
cat $O/r04_tasks_ab.txt
