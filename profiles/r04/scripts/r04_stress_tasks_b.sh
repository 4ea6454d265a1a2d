#!/bin/bash
# stress comparison with the edge tasks: forced on (all edges published from the start / only when the queue is dry / the
# forced in-order redo / tiny first-pass capacities: anchors that overflow in another wave), and as the batch decides
O=gpurun_out
S=profiles/r04/scripts/r04_stress.sh
TALC_EDGE_TASKS=1 TALC_EDGE_TASK_MIN=0 TALC_EDGE_TASK_HEAVY=4095 TALC_SEARCH_SLOTS=900 TALC_EDGE_LINGER_MOD=2 STRESS_READS=4000 bash $S 102 103 304 || exit 1
TALC_EDGE_TASKS=1 TALC_EDGE_TASK_MIN=0 TALC_EDGE_TASK_HEAVY=0 TALC_TEST_EDGE_REDO=1 STRESS_READS=3000 bash $S 101 104 302 || exit 1
TALC_EDGE_TASKS=1 TALC_EDGE_TASK_MIN=0 TALC_EDGE_TASK_HEAVY=0 TALC_TEST_TINY_CAPS=1 STRESS_READS=2500 bash $S 101 103 304 || exit 1
TALC_EDGE_TASKS=1 TALC_EDGE_TASK_MIN=0 TALC_EDGE_TASK_HEAVY=0 STRESS_READS=3000 STRESS_N=0.003 bash $S 101 303 || exit 1
TALC_EDGE_TASKS=1 TALC_EDGE_TASK_MIN=0 TALC_EDGE_TASK_HEAVY=0 STRESS_READS=300 STRESS_CLEAN=0.5 bash $S 101 || exit 1
STRESS_READS=6000 bash $S 101 102 103 201 301 || exit 1
grep -c "mismatches 0" $O/r04_stress.log; grep "TOTAL" $O/r04_stress.log | sort | uniq -c
