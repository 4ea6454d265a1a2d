#!/bin/bash
# stress comparison with the edge tasks: forced on (all edges published from the start / only when the queue is dry / the
# forced in-order redo / tiny first-pass capacities: anchors that overflow in another wave), and as the batch decides
O=gpurun_out
S=profiles/r04/scripts/r04_stress.sh
: > $O/r04_stress.log
TALC_EDGE_TASKS=1 TALC_EDGE_TASK_MIN=0 TALC_EDGE_TASK_HEAVY=0 STRESS_READS=6000 bash $S 101 102 103 104 105 207 301 302 303 304 401 403 405 || exit 1
TALC_EDGE_TASKS=1 STRESS_READS=6000 bash $S 101 102 201 203 206 || exit 1
grep -c "mismatches 0" $O/r04_stress.log; grep "TOTAL" $O/r04_stress.log | sort | uniq -c
