#!/bin/bash
# closing stress run on the final build: every set as shipped (3000 reads), every set with the edge tasks forced on and every edge published (2000 reads)
O=gpurun_out
S=profiles/r04/scripts/r04_stress.sh
: > $O/r04_stress.log
STRESS_READS=3000 bash $S || exit 1
TALC_EDGE_TASKS=1 TALC_EDGE_TASK_MIN=0 TALC_EDGE_TASK_HEAVY=0 STRESS_READS=2000 bash $S || exit 1
grep -c "mismatches 0" $O/r04_stress.log; grep "TOTAL" $O/r04_stress.log | sort | uniq -c
