#!/bin/bash
# the branching workload with the edge tasks on (the -DTALC_PROF build): category profile and per-read table (start, duration)
O=gpurun_out
mkdir -p $O
L=talc_amd/_build/libtalc_hip_prof.so
for sc in 256 768; do
TALC_EDGE_TASK_ROUNDS=65535 TALC_ORDER_GAP_SCALE=$sc TALC_PROF_PRINT=1 TALC_PROF_READS=$O/reads_paralog_tasks_$sc.tsv TALC_LIB=$L timeout -k 10 300 python3 tools/search_bench.py --reps 1 --no-main > $O/r04_tasks_prof_paralog_$sc.txt 2>&1 || { tail -5 $O/r04_tasks_prof_paralog_$sc.txt; exit 1; }
gzip -f $O/reads_paralog_tasks_$sc.tsv
tail -90 $O/r04_tasks_prof_paralog_$sc.txt | grep -E "edges|anchors by|t\.|utilisation|dry|lib=|srch|total"
done
