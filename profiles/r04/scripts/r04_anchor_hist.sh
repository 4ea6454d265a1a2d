#!/bin/bash
# the branching workload's edge anchors (profile build): histogram of their durations, per-read count and longest
O=gpurun_out
L=talc_amd/_build/libtalc_hip_prof.so
TALC_NO_EDGE_TASKS=1 TALC_PROF_PRINT=1 TALC_PROF_READS=$O/reads_paralog_anch.tsv TALC_LIB=$L timeout -k 10 300 python3 tools/search_bench.py --reps 1 --no-main > $O/r04_tasks_prof_paralog.txt 2>&1
gzip -f $O/reads_paralog_anch.tsv
tail -80 $O/r04_tasks_prof_paralog.txt | grep -E "edge anchors|ticks anchors|lib="
