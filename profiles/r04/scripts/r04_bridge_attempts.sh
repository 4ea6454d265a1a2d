#!/bin/bash
# per read: start anchors of bridge searches walked and the longest walk (profile build) — the branching workload and stress set 105
O=gpurun_out
mkdir -p $O
L=talc_amd/_build/libtalc_hip_prof.so
TALC_PROF_READS=$O/reads_paralog_br.tsv TALC_LIB=$L timeout -k 10 300 python3 tools/search_bench.py --reps 1 --no-main > $O/r04_br.txt 2>&1 || { tail -5 $O/r04_br.txt; exit 1; }
TALC_LIB=$L TALC_PROF_READS=$O/reads_set105_br.tsv timeout -k 10 400 python3 tools/heavy_reads.py 105 6000 >> $O/r04_br.txt 2>&1 || { tail -5 $O/r04_br.txt; exit 1; }
gzip -f $O/reads_paralog_br.tsv $O/reads_set105_br.tsv
tail -3 $O/r04_br.txt
