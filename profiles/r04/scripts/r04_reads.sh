#!/bin/bash
# per-read table (cost estimate, its inputs, the search's duration) from the -DTALC_PROF build: config 2, the branching
# workload and a config-5-like one (K = 31, mixed lengths, 100 M k-mers) — input for replaying the work queue offline
O=gpurun_out
mkdir -p $O
L=talc_amd/_build/libtalc_hip_prof.so
TALC_PROF_READS=$O/reads_config2.tsv TALC_LIB=$L timeout -k 10 300 python3 tools/search_bench.py --reps 1 --no-paralog > $O/r04_reads.log 2>&1 &&
TALC_PROF_READS=$O/reads_paralog.tsv TALC_LIB=$L timeout -k 10 300 python3 tools/search_bench.py --reps 1 --no-main >> $O/r04_reads.log 2>&1 &&
TALC_PROF_READS=$O/reads_mixed31.tsv TALC_LIB=$L timeout -k 10 400 python3 tools/search_bench.py --reps 1 --no-paralog --k 31 --mixed --kmers 100000000 >> $O/r04_reads.log 2>&1
tail -5 $O/r04_reads.log
for f in config2 paralog mixed31; do gzip -f $O/reads_$f.tsv; done
ls -la $O/reads_*
