#!/bin/bash
# the heavy reads of stress set 105 (K = 31, mixed lengths, 40 % paralogs): category profile and per-read table
O=gpurun_out
mkdir -p $O
TALC_LIB=talc_amd/_build/libtalc_hip_prof.so TALC_PROF_PRINT=1 TALC_PROF_READS=$O/reads_set105.tsv timeout -k 10 400 python3 tools/heavy_reads.py 105 6000 > $O/r04_heavy_105.txt 2>&1 || { tail -20 $O/r04_heavy_105.txt; exit 1; }
gzip -f $O/reads_set105.tsv
grep -E "case|prof" $O/r04_heavy_105.txt | tail -75
