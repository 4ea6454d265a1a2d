#!/bin/bash
# category profile (-DTALC_PROF build) of config 2 and of the branching workload: tools/search_bench.py with TALC_PROF_PRINT=1
# (the last launch's profile is the one to read: every correct() call prints one)
O=gpurun_out
mkdir -p $O
L=${1:-libtalc_hip_prof.so}
TALC_PROF_PRINT=1 TALC_LIB=talc_amd/_build/$L timeout -k 10 300 python3 tools/search_bench.py --reps 1 --no-paralog > $O/r04_prof_config2.txt 2>&1 || { tail -5 $O/r04_prof_config2.txt; exit 1; }
TALC_PROF_PRINT=1 TALC_LIB=talc_amd/_build/$L timeout -k 10 300 python3 tools/search_bench.py --reps 1 --no-main > $O/r04_prof_paralog.txt 2>&1 || { tail -5 $O/r04_prof_paralog.txt; exit 1; }
tail -62 $O/r04_prof_config2.txt
