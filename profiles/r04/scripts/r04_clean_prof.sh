#!/bin/bash
# the half-corrected reads of set 101 (STRESS_CLEAN) through the profile build: where the time of nearly clean reads over a
# branching graph goes (first pass and retry stage print one profile each)
O=gpurun_out
mkdir -p $O
STRESS_CLEAN=0.5 STRESS_READS=${1:-100} TALC_PROF_PRINT=1 TALC_LIB=talc_amd/_build/libtalc_hip_prof.so timeout -k 10 900 python3 tools/stress_branching.py 101 > $O/r04_clean_prof.txt 2>&1
grep -v "#reads\|maxread" $O/r04_clean_prof.txt | grep -E "total|xdrop|x\.|stepb|stepe|srch|scorebr|cycle|child|probe|garden|edgemisc|extnw|utilisation|mismatch|evalfull|#xdrop|anchors" 
