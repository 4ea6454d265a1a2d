#!/bin/bash
# category profile (-DTALC_PROF builds) of the libraries named, config 2
O=gpurun_out
for v in "$@"; do
  echo "== $v" | tee -a $O/prof_ab.txt
  TALC_PROF_PRINT=1 TALC_LIB=talc_amd/_build/$v python3 tools/cov_bench.py --full --reps 1 2>&1 | grep -v "^\[prof\] #reads\|maxread" | tail -45 | tee -a $O/prof_ab.txt
done
