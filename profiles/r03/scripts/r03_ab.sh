#!/bin/bash
# same-box A/B of library builds: tools/cov_bench.py --full (coverage / structure / search ms on config 2) for every
# library named, twice, alternating
O=gpurun_out
for rep in 1 2; do
  for v in "$@"; do
    TALC_LIB=talc_amd/_build/$v python3 tools/cov_bench.py --full --reps 3 2>> $O/ab.err | tee -a $O/ab.txt
  done
done
