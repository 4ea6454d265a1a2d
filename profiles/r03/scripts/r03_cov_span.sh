#!/bin/bash
# k_coverage: minimizer length M = K - span (longer runs of one filter line per wave vs skewed blocks)
O=gpurun_out
for v in m7 m8 m9; do
  TALC_LIB=talc_amd/_build/libtalc_hip_$v.so python3 tools/cov_bench.py 2>> $O/cov_span.err | tee -a $O/cov_span.txt
done
TALC_FILTER_BITS=28 TALC_LIB=talc_amd/_build/libtalc_hip_m9.so python3 tools/cov_bench.py 2>> $O/cov_span.err | tee -a $O/cov_span.txt
