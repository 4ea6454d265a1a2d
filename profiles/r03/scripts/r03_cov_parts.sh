#!/bin/bash
# k_coverage by parts: timing builds without table traffic (1), without filter traffic (2), without either (3)
set -o pipefail
O=gpurun_out
for v in 0 1 2 3; do
  python3 -c "
from talc_amd import build as B
B.build_hip(False, ('-DTALC_COV_EXP=$v',), 'libtalc_hip_covexp$v.so')" > /dev/null 2>&1 || { echo "build $v failed"; exit 1; }
  TALC_LIB=talc_amd/_build/libtalc_hip_covexp$v.so python3 tools/cov_bench.py 2>> $O/cov_parts.err | tee -a $O/cov_parts.txt
done
