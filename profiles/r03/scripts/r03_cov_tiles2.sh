#!/bin/bash
# k_coverage: smaller tiles, more filter bits (prebuilt variant libraries, selected with TALC_LIB)
O=gpurun_out
for v in t256_64 t512_64; do
  TALC_LIB=talc_amd/_build/libtalc_hip_$v.so python3 tools/cov_bench.py 2>> $O/cov_tiles.err | tee -a $O/cov_tiles2.txt
done
TALC_FILTER_BITS=28 TALC_LIB=talc_amd/_build/libtalc_hip_t512_64.so python3 tools/cov_bench.py 2>> $O/cov_tiles.err | tee -a $O/cov_tiles2.txt
TALC_FILTER_BITS=14 TALC_LIB=talc_amd/_build/libtalc_hip_t512_64.so python3 tools/cov_bench.py 2>> $O/cov_tiles.err | tee -a $O/cov_tiles2.txt
