#!/bin/bash
# k_coverage: tile size x workgroup size (prebuilt variant libraries, selected with TALC_LIB)
O=gpurun_out
for v in t512_64 t1024_128 t1024_64 t2048_128 t4096_256; do
  TALC_LIB=talc_amd/_build/libtalc_hip_$v.so python3 tools/cov_bench.py --full 2>> $O/cov_tiles.err | tee -a $O/cov_tiles.txt
done
python3 tools/cov_bench.py --full 2>> $O/cov_tiles.err | tee -a $O/cov_tiles.txt
