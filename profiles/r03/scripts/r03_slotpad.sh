#!/bin/bash
# does the stride between the waves' scratch slots matter? (TALC_SLOT_PAD adds bytes to it)
O=gpurun_out
for pad in 0 256 512 1024 2048 4096 8192 12288 65536 1048576; do
  echo -n "pad=$pad " | tee -a $O/slotpad.txt
  TALC_SLOT_PAD=$pad python3 tools/cov_bench.py --full --reps 1 2>> $O/slotpad.err | tee -a $O/slotpad.txt
done
