#!/bin/bash
# same-box A/B of library builds (names under talc_amd/_build/): tools/search_bench.py for every library named, twice,
# alternating.  AB_ARGS passes options to the tool.
O=gpurun_out
mkdir -p $O
for rep in 1 2; do
  for v in "$@"; do
    TALC_LIB=talc_amd/_build/$v timeout -k 10 300 python3 tools/search_bench.py $AB_ARGS 2>> $O/r04_ab.err | tee -a $O/r04_ab.txt || exit 1
  done
done
