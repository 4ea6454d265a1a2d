#!/bin/bash
# k_search with 3 / 4 / 6 waves per SIMD against the default 5 (how much does the search depend on occupancy?)
set -o pipefail
O=gpurun_out
for w in 3 4 6; do
TALC_LIB=$PWD/talc_amd/_build/libtalc_hip_w$w.so timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu --no-h2h --no-paralog > $O/occ_w$w.json 2> $O/occ_w$w.err || { tail -5 $O/occ_w$w.err; exit 1; }
python -c "import json; d=json.load(open('$O/occ_w$w.json')); print('waves/SIMD $w', d['kernels_ms']['search_ms'])"
done
