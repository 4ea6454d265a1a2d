#!/bin/bash
set -o pipefail
O=gpurun_out
for i in 1 2; do
TALC_LIB=$PWD/talc_amd/_build/libtalc_hip_covexp$i.so timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu --no-h2h --no-paralog > $O/r02u_exp$i.json 2> $O/r02u_exp$i.err || { tail -5 $O/r02u_exp$i.err; exit 1; }
python -c "import json; d=json.load(open('$O/r02u_exp$i.json')); print('exp$i', d['kernels_ms'])"
done
