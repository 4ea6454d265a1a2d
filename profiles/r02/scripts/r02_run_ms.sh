#!/bin/bash
set -o pipefail
O=gpurun_out
for v in 5 8 9; do
TALC_LIB=$PWD/talc_amd/_build/libtalc_hip_ms$v.so timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu --no-paralog --no-h2h > $O/ms.json 2> $O/ms.err || { tail -3 $O/ms.err; exit 1; }
python -c "import json; d=json.load(open('$O/ms.json')); print('span $v', d['kernels_ms']['coverage_ms'])"
done
