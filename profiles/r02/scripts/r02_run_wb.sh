#!/bin/bash
set -o pipefail
O=gpurun_out
for v in 300 450 550; do
TALC_LIB=$PWD/talc_amd/_build/libtalc_hip_wb$v.so timeout -k 10 400 python bench.py --config 5 --steps 2 --warmup 1 --no-cpu --no-h2h --no-paralog > $O/wb.json 2> $O/wb.err || { tail -3 $O/wb.err; exit 1; }
python -c "import json; d=json.load(open('$O/wb.json')); print('min path $v', d['kernels_ms']['search_ms'])"
done
