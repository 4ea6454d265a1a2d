#!/bin/bash
set -o pipefail
O=gpurun_out
for r in 1 2; do
for v in base c1 X; do
L=$PWD/talc_amd/_build/libtalc_hip_$v.so; [ $v = X ] && L=$PWD/talc_amd/_build/libtalc_hip.so
TALC_LIB=$L timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu --no-h2h > $O/ab.json 2> $O/ab.err || exit 1
python -c "import json; d=json.load(open('$O/ab.json')); print('[$v]', d['ms_per_step'], d['kernels_ms']['search_ms'], d['paralog_workload']['ms_per_step'])"
done; done
