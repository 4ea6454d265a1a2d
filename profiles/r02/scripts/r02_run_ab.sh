#!/bin/bash
set -o pipefail
O=gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "walk or default or variants" > $O/ab_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -2 $O/ab_pytest.log
[ $rc -eq 0 ] || exit 1
for r in 1 2 3; do
for v in base X; do
L=$PWD/talc_amd/_build/libtalc_hip_$v.so; [ $v = X ] && L=$PWD/talc_amd/_build/libtalc_hip.so
TALC_LIB=$L timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu --no-h2h --no-paralog > $O/ab.json 2> $O/ab.err || exit 1
python -c "import json; d=json.load(open('$O/ab.json')); print('[$v]', d['ms_per_step'], d['kernels_ms']['search_ms'])"
done; done
for v in base X; do
L=$PWD/talc_amd/_build/libtalc_hip_$v.so; [ $v = X ] && L=$PWD/talc_amd/_build/libtalc_hip.so
TALC_WALK=0 TALC_LIB=$L timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu --no-h2h --no-paralog > $O/ab.json 2> $O/ab.err || exit 1
python -c "import json; d=json.load(open('$O/ab.json')); print('[$v walk off]', d['ms_per_step'], d['kernels_ms']['search_ms'])"
done
