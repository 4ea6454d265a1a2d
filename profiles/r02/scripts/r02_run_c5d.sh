#!/bin/bash
set -o pipefail
O=gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -x -q > $O/c5d_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -2 $O/c5d_pytest.log
[ $rc -eq 0 ] || exit 1
for i in 1 2; do
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu --no-paralog --no-h2h > $O/c5d_c2.json 2> $O/c5d_c2.err || exit 1
python -c "import json; d=json.load(open('$O/c5d_c2.json')); print(d['value'], d['ms_per_step'], d['kernels_ms'])"
done
timeout -k 10 500 python bench.py --config 5 --steps 3 --warmup 1 --no-cpu --no-paralog --no-h2h > $O/c5d.json 2> $O/c5d.err || exit 1
python -c "import json; d=json.load(open('$O/c5d.json')); print(d['value'], d['ms_per_step'], d['kernels_ms'])"
