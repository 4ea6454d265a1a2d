#!/bin/bash
set -o pipefail
O=gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -x -q > $O/r02x_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -6 $O/r02x_pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-paralog --no-cpu > $O/r02x_bench.json 2> $O/r02x_bench.err || exit 1
cat $O/r02x_bench.json | python -c "import json,sys; d=json.load(sys.stdin); print(d['value'], d['ms_per_step'], d['kernels_ms'])"
