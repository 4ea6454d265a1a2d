#!/bin/bash
set -o pipefail
O=gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -x -q > $O/nw_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -2 $O/nw_pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu --no-h2h > $O/nw_bench.json 2> $O/nw_bench.err || exit 1
python -c "import json; d=json.load(open('$O/nw_bench.json')); print(d['value'], d['ms_per_step'], d['kernels_ms']['search_ms']); print(d['paralog_workload'])"
