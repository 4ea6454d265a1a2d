#!/bin/bash
set -o pipefail
O=gpurun_out
python -m pytest tests -m gpu -x -q > $O/r02d_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/r02d_pytest.log
python bench.py --steps 5 --warmup 2 > $O/r02d_bench_c2.json 2> $O/r02d_bench_c2.err || exit 1
grep -h "warmup 0\|host-to-host" $O/r02d_bench_c2.err
