#!/bin/bash
set -o pipefail
O=gpurun_out
python -m pytest tests -m gpu -x -q > $O/r02m_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/r02m_pytest.log
python bench.py --config 3 --steps 2 --warmup 1 --no-cpu --no-paralog > $O/r02m_bench_c3.json 2> $O/r02m_bench_c3.err || exit 1
grep -h "warmup 0\|host-to-host" $O/r02m_bench_c3.err
