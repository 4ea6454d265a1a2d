#!/bin/bash
set -o pipefail
O=gpurun_out
python -m pytest tests -m gpu -x -q -k "default_parameters or golden or structure or trace or read_stats or parameter_variants" > $O/r02s_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r02s_pytest.log
python bench.py --steps 6 --warmup 2 --no-cpu --no-h2h --no-paralog > $O/r02s_bench_c2.json 2> $O/r02s_bench_c2.err || exit 1
grep -h "warmup 1" $O/r02s_bench_c2.err
