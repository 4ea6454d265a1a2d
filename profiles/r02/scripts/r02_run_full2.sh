#!/bin/bash
set -o pipefail
O=gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/full_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/full_pytest.log
[ $rc -eq 0 ] || exit 1
for i in 1 2; do
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu --no-paralog --no-h2h > $O/f2_c2.json 2> $O/f2_c2.err || exit 1
python -c "import json; d=json.load(open('$O/f2_c2.json')); print(d['value'], d['ms_per_step'], d['kernels_ms']['search_ms'])"
done
