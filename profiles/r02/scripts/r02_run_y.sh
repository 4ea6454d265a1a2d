#!/bin/bash
set -o pipefail
O=gpurun_out
for b in 10 14 20 28; do
TALC_FILTER_BITS=$b timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu --no-h2h --no-paralog > $O/r02y_b$b.json 2> $O/r02y_b$b.err || { tail -5 $O/r02y_b$b.err; exit 1; }
python -c "import json; d=json.load(open('$O/r02y_b$b.json')); print('bits $b', d['kernels_ms']['coverage_ms'])"
done
