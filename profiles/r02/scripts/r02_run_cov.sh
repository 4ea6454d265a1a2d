#!/bin/bash
set -o pipefail
O=gpurun_out
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu --no-paralog --no-h2h > $O/cov.json 2> $O/cov.err || exit 1
python -c "import json; d=json.load(open('$O/cov.json')); print(d['value'], d['ms_per_step'], d['kernels_ms'], d['roofline']['frac'])"
