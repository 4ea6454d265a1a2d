#!/bin/bash
# work-queue order: sweep of the edge cost coefficients (k_structure's estimate), search_ms of config 2
set -o pipefail
O=gpurun_out
for q in "2800 135" "2800 90" "2800 60" "2800 40" "4000 60" "4000 30" "2000 80"; do
set -- $q
TALC_COST_LIN=$1 TALC_COST_QUAD=$2 timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu --no-h2h --no-paralog > $O/cost.json 2> $O/cost.err || { tail -5 $O/cost.err; exit 1; }
python -c "import json; d=json.load(open('$O/cost.json')); print('lin $1 quad $2', d['kernels_ms']['search_ms'])"
done
