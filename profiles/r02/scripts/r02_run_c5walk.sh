#!/bin/bash
# config 5 with the walk tables forced on (TALC_WALK=1): is the 96 GB rule still right?
set -o pipefail
O=gpurun_out
TALC_WALK=1 timeout -k 10 600 python bench.py --config 5 --steps 2 --warmup 1 --no-cpu --no-h2h --no-paralog > $O/c5walk.json 2> $O/c5walk.err || { tail -5 $O/c5walk.err; exit 1; }
python -c "import json; d=json.load(open('$O/c5walk.json')); print(d['value'], d['ms_per_step'], d['kernels_ms'], d['config']['table_device_bytes'])"
