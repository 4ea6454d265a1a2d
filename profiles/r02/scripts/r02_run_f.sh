#!/bin/bash
set -o pipefail
O=gpurun_out
TALC_LIB=$PWD/talc_amd/_build/libtalc_hip_covinline.so python bench.py --steps 5 --warmup 2 --no-cpu --no-h2h --no-paralog > $O/r02f_bench_covinline.json 2> $O/r02f_bench_covinline.err || exit 1
python bench.py --steps 5 --warmup 2 --cpu-backend map > $O/r02f_bench_c2_map.json 2> $O/r02f_bench_c2_map.err || exit 1
grep -h "warmup 0\|paralog\|host-to-host" $O/r02f_bench_*.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02f_bench_c2_map.json'))
print(d['cpu_baseline']); print(d['paralog_workload'])
PY
