#!/bin/bash
# the full-size cases (configs 2-5 through properties + oracle spot checks) and the bench lines of configs 3, 4, 5
set -o pipefail
O=gpurun_out
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -s > $O/r04_fullsize_pytest.log 2>&1 || { tail -30 $O/r04_fullsize_pytest.log; exit 1; }
grep -E "passed|failed|config" $O/r04_fullsize_pytest.log | tail -12
for c in 3 4 5; do
  timeout -k 10 600 python bench.py --config $c --steps 2 --warmup 1 --no-cpu --no-h2h --no-paralog --no-e2e > $O/r04_bench_c$c.json 2> $O/r04_bench_c$c.err || { tail -20 $O/r04_bench_c$c.err; exit 1; }
  python -c "
import json; d=json.loads(open('$O/r04_bench_c$c.json').read().strip().splitlines()[-1])
print('config $c value %.3g ms/step %.1f cov_frac %.3f cov_ms %.2f search_ms %.1f table_bytes %.1f GB' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['kernels_ms']['coverage_ms'], d['kernels_ms']['search_ms'], d['config']['table_device_bytes']/1e9))"
done
