#!/bin/bash
set -o pipefail
O=gpurun_out
for v in base w6 w4 mmc base; do
  if [ $v = base ]; then unset TALC_LIB; else export TALC_LIB=$PWD/talc_amd/_build/libtalc_hip_v_$v.so; fi
  python bench.py --steps 6 --warmup 2 --no-cpu --no-h2h --no-paralog > $O/var_$v.json 2> $O/var_$v.err || { tail -5 $O/var_$v.err; exit 1; }
  python -c "import json; d=json.loads(open('$O/var_$v.json').read().strip().splitlines()[-1]); print('$v', round(d['ms_per_step'],2), {k: round(x,2) for k,x in d['kernels_ms'].items()})"
done
