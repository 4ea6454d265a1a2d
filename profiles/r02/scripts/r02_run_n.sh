#!/bin/bash
set -o pipefail
O=gpurun_out
for v in "" _noprio _prio1024; do
  if [ -n "$v" ]; then export TALC_LIB=$PWD/talc_amd/_build/libtalc_hip$v.so; else unset TALC_LIB; fi
  python bench.py --steps 6 --warmup 2 --no-cpu --no-h2h --no-paralog > $O/r02n_bench_c2$v.json 2> $O/r02n_bench_c2$v.err || exit 1
  echo "variant [$v]"; grep -h "warmup 1" $O/r02n_bench_c2$v.err
done
