#!/bin/bash
# per-read rows (cost estimate, its inputs, the search's duration) from the profile build: paralog workload and config 2
set -o pipefail
O=gpurun_out
export TALC_LIB=$PWD/talc_amd/_build/libtalc_hip_prof.so
TALC_PROF_READS=$O/rows_paralog.tsv python -c "
import sys; sys.path.insert(0,'.')
import bench
from talc_amd import lib as T
r = bench.paralog_workload(T, 0, lambda m: None)
print('paralog', r['ms_per_step'])
" || exit 1
TALC_PROF_READS=$O/rows_config2.tsv python bench.py --steps 1 --warmup 0 --no-cpu --no-h2h --no-paralog > $O/rows_c2.json 2> $O/rows_c2.err || exit 1
gzip -f $O/rows_paralog.tsv $O/rows_config2.tsv
ls -la $O/rows_*.gz
