#!/bin/bash
set -o pipefail
O=gpurun_out
TALC_LIB=$PWD/talc_amd/_build/libtalc_hip_prof.so TALC_PROF_PRINT=1 python bench.py --config 5 --reads 20000 --steps 1 --warmup 0 --no-cpu --no-h2h --no-paralog > $O/r02g_prof_c5.json 2> $O/r02g_prof_c5.err || exit 1
grep "prof\]" $O/r02g_prof_c5.err | head -28
bash profiles/collect.sh r02 mid > $O/r02g_collect.log 2>&1; echo "collect rc=$?"; tail -5 $O/r02g_collect.log
