#!/bin/bash
set -o pipefail
O=gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "edit_distance or lcs or mixed_lengths or golden or long_gaps" > $O/c5b_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $O/c5b_pytest.log
[ $rc -eq 0 ] || exit 1
TALC_LIB=$PWD/talc_amd/_build/libtalc_hip_prof.so TALC_PROF_PRINT=1 TALC_PROF_SLOW=1 timeout -k 10 500 python bench.py --config 5 --steps 1 --warmup 1 --no-cpu --no-h2h --no-paralog > $O/c5prof.json 2> $O/c5prof.err || { tail -5 $O/c5prof.err; exit 1; }
grep "prof\]" $O/c5prof.err | tail -41 | grep "utilis\|evalfull\|cycle\|xdrop\|ffwd\|stepb\|total \|maxread"
grep "slow\]" $O/c5prof.err | tail -14
timeout -k 10 500 python bench.py --config 5 --steps 3 --warmup 1 --no-cpu --no-paralog > $O/c5b.json 2> $O/c5b.err || exit 1
python -c "import json; d=json.load(open('$O/c5b.json')); print(d['value'], d['ms_per_step'], d['kernels_ms'], d['host_to_host']['value'])"
