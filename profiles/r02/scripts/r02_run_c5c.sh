#!/bin/bash
set -o pipefail
O=gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_fullsize.py -m gpu -x -q -k "not config4" > $O/c5c_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $O/c5c_pytest.log
[ $rc -eq 0 ] || exit 1
TALC_LIB=$PWD/talc_amd/_build/libtalc_hip_prof.so TALC_PROF_PRINT=1 TALC_PROF_SLOW=1 timeout -k 10 500 python bench.py --config 5 --steps 1 --warmup 1 --no-cpu --no-h2h --no-paralog > $O/c5prof.json 2> $O/c5prof.err || { tail -5 $O/c5prof.err; exit 1; }
grep "prof\]" $O/c5prof.err | tail -41 | grep "utilis\|evalfull\|cycle\|xdrop\|ffwd\|stepb\|probe\|child\|total \|maxread"
grep "slow\]" $O/c5prof.err | tail -6
timeout -k 10 500 python bench.py --config 5 --steps 3 --warmup 1 --no-cpu --no-paralog --no-h2h > $O/c5c.json 2> $O/c5c.err || exit 1
python -c "import json; d=json.load(open('$O/c5c.json')); print(d['value'], d['ms_per_step'], d['kernels_ms'])"
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu --no-paralog --no-h2h > $O/c5c_c2.json 2> $O/c5c_c2.err || exit 1
python -c "import json; d=json.load(open('$O/c5c_c2.json')); print(d['value'], d['ms_per_step'], d['kernels_ms'])"
