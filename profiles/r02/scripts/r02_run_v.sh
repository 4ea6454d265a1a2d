#!/bin/bash
set -o pipefail
O=gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -x -q > $O/r02v_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -6 $O/r02v_pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-paralog > $O/r02v_bench.json 2> $O/r02v_bench.err || exit 1
cat $O/r02v_bench.json | python -c "import json,sys; d=json.load(sys.stdin); print(d['value'], d['ms_per_step'], d['kernels_ms'], d['roofline']['traffic'], d.get('cpu_baseline',{}).get('parity_with_gpu_on_sample'))"
python -m talc_amd.build --prof > /dev/null 2>&1
TALC_LIB=$PWD/talc_amd/_build/libtalc_hip_prof.so TALC_PROF_PRINT=1 timeout -k 10 200 python bench.py --steps 1 --warmup 1 --no-cpu --no-h2h --no-paralog > $O/r02v_prof.json 2> $O/r02v_prof.err || exit 1
grep "prof\]" $O/r02v_prof.err | tail -36
