#!/bin/bash
set -o pipefail
O=gpurun_out
python -m pytest tests -m gpu -x -q > $O/r02i_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/r02i_pytest.log
python bench.py --steps 5 --warmup 2 --no-cpu > $O/r02i_bench_c2.json 2> $O/r02i_bench_c2.err || exit 1
python bench.py --config 5 --steps 2 --warmup 1 --no-cpu --no-paralog > $O/r02i_bench_c5.json 2> $O/r02i_bench_c5.err || exit 1
python bench.py --config 3 --steps 2 --warmup 1 --no-cpu --no-paralog > $O/r02i_bench_c3.json 2> $O/r02i_bench_c3.err || exit 1
grep -h "warmup 0\|host-to-host\|paralog" $O/r02i_bench_*.err
TALC_LIB=$PWD/talc_amd/_build/libtalc_hip_prof.so TALC_PROF_PRINT=1 python bench.py --config 5 --reads 20000 --steps 1 --warmup 0 --no-cpu --no-h2h --no-paralog > $O/r02i_prof_c5.json 2> $O/r02i_prof_c5.err || exit 1
grep "prof\]" $O/r02i_prof_c5.err | head -22
