#!/bin/bash
set -o pipefail
O=gpurun_out
python -m pytest tests -m gpu -x -q > $O/r02q_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/r02q_pytest.log
bash profiles/collect.sh r02 final > $O/r02q_collect.log 2>&1; echo "collect rc=$?"; tail -3 $O/r02q_collect.log
python bench.py --config 3 --steps 3 --warmup 1 --no-cpu --no-paralog > $O/r02q_bench_c3.json 2> $O/r02q_bench_c3.err || exit 1
python bench.py --config 5 --steps 3 --warmup 1 --no-cpu --no-paralog > $O/r02q_bench_c5.json 2> $O/r02q_bench_c5.err || exit 1
python bench.py --config 4 --steps 3 --warmup 1 --no-cpu --no-paralog --no-h2h > $O/r02q_bench_c4.json 2> $O/r02q_bench_c4.err || exit 1
grep -h "warmup 0\|host-to-host" $O/r02q_bench_c*.err
TALC_LIB=$PWD/talc_amd/_build/libtalc_hip_prof.so TALC_PROF_PRINT=1 python bench.py --steps 1 --warmup 1 --no-cpu --no-h2h --no-paralog > $O/r02q_prof.json 2> $O/r02q_prof.err || exit 1
grep "prof\]" $O/r02q_prof.err | tail -36
