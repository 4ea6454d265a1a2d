#!/bin/bash
# GPU session A of round 2: suite, default bench, occupancy variants, config-5 walk A/B, category profile
set -o pipefail
O=gpurun_out
python -m pytest tests -m gpu -x -q -s > $O/r02a_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/r02a_pytest.log
python bench.py --steps 5 --warmup 2 > $O/r02a_bench_c2.json 2> $O/r02a_bench_c2.err || exit 1
for w in 4 6; do TALC_LIB=$PWD/talc_amd/_build/libtalc_hip_w$w.so python bench.py --steps 5 --warmup 2 --no-cpu --no-h2h > $O/r02a_bench_c2_w$w.json 2> $O/r02a_bench_c2_w$w.err || exit 1; done
TALC_LIB=$PWD/talc_amd/_build/libtalc_hip_prof.so TALC_PROF_PRINT=1 python bench.py --steps 1 --warmup 1 --no-cpu --no-h2h --reads 30000 > $O/r02a_prof.json 2> $O/r02a_prof.err || exit 1
TALC_WALK=0 python bench.py --config 5 --steps 2 --warmup 1 --no-cpu --no-h2h > $O/r02a_bench_c5_walk0.json 2> $O/r02a_bench_c5_walk0.err || exit 1
TALC_WALK=1 python bench.py --config 5 --steps 2 --warmup 1 --no-cpu --no-h2h > $O/r02a_bench_c5_walk1.json 2> $O/r02a_bench_c5_walk1.err || exit 1
grep -h "warmup 0" $O/r02a_bench_*.err
