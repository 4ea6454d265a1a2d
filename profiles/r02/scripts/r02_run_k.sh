#!/bin/bash
set -o pipefail
O=gpurun_out
python -m pytest tests -m gpu -x -q -k "default_parameters or golden or batch_composition or edge_inputs or read_stats" > $O/r02k_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r02k_pytest.log
python bench.py --steps 5 --warmup 2 --no-cpu --no-h2h --no-paralog > $O/r02k_bench_c2.json 2> $O/r02k_bench_c2.err || exit 1
TALC_ORDER=length python bench.py --steps 5 --warmup 2 --no-cpu --no-h2h --no-paralog > $O/r02k_bench_c2_len.json 2> $O/r02k_bench_c2_len.err || exit 1
grep -h "warmup 1" $O/r02k_bench_c2.err $O/r02k_bench_c2_len.err
TALC_LIB=$PWD/talc_amd/_build/libtalc_hip_prof.so TALC_PROF_PRINT=1 python bench.py --steps 1 --warmup 1 --no-cpu --no-h2h --no-paralog > $O/r02k_prof.json 2> $O/r02k_prof.err || exit 1
grep "prof\]" $O/r02k_prof.err | tail -9
