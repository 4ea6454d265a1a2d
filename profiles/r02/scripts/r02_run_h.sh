#!/bin/bash
set -o pipefail
O=gpurun_out
python -m pytest tests -m gpu -x -q -k "edit_distance or mixed_lengths or long_gaps or default_parameters or golden or branching" > $O/r02h_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/r02h_pytest.log
python bench.py --config 5 --steps 2 --warmup 1 --no-cpu --no-h2h --no-paralog > $O/r02h_bench_c5.json 2> $O/r02h_bench_c5.err || exit 1
grep -h "warmup 0" $O/r02h_bench_c5.err
TALC_LIB=$PWD/talc_amd/_build/libtalc_hip_prof.so TALC_PROF_PRINT=1 python bench.py --config 5 --reads 20000 --steps 1 --warmup 0 --no-cpu --no-h2h --no-paralog > $O/r02h_prof_c5.json 2> $O/r02h_prof_c5.err || exit 1
grep "prof\]" $O/r02h_prof_c5.err | head -22
