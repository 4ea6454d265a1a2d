#!/bin/bash
set -o pipefail
O=gpurun_out
python -m pytest tests -m gpu -x -q -k "default_parameters or branching or golden or junction or reverse or edge_inputs" > $O/r02e_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r02e_pytest.log
python bench.py --steps 5 --warmup 2 --no-cpu --no-h2h > $O/r02e_bench_c2.json 2> $O/r02e_bench_c2.err || exit 1
TALC_LIB=$PWD/talc_amd/_build/libtalc_hip_prof.so TALC_PROF_PRINT=1 python bench.py --steps 1 --warmup 0 --no-cpu --no-h2h --reads 30000 > $O/r02e_prof.json 2> $O/r02e_prof.err || exit 1
grep -h "warmup 0" $O/r02e_bench_c2.err; grep "prof\]" $O/r02e_prof.err | head -30
