#!/bin/bash
# round 4, first call: the new tests, the whole GPU suite with durations, the default bench line
set -o pipefail
O=gpurun_out
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=30 > $O/r04_first_pytest.log 2>&1 || { tail -40 $O/r04_first_pytest.log; exit 1; }
tail -45 $O/r04_first_pytest.log
timeout -k 10 400 python bench.py > $O/r04_first_bench.json 2> $O/r04_first_bench.err || { tail -30 $O/r04_first_bench.err; exit 1; }
grep "bench r0" $O/r04_first_bench.err | tail -12
