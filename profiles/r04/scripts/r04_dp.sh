#!/bin/bash
set -o pipefail
O=gpurun_out
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -x -q > $O/r04_dp_pytest.log 2>&1 || { tail -30 $O/r04_dp_pytest.log; exit 1; }
tail -2 $O/r04_dp_pytest.log
bash profiles/r04/scripts/r04_ab.sh "$@"
