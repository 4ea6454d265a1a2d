#!/bin/bash
set -o pipefail
O=gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/full14_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/full14_pytest.log
[ $rc -eq 0 ] || exit 1
bash profiles/r02/scripts/r02_run_final14.sh
