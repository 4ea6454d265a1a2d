#!/bin/bash
set -o pipefail
O=gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/full_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/full_pytest.log
