#!/bin/bash
# the CLI's GPU tests (files byte for byte against talc_ref; two-GPU rehearsal; config 1), then the whole program on configs 2, 3, 5
set -o pipefail
O=gpurun_out
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_cli.py tests/test_gpu_stress.py -m gpu -x -q -k "cli or config1" > $O/r04_cli_pytest.log 2>&1 || { tail -40 $O/r04_cli_pytest.log; exit 1; }
tail -3 $O/r04_cli_pytest.log
bash profiles/r04/scripts/r04_e2e.sh 2 3 5
