#!/bin/bash
set -o pipefail
O=gpurun_out
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_cli.py tests/test_gpu_stress.py -m gpu -x -q -k "text_dump or device_built or cli or config1" > $O/r04_parse_pytest.log 2>&1 || { tail -40 $O/r04_parse_pytest.log; exit 1; }
tail -3 $O/r04_parse_pytest.log
bash profiles/r04/scripts/r04_e2e.sh 2 3 5
