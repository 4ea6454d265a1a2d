#!/bin/bash
set -o pipefail
O=gpurun_out
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "seed_and_extension or kept_wavefront or multi_xdrop or nearly_clean" > $O/r04_wide_pytest.log 2>&1 || { tail -30 $O/r04_wide_pytest.log; exit 1; }
tail -3 $O/r04_wide_pytest.log
bash profiles/r04/scripts/r04_clean_prof.sh 100 | grep -E "extnw|xdrop  |x.levels|total|utilisation|mismatch|#xdrop"
