#!/bin/bash
# fork_step: parity tests of the correction path, then the same-box A/B against the round-3 library, then the category profile
set -o pipefail
O=gpurun_out
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_stress.py -m gpu -x -q -k "not half_corrected" > $O/r04_fork_pytest.log 2>&1 || { tail -30 $O/r04_fork_pytest.log; exit 1; }
tail -3 $O/r04_fork_pytest.log
bash profiles/r04/scripts/r04_ab.sh libtalc_hip_r3.so libtalc_hip.so libtalc_hip_w6d256.so || exit 1
bash profiles/r04/scripts/r04_prof.sh > /dev/null || exit 1
grep -E "stepb|forkstep|ffwd|total|stepe|#ffstop|probe|child" $O/r04_prof_config2.txt | tail -24
