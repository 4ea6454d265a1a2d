#!/bin/bash
O=gpurun_out
T="tests/test_gpu_parity.py::test_correction_parameter_variants"
timeout -k 10 200 python -m pytest -m gpu -x -q "$T" > $O/dbg1.log 2>&1; echo "all variants rc=$?"; tail -1 $O/dbg1.log
[ -s $O/dbg1.log ] && head -1 $O/dbg1.log
TALC_NO_ROWS=1 timeout -k 10 200 python -m pytest -m gpu -x -q "$T" > $O/dbg2.log 2>&1; echo "no-rows rc=$?"; tail -1 $O/dbg2.log
