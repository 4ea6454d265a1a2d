#!/bin/bash
# the stress comparison (HIP path against the oracle, read by read) on the sets named (default: all), STRESS_READS each
O=gpurun_out
mkdir -p $O
timeout -k 10 1100 python3 tools/stress_branching.py "$@" 2>&1 | tee -a $O/r04_stress.log
