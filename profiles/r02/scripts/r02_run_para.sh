#!/bin/bash
set -o pipefail
O=gpurun_out
TALC_LIB=$PWD/talc_amd/_build/libtalc_hip_prof.so TALC_PROF_PRINT=1 TALC_PROF_SLOW=1 timeout -k 10 400 python bench.py --steps 1 --warmup 0 --no-cpu --no-h2h > $O/para.json 2> $O/para.err || { tail -5 $O/para.err; exit 1; }
grep "prof\]\|slow\]" $O/para.err | tail -70 | grep -v "#reads<\|ff\.\|#ff\|x\.s"
