#!/bin/bash
set -o pipefail
O=gpurun_out
TALC_LIB=$PWD/talc_amd/_build/libtalc_hip_prof.so TALC_PROF_PRINT=1 timeout -k 10 500 python tools/slow_reads.py 5 49157 87728 69480 8881 30632 80436 98024 2> $O/slowreads.err || { tail -5 $O/slowreads.err; exit 1; }
grep "pass\|read " $O/slowreads.err; grep "prof\]" $O/slowreads.err | tail -44 | awk '$3 > 0'
