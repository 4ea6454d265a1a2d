#!/bin/bash
set -o pipefail
O=gpurun_out
TALC_LIB=$PWD/talc_amd/_build/libtalc_hip_prof.so TALC_PROF_PRINT=1 timeout -k 10 500 python tools/slow_reads.py paralog 7006 7185 541 2> $O/slowpara.err || { tail -5 $O/slowpara.err; exit 1; }
grep "pass\|read " $O/slowpara.err; grep "prof\]" $O/slowpara.err | tail -44 | awk '$3 > 0' | grep -v "#reads\|x\.s"
