#!/bin/bash
# kept alignment rows for references to 8191 bases: the device test, stress set 105 up to its heaviest read (parity), the whole set's time, config 2 A/B
O=gpurun_out
mkdir -p $O
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "alignment_rows or kept_rows or arena" > $O/r04_rows_test.txt 2>&1 || { tail -30 $O/r04_rows_test.txt; exit 1; }
tail -2 $O/r04_rows_test.txt
: > $O/r04_stress.log
STRESS_READS=1600 bash profiles/r04/scripts/r04_stress.sh 105 || exit 1
timeout -k 10 300 python3 tools/heavy_reads.py 105 6000 2>&1 | tail -1
: > $O/r04_rows_ab.txt
TALC_LIB=talc_amd/_build/libtalc_hip_base.so timeout -k 10 300 python3 tools/search_bench.py --reps 5 --no-paralog >> $O/r04_rows_ab.txt 2>&1 &&
timeout -k 10 300 python3 tools/search_bench.py --reps 5 --no-paralog >> $O/r04_rows_ab.txt 2>&1
grep -v "^+" $O/r04_rows_ab.txt
