#!/bin/bash
# GPU session B: k_coverage v3 (queue compaction, 24-bit-multiply hashes) with and without minimizer blocks + SQ counters
set -o pipefail
O=gpurun_out
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q -k "coverage or lookups or default_parameters or golden or mixed_lengths" > $O/r02b_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r02b_pytest.log
python bench.py --steps 5 --warmup 2 --no-cpu --no-h2h > $O/r02b_bench_c2.json 2> $O/r02b_bench_c2.err || exit 1
TALC_LIB=$PWD/talc_amd/_build/libtalc_hip_mini.so python bench.py --steps 5 --warmup 2 --no-cpu --no-h2h > $O/r02b_bench_c2_mini.json 2> $O/r02b_bench_c2_mini.err || exit 1
python bench.py --config 5 --steps 2 --warmup 1 --no-cpu --no-h2h > $O/r02b_bench_c5.json 2> $O/r02b_bench_c5.err || exit 1
grep -h "warmup 0" $O/r02b_bench_*.err
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/r02b_pmc_sq -o pmc -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-h2h > $O/r02b_pmc_sq.json 2> $O/r02b_pmc_sq.err || echo "pmc failed"
python3 - <<'PY'
import csv, glob
agg={}
for f in glob.glob('gpurun_out/r02b_pmc_sq/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k=row['Kernel_Name'].split('(')[0][:40]
        if 'k_coverage' in k or 'k_search' in k or 'k_structure' in k:
            agg.setdefault(k,{}).setdefault(row['Counter_Name'],0.0)
            agg[k][row['Counter_Name']]+=float(row['Counter_Value'])
for k,v in agg.items(): print(k, {a:'%.3e'%b for a,b in v.items()})
PY
