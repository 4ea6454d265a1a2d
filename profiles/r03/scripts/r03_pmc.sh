#!/bin/bash
# FETCH_SIZE / WRITE_SIZE (separate passes) + VMEM instruction counts of the kernels of one config-2 step
set -o pipefail
O=gpurun_out
TAG=${1:-pmc}
export TMPDIR=/tmp
ROOT=$(pwd)
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU"; do
  i=$((i+1))
  timeout -k 10 250 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$O/${TAG}_$i" -o pmc -- python3 "$ROOT/bench.py" --steps 1 --warmup 0 --no-cpu --no-h2h --no-paralog --no-e2e > "$O/${TAG}_$i.json" 2> "$O/${TAG}_$i.err" || { echo "pass $i failed"; tail -5 "$O/${TAG}_$i.err"; exit 1; }
done
python3 - "$TAG" <<'PY'
import csv, glob, sys
tag = sys.argv[1]
agg = {}
for f in glob.glob("gpurun_out/%s_*/**/*counter_collection.csv" % tag, recursive=True):
    for row in csv.DictReader(open(f)):
        for k in ("k_search", "k_coverage", "k_structure"):
            if k in row["Kernel_Name"]:
                d = agg.setdefault((k, row["Counter_Name"]), {})
                d[row["Dispatch_Id"]] = d.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
for (k, c), d in sorted(agg.items()):
    print(k, c, "%.4g" % max(d.values()))
PY
