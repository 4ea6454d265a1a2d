#!/bin/bash
set -o pipefail
O=gpurun_out
export TMPDIR=/tmp
ROOT=$(pwd)
for v in old X; do
L=$ROOT/talc_amd/_build/libtalc_hip_$v.so; [ $v = X ] && L=$ROOT/talc_amd/_build/libtalc_hip.so
export TALC_LIB=$L
timeout -k 10 250 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d "$O/ic_$v" -o pmc -- python3 "$ROOT/bench.py" --steps 1 --warmup 0 --no-cpu --no-h2h --no-paralog > "$O/ic_$v.json" 2> "$O/ic_$v.err" || { tail -3 $O/ic_$v.err; exit 1; }
done
python3 - <<'PY'
import csv, glob
for v in ("old","X"):
    agg = {}
    for f in glob.glob("gpurun_out/ic_%s/**/*counter_collection.csv" % v, recursive=True):
        for row in csv.DictReader(open(f)):
            if "k_search" in row["Kernel_Name"]:
                d = agg.setdefault(row["Counter_Name"], {})
                d[row["Dispatch_Id"]] = d.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
    print(v, {c: "%.4g" % max(d.values()) for c, d in sorted(agg.items())})
PY
