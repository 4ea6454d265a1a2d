#!/bin/bash
# Instruction-cache counters of k_search (config 2 and the branching workload's launch are both in the run): separate --pmc
# passes, --kernel-trace only.
set -o pipefail
O=gpurun_out
TAG=${1:-icache}
export TMPDIR=/tmp
ROOT=$(pwd)
i=0
for C in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQC_TC_INST_REQ SQC_TC_REQ SQC_TC_STALL SQC_ICACHE_BUSY_CYCLES"; do
  i=$((i+1))
  timeout -k 10 250 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$O/${TAG}_$i" -o pmc -- python3 "$ROOT/bench.py" --steps 1 --warmup 0 --no-cpu --no-h2h --no-e2e > "$O/${TAG}_$i.json" 2> "$O/${TAG}_$i.err" || { echo "pass $i failed"; tail -5 "$O/${TAG}_$i.err"; exit 1; }
  echo "pass $i done"
done
python3 - "$TAG" <<'PY'
import csv, glob, json, sys
tag = sys.argv[1]
agg = {}
for f in glob.glob("gpurun_out/%s_*/**/*counter_collection.csv" % tag, recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_search" in row["Kernel_Name"]:
            d = agg.setdefault(row["Counter_Name"], {})
            d[row["Dispatch_Id"]] = d.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
# the dispatches of k_search in launch order: config 2's step (+ its tiny retry launch), then the branching workload's
res = {c: [v for _, v in sorted(d.items(), key=lambda kv: int(kv[0]))] for c, d in sorted(agg.items())}
json.dump(res, open("gpurun_out/%s.json" % tag, "w"), indent=1)
print(json.dumps(res, indent=1))
PY
