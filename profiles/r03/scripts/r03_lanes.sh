#!/bin/bash
# Lane occupancy of the vector instructions of k_search / k_coverage / k_structure: separate --pmc passes, --kernel-trace
# only.  SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU = average share of the 64 lanes that are active per vector instruction
# (both in quad-cycles x lanes resp. quad-cycles); SQ_INSTS_VALU for the instruction count.
set -o pipefail
O=gpurun_out
TAG=${1:-lanes}
export TMPDIR=/tmp
ROOT=$(pwd)
rocprofv3 -L > "$O/${TAG}_counters_list.txt" 2>&1 || true
i=0
for C in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout -k 10 250 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$O/${TAG}_$i" -o pmc -- python3 "$ROOT/bench.py" --steps 1 --warmup 0 --no-cpu --no-h2h --no-paralog --no-e2e > "$O/${TAG}_$i.json" 2> "$O/${TAG}_$i.err" || { echo "pass $i failed"; tail -5 "$O/${TAG}_$i.err"; exit 1; }
  echo "pass $i done"
done
python3 - "$TAG" <<'PY'
import csv, glob, json, sys
tag = sys.argv[1]
agg = {}
for f in glob.glob("gpurun_out/%s_*/**/*counter_collection.csv" % tag, recursive=True):
    for row in csv.DictReader(open(f)):
        for k in ("k_search", "k_coverage", "k_structure"):
            if k in row["Kernel_Name"]:
                d = agg.setdefault((k, row["Counter_Name"]), {})
                d[row["Dispatch_Id"]] = d.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
res = {}
for (k, c), d in sorted(agg.items()):
    res.setdefault(k, {})[c] = max(d.values())
for k, d in res.items():
    if d.get("SQ_ACTIVE_INST_VALU") and d.get("SQ_THREAD_CYCLES_VALU"):
        d["active_lanes_per_valu_inst"] = d["SQ_THREAD_CYCLES_VALU"] / d["SQ_ACTIVE_INST_VALU"]
json.dump(res, open("gpurun_out/%s_sq.json" % tag, "w"), indent=1)
print(json.dumps(res, indent=1))
PY
