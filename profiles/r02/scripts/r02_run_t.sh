#!/bin/bash
# SQ counters of k_search (is the search bound by instruction issue?): separate --pmc passes, --kernel-trace only
set -o pipefail
O=gpurun_out
export TMPDIR=/tmp
ROOT=$(pwd)
i=0
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAVES"; do
  i=$((i+1))
  timeout -k 10 250 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$O/sq_$i" -o pmc -- python3 "$ROOT/bench.py" --steps 1 --warmup 0 --no-cpu --no-h2h --no-paralog > "$O/sq_$i.json" 2> "$O/sq_$i.err" || { echo "pass $i failed"; tail -5 "$O/sq_$i.err"; exit 1; }
  echo "pass $i done"
done
python3 - <<'PY'
import csv, glob
agg = {}
for f in glob.glob("gpurun_out/sq_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        for k in ("k_search", "k_coverage", "k_structure"):
            if k in row["Kernel_Name"]:
                d = agg.setdefault((k, row["Counter_Name"]), {})
                d[row["Dispatch_Id"]] = d.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
for (k, c), d in sorted(agg.items()):
    print(k, c, "%.4g" % max(d.values()))
PY
