#!/bin/bash
# instruction / scalar cache and LDS counters of k_search: separate --pmc passes, --kernel-trace only
set -o pipefail
O=gpurun_out
export TMPDIR=/tmp
ROOT=$(pwd)
i=0
for C in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_IFETCH" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL" "SQ_IFETCH_LEVEL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM"; do
  i=$((i+1))
  timeout -k 10 250 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$O/sq2_$i" -o pmc -- python3 "$ROOT/bench.py" --steps 1 --warmup 0 --no-cpu --no-h2h --no-paralog > "$O/sq2_$i.json" 2> "$O/sq2_$i.err" || { echo "pass $i failed"; tail -5 "$O/sq2_$i.err"; exit 1; }
  echo "pass $i done"
done
python3 - <<'PY'
import csv, glob
agg = {}
for f in glob.glob("gpurun_out/sq2_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        for k in ("k_search", "k_coverage", "k_structure"):
            if k in row["Kernel_Name"]:
                d = agg.setdefault((k, row["Counter_Name"]), {})
                d[row["Dispatch_Id"]] = d.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
for (k, c), d in sorted(agg.items()):
    print(k, c, "%.4g" % max(d.values()))
PY
