#!/bin/bash
# Collects the profile artifacts of one round on the GPU box (run from the repo root):
#   profiles/collect.sh r01 final
# writes profiles/<round>/<tag>_bench.json, <tag>_bench_under_rocprof.json, <tag>_kernel_stats.csv and
# <tag>_pmc_traffic.json (FETCH_SIZE / WRITE_SIZE in two separate counter passes, --kernel-trace only).
# The program itself follows "--" (no env/bash wrappers under rocprofv3).
set -e -o pipefail
ROUND=${1:-r01}
TAG=${2:-final}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
DST=$ROOT/gpurun_out/profiles/$ROUND
mkdir -p "$OUT" "$DST"
export TMPDIR=/tmp

python3 bench.py --steps 10 --warmup 3 > "$DST/${TAG}_bench.json"
echo "[collect] bench done"

rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- python3 "$ROOT/bench.py" --steps 10 --warmup 3 --no-cpu --no-h2h --no-paralog --no-e2e > "$DST/${TAG}_bench_under_rocprof.json"
STATS=$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1)
cp "$STATS" "$DST/${TAG}_kernel_stats.csv"
echo "[collect] kernel stats done"

for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/pmc_$C" -o pmc -- python3 "$ROOT/bench.py" --steps 1 --warmup 0 --no-cpu --no-h2h --no-paralog --no-e2e > "$OUT/pmc_$C.json"
  echo "[collect] pmc $C done"
done

python3 - "$OUT" "$DST/${TAG}_pmc_traffic.json" <<'EOF'
import csv, glob, json, os, sys
out, dst = sys.argv[1], sys.argv[2]
sys.path.insert(0, os.getcwd())
from talc_amd import build as B
res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes, --kernel-trace only), python3 bench.py "
                 "--steps 1 --warmup 0 --no-cpu --no-h2h --no-paralog, config 2 (100000 reads); counter values are KB per dispatch; "
                 "gfx950 note: FETCH_SIZE = TCC_EA0_RDREQ x 64 B and counts Infinity-Cache hits (requests that leave the XCD L2s)",
       "lib_source_hash": B.source_hash(), "baseline_config": 2, "reads": 100000}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob("%s/pmc_%s/**/*counter_collection.csv" % (out, c), recursive=True)
    agg = {}
    for f in files:
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != c:
                continue
            name = row["Kernel_Name"]
            for k in ("k_coverage", "k_search", "k_structure", "k_build_walk"):
                if k in name:
                    d = agg.setdefault(k, {})
                    d[row["Dispatch_Id"]] = d.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
    for k, d in agg.items():
        res.setdefault(k, {})[c + "_KB"] = max(d.values())   # the step's dispatch (k_search's retry launch is tiny)
json.dump(res, open(dst, "w"), indent=1)
json.dump(res, open(os.path.join(os.getcwd(), "gpurun_out", "pmc_traffic.json"), "w"), indent=1)   # -> profiles/pmc_traffic.json (bench.py reads it)
print(json.dumps(res))
EOF
echo "[collect] wrote $DST"
