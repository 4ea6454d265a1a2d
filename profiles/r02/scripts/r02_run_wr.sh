#!/bin/bash
set -o pipefail
O=gpurun_out
export TMPDIR=/tmp
ROOT=$(pwd)
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -x -q > $O/wr_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -2 $O/wr_pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 250 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$O/wr_pmc" -o pmc -- python3 "$ROOT/bench.py" --steps 1 --warmup 0 --no-cpu --no-h2h --no-paralog > "$O/wr.json" 2> "$O/wr.err" || { tail -5 $O/wr.err; exit 1; }
python3 - <<'PY'
import csv, glob
agg = {}
for f in glob.glob("gpurun_out/wr_pmc/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        for k in ("k_search", "k_coverage"):
            if k in row["Kernel_Name"]:
                d = agg.setdefault((k, row["Counter_Name"]), {})
                d[row["Dispatch_Id"]] = d.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
for (k, c), d in sorted(agg.items()):
    print(k, c, "%.4g GB" % (max(d.values()) / 1e6))
PY
timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu --no-h2h --no-paralog > $O/wr_bench.json 2>/dev/null || exit 1
python -c "import json; d=json.load(open('$O/wr_bench.json')); print(d['value'], d['ms_per_step'], d['kernels_ms'])"
