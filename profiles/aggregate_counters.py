#!/usr/bin/env python3
"""Turns the rocprofv3 --pmc passes of one collection into the two files bench.py reads (profiles/pmc_traffic.json,
profiles/sq_search.json) and their per-round copies.
    python3 profiles/aggregate_counters.py <collection dir> <destination dir> <tag>
The collection dir holds one sub-directory per pass: pmc_c<config>_<COUNTER>/ (FETCH_SIZE, WRITE_SIZE) and sq_<n>/ (SQ_*
groups, config 2).  A pass is one process, so a dispatch is identified by (file, Dispatch_Id): values are summed over the
counter's instances of one dispatch of one file, and the kernel's figure is its largest dispatch (k_search's retry launch is
tiny).  Counter units as the guide prescribes: FETCH_SIZE / WRITE_SIZE in KB (gfx950: FETCH_SIZE = TCC_EA0_RDREQ x 64 B and
counts Infinity-Cache hits: requests that leave the XCD L2s); SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* in quad-cycles
summed over the waves, SQ_BUSY_CYCLES in cycles summed over the 32 shader engines (34.4 ms at 2.35 GHz = 80.9 M cycles:
2.59 G / 32)."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
KERNELS = ("k_coverage", "k_search", "k_structure", "k_build_walk")
READS = {2: 100_000, 3: 1_000_000, 4: 1_000_000, 5: 100_000}


def per_kernel(pass_dir, counters=None):
    """{kernel: {counter: value of the kernel's largest dispatch}} of one pass directory."""
    agg = {}
    for f in glob.glob(os.path.join(pass_dir, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            c = row.get("Counter_Name")
            if counters and c not in counters:
                continue
            for k in KERNELS:
                if k in row["Kernel_Name"]:
                    d = agg.setdefault((k, c), {})
                    key = (f, row["Dispatch_Id"])
                    d[key] = d.get(key, 0.0) + float(row["Counter_Value"])
    out = {}
    for (k, c), d in agg.items():
        out.setdefault(k, {})[c] = max(d.values())
    return out


def main():
    coll, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3]
    from talc_amd import build as B
    lib_hash = B.source_hash()
    os.makedirs(dst, exist_ok=True)
    traffic = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only), python3 bench.py --config C "
                         "--steps 1 --warmup 0 --no-cpu --no-h2h --no-paralog --no-e2e; KB per dispatch; gfx950: FETCH_SIZE = TCC_EA0_RDREQ x 64 B, "
                         "Infinity-Cache hits included (requests that leave the XCD L2s)",
               "lib_source_hash": lib_hash, "configs": {}}
    for cfg in (2, 3, 4, 5):
        entry = {}
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(coll, "pmc_c%d_%s" % (cfg, c))
            if not os.path.isdir(d):
                continue
            for k, v in per_kernel(d, {c}).items():
                entry.setdefault(k, {})[c + "_KB"] = v[c]
        if entry:
            entry["reads"] = READS[cfg]
            traffic["configs"][str(cfg)] = entry
    sq = {}
    for d in sorted(glob.glob(os.path.join(coll, "sq_*"))):
        for k, v in per_kernel(d).items():
            sq.setdefault(k, {}).update(v)
    res = {"source": "rocprofv3 --pmc SQ_* (three separate passes, --kernel-trace only), python3 bench.py --steps 1 --warmup 0 --no-cpu --no-h2h "
                     "--no-paralog --no-e2e, config 2; per-wave counters in quad-cycles summed over the waves, SQ_BUSY_CYCLES over the 32 shader engines",
           "lib_source_hash": lib_hash, "baseline_config": 2, "reads": READS[2], "kernels": sq}
    s = sq.get("k_search", {})
    if s.get("SQ_WAVE_CYCLES") and s.get("SQ_BUSY_CYCLES"):
        launch_quads = s["SQ_BUSY_CYCLES"] / 32.0 / 4.0      # the launch's duration in quad-cycles (SQ_BUSY_CYCLES: cycles, summed over the 32 shader engines)
        simds = 256 * 4
        res["wait_frac"] = s.get("SQ_WAIT_ANY", 0.0) / s["SQ_WAVE_CYCLES"]              # share of wave life parked at s_waitcnt
        res["wait_inst_frac"] = s.get("SQ_WAIT_INST_ANY", 0.0) / s["SQ_WAVE_CYCLES"]    # ... stalled at issue
        res["valu_issue_frac"] = s.get("SQ_ACTIVE_INST_VALU", 0.0) / (simds * launch_quads)   # share of the SIMDs' time a vector instruction was executing
        res["waves_per_simd"] = s["SQ_WAVE_CYCLES"] / (simds * launch_quads)            # resident waves per SIMD, averaged over the launch
        if s.get("SQ_ACTIVE_INST_VALU"):
            res["lanes_per_valu"] = s.get("SQ_THREAD_CYCLES_VALU", 0.0) / s["SQ_ACTIVE_INST_VALU"]
        res["derived_note"] = ("one wave64 vector instruction occupies a SIMD's vector pipe for one quad-cycle (tools/issue_rate.hip: 0.24 "
                               "instructions per cycle per SIMD for every integer, 64-bit, double, lane-read and DPP instruction tried), so "
                               "valu_issue_frac is also SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x launch cycles)")
    for name, obj in (("pmc_traffic.json", traffic), ("sq_search.json", res)):
        json.dump(obj, open(os.path.join(dst, "%s_%s" % (tag, name)), "w"), indent=1)
        json.dump(obj, open(os.path.join(ROOT, "gpurun_out", name), "w"), indent=1)    # -> profiles/<name> (bench.py reads those)
    print(json.dumps({k: res.get(k) for k in ("wait_frac", "wait_inst_frac", "valu_issue_frac", "waves_per_simd", "lanes_per_valu")}))
    print(json.dumps({c: {k: v for k, v in e.items() if k in ("k_coverage", "k_search")} for c, e in traffic["configs"].items()}))


if __name__ == "__main__":
    main()
