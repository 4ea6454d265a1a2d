#!/usr/bin/env python3
"""Times k_coverage / k_structure alone on BASELINE config 2 (or --kmers / --reads), for the library TALC_LIB names.
    python tools/cov_bench.py [--kmers N] [--reads N] [--k K] [--reps R]
Prints one line: coverage ms (min / median), structure ms, search ms of one full correction."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from talc_amd import lib as T  # noqa: E402
from talc_amd.synth import Synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--kmers", type=int, default=50_000_000)
ap.add_argument("--reads", type=int, default=100_000)
ap.add_argument("--k", type=int, default=21)
ap.add_argument("--reps", type=int, default=7)
ap.add_argument("--full", action="store_true", help="also one full correction (structure / search times)")
a = ap.parse_args()
S = Synth(target_kmers=a.kmers, k=a.k, seed=0)
keys, counts = S.dump_arrays()
p = T.default_params(k=a.k)
tab = T.Table.from_arrays(keys, counts, p, device=0)
tab.decolour_repeats()
tab.upload(0)
ctx = T.Context(tab, p, 0)
bases, offs = S.reads(0, a.reads)
b = ctx.batch(bases, offs)
ts = []
for _ in range(a.reps):
    b.coverage()
    ts.append(ctx.timing().coverage_ms)
line = "lib=%s coverage_ms min %.3f median %.3f" % (os.path.basename(os.environ.get("TALC_LIB", "libtalc_hip.so")), min(ts), float(np.median(ts)))
if a.full:
    b.correct(); b.correct()
    t = ctx.timing()
    line += " | structure %.3f search %.3f" % (t.structure_ms, t.search_ms)
print(line, flush=True)
