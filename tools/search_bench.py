#!/usr/bin/env python3
"""Same-box A/B helper: for the library TALC_LIB names, the kernel times of one full correction of BASELINE config 2
(100 k reads, 54 M k-mers) and of the branching side workload (20 k reads, 60 % paralogs, K = 25), each with a hash of
the corrected records (two builds that disagree on a hash disagree on a record).
    python tools/search_bench.py [--reps R] [--no-paralog] [--reads N] [--kmers N]"""
import argparse
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from talc_amd import lib as T  # noqa: E402
from talc_amd.synth import Synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--kmers", type=int, default=50_000_000)
ap.add_argument("--reads", type=int, default=100_000)
ap.add_argument("--k", type=int, default=21)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--no-paralog", action="store_true")
ap.add_argument("--no-main", action="store_true")
ap.add_argument("--mixed", action="store_true", help="config 5's read lengths (500 b - 20 kb log-uniform)")
a = ap.parse_args()
name = os.path.basename(os.environ.get("TALC_LIB", "libtalc_hip.so"))


def run(S, k, n, label):
    keys, counts = S.dump_arrays()
    p = T.default_params(k=k)
    tab = T.Table.from_arrays(keys, counts, p, device=0)
    tab.decolour_repeats()
    tab.upload(0)
    ctx = T.Context(tab, p, 0)
    bases, offs = S.reads(0, n)
    b = ctx.batch(bases, offs)
    b.correct()
    ts, walls = [], []
    for _ in range(a.reps):
        t0 = time.perf_counter()
        b.correct()
        walls.append(1e3 * (time.perf_counter() - t0))
        t = ctx.timing()
        ts.append((t.coverage_ms, t.structure_ms, t.search_ms + t.retry_ms))
    out, oo, st = b.fetch_corrected()
    h = hashlib.sha256(out.tobytes() + oo.tobytes() + st.tobytes()).hexdigest()[:12]
    ts = np.array(ts)
    print("lib=%s %s: coverage %.3f structure %.3f search %.2f (min %.2f) step %.2f ms  retried %d  sha %s" %
          (name, label, np.median(ts[:, 0]), np.median(ts[:, 1]), np.median(ts[:, 2]), ts[:, 2].min(), float(np.median(walls)),
           ctx.timing().n_retried, h), flush=True)
    b.close(); ctx.close(); tab.close()


if not a.no_main:
    run(Synth(target_kmers=a.kmers, k=a.k, seed=0, mixed_lengths=int(a.mixed)), a.k, a.reads,
        "config2" if (a.kmers, a.reads, a.k, a.mixed) == (50_000_000, 100_000, 21, False) else "custom(k=%d,kmers=%d,reads=%d%s)" % (a.k, a.kmers, a.reads, ",mixed" if a.mixed else ""))
if not a.no_paralog:
    run(Synth(target_kmers=2_000_000, k=25, seed=77, paralog_frac=0.6, paralog_div=0.05), 25, 20_000, "paralog")
