#!/usr/bin/env python3
"""k_coverage time against the presence filter's size (TALC_FILTER_BITS bits per stored k-mer), one synthetic dump, the
table rebuilt per setting (the filter is made at upload).
    python tools/cov_filter_sweep.py --kmers 200000000 --reads 200000 --bits 8,10,12,16,20"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from talc_amd import lib as T  # noqa: E402
from talc_amd.synth import Synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--kmers", type=int, default=200_000_000)
ap.add_argument("--reads", type=int, default=200_000)
ap.add_argument("--k", type=int, default=21)
ap.add_argument("--bits", default="8,10,12,16,20")
ap.add_argument("--reps", type=int, default=7)
ap.add_argument("--mixed", action="store_true")
ap.add_argument("--slots", default="20", help="table slots per stored k-mer x 10 (TALC_TABLE_SLOTS_X10), comma list")
a = ap.parse_args()
S = Synth(target_kmers=a.kmers, k=a.k, seed=0, mixed_lengths=int(a.mixed))
keys, counts = S.dump_arrays()
bases, offs = S.reads(0, a.reads)
p = T.default_params(k=a.k)
for slots, bits in [(sl, int(x)) for sl in a.slots.split(",") for x in a.bits.split(",")]:
    os.environ["TALC_FILTER_BITS"] = str(bits)
    os.environ["TALC_TABLE_SLOTS_X10"] = slots
    tab = T.Table.from_arrays(keys, counts, p, device=0)
    tab.decolour_repeats()
    tab.upload(0)
    ctx = T.Context(tab, p, 0)
    b = ctx.batch(bases, offs)
    ts = []
    for _ in range(a.reps):
        b.coverage()
        ts.append(ctx.timing().coverage_ms)
    nk = b.n_kmers
    med = float(np.median(ts))
    print("slots x10 %s filter %2d bits/k-mer (%.0f MB): coverage_ms min %.3f median %.3f  -> %.1f%% of 8 TB/s at 25 B/k-mer (%d k-mers, table %d)" %
          (slots, bits, len(tab) * bits / 8e6, min(ts), med, 100.0 * nk * 25 / (med * 1e-3) / 8e12, nk, len(tab)), flush=True)
    b.close(); ctx.close(); tab.close()
