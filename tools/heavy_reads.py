#!/usr/bin/env python3
"""Development tool: the HIP path alone on a stress set (tests/stress_cases.py), with the -DTALC_PROF build's category
profile and per-read table — which reads of the set are the heavy ones and where their time goes.
    TALC_LIB=talc_amd/_build/libtalc_hip_prof.so TALC_PROF_PRINT=1 TALC_PROF_READS=out.tsv python tools/heavy_reads.py 105 6000"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
from stress_cases import CASES  # noqa: E402
from talc_amd import lib as T  # noqa: E402
from talc_amd.synth import Synth  # noqa: E402

case, n = int(sys.argv[1]), int(sys.argv[2])
kw, pkw = CASES[case]
S = Synth(target_kmers=kw["target_kmers"], k=kw["k"], seed=kw["seed"], **kw.get("synth_kw", {}))
keys, counts = S.dump_arrays()
p = T.default_params(k=kw["k"], **pkw)
tab = T.Table.from_arrays(keys, counts, p, device=0)
tab.decolour_repeats()
tab.upload(0)
ctx = T.Context(tab, p, 0)
bases, offs = S.reads(0, n)
b = ctx.batch(bases, offs)
t0 = time.perf_counter()
b.correct()
print("case %d: %d reads corrected in %.2f s (search %.1f ms, retry %.1f ms, retried %d)" %
      (case, n, time.perf_counter() - t0, ctx.timing().search_ms, ctx.timing().retry_ms, ctx.timing().n_retried), flush=True)
