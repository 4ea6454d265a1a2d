"""Developer script (run on the GPU box): staged parity check product vs oracle with diagnostics.
usage: python tests/gpu_dev_check.py [n_reads] [target_kmers] [seed]"""
import sys
import time

import numpy as np

import parity_util as PU

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 64
target = int(sys.argv[2]) if len(sys.argv) > 2 else 300_000
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
kw = {}
if len(sys.argv) > 4 and sys.argv[4] == "rev":
    kw["reverse"] = 1

t0 = time.time()
pair = PU.Pair(target_kmers=target, seed=seed, junctions=(len(sys.argv) > 5), **kw)
print("tables: oracle %d  product %d  (%.1fs)" % (len(pair.otab), len(pair.ttab), time.time() - t0), flush=True)
assert len(pair.otab) == len(pair.ttab)
pair.upload(0)

# 1. point lookups + successors
rng = np.random.default_rng(seed)
q = np.concatenate([pair.keys[:2000], rng.integers(0, 1 << 42, 2000, dtype=np.uint64)])
oc, oj = pair.otab.lookup_packed(q)
gc, gj = pair.ttab.lookup(q)
assert (oc == gc).all() and (oj == gj).all(), "lookup mismatch"
print("lookup ok", flush=True)

# 2. coverage
bases, offs = pair.reads(0, n_reads)
b = pair.ctx.batch(bases, offs)
b.coverage()
c, j, ko, nin = b.fetch_coverage()
seqs = PU.seqs_of(bases, offs)
COMP = str.maketrans("ACGTN", "TGCAN")
for i, s in enumerate(seqs):
    if kw.get("reverse"):
        s = s.upper().translate(COMP)[::-1]
    oc, oj, onin = pair.otab.coverage(s)
    gc_ = c[int(ko[i]):int(ko[i + 1])]
    if len(s) >= pair.p.k:
        assert len(oc) == len(gc_), (i, len(oc), len(gc_))
        assert (oc == gc_).all(), ("coverage mismatch read", i)
        assert (oj == j[int(ko[i]):int(ko[i + 1])]).all()
        assert onin == nin[i], (i, onin, nin[i])
print("coverage ok; timing", pair.ctx.timing().as_dict(), flush=True)
b.close()

# 3. correction
t0 = time.time()
bad, (so, ost), (sg, gst) = PU.compare_correction(pair, bases, offs)
print("correction compare took %.1fs; timing %s" % (time.time() - t0, pair.ctx.timing().as_dict()), flush=True)
for i in bad[:5]:
    print("READ", i, "status oracle", ost[i], "gpu", gst[i], "len", len(so[i]), len(sg[i]))
    d = PU.first_trace_diff(pair, bases, offs, i)
    print("  first trace diff:", d)
print("PARITY", "OK" if not bad else "FAIL %d/%d" % (len(bad), len(so)))
sys.exit(1 if bad else 0)
