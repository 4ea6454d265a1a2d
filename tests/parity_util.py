"""Shared helpers for the parity tests: build the same table on both sides (oracle = checker,
libtalc_hip = product) from one synthetic spec and compare the two on seeded reads."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import oracle_lib as O  # noqa: E402
from talc_amd import lib as T  # noqa: E402
from talc_amd.synth import Synth  # noqa: E402

PARAM_FIELDS = [f for f, _ in T.Params._fields_]


def both_params(**kw):
    """(product Params, oracle OrcParams) with identical field values."""
    p = T.default_params(**kw)
    q = O.params(**{f: getattr(p, f) for f in PARAM_FIELDS})
    return p, q


class Pair:
    """Oracle table + product table (uploaded) from one synthetic transcriptome."""

    def __init__(self, target_kmers=300_000, k=21, seed=1, junctions=False, oracle_backend=O.OracleTable.FLAT,
                 synth_kw=None, **params_kw):
        self.synth = Synth(target_kmers=target_kmers, k=k, seed=seed, **(synth_kw or {}))
        self.p, self.q = both_params(k=k, use_junctions=int(junctions), **params_kw)
        keys, counts = self.synth.dump_arrays()
        self.keys, self.counts = keys, counts
        self.otab = O.OracleTable(self.q, oracle_backend)
        self.otab.insert_packed(keys, counts)
        self.ttab = T.Table.from_arrays(keys, counts, self.p)
        if junctions:
            jk, jc = self.synth.junction_arrays()
            self.otab.colour_packed(jk, jc)
            self.ttab.colour(jk, jc)
        self.otab.decolour()
        self.ttab.decolour_repeats()
        self.ctx = None

    def upload(self, device=0):
        self.ttab.upload(device)
        self.ctx = T.Context(self.ttab, self.p, device)
        return self.ctx

    def reads(self, first, n):
        return self.synth.reads(first, n)


class CustomPair:
    """Oracle table + product table from explicit transcripts (forward-strand k-mers, `depth` per occurrence): for graph
    shapes the synthetic generator does not make (tandem repeats -> cycles)."""

    def __init__(self, transcripts, k=21, depth=20, **params_kw):
        self.p, self.q = both_params(k=k, **params_kw)
        code = {"A": 0, "C": 1, "G": 2, "T": 3}
        cnt = {}
        mask = (1 << (2 * k)) - 1
        for t in transcripts:
            v = 0
            for i, ch in enumerate(t):
                v = ((v << 2) | code[ch]) & mask
                if i >= k - 1:
                    cnt[v] = cnt.get(v, 0) + depth
        self.keys = np.fromiter(cnt.keys(), dtype=np.uint64, count=len(cnt))
        self.counts = np.fromiter(cnt.values(), dtype=np.uint32, count=len(cnt))
        self.otab = O.OracleTable(self.q, O.OracleTable.FLAT)
        self.otab.insert_packed(self.keys, self.counts)
        self.ttab = T.Table.from_arrays(self.keys, self.counts, self.p)
        self.otab.decolour()
        self.ttab.decolour_repeats()
        self.ctx = None

    def upload(self, device=0):
        self.ttab.upload(device)
        self.ctx = T.Context(self.ttab, self.p, device)
        return self.ctx


def seqs_of(buf, offs):
    b = bytes(buf)
    return [b[int(offs[i]):int(offs[i + 1])].decode() for i in range(len(offs) - 1)]


def compare_correction(pair, bases, offs, nthreads=8, verbose=True):
    """Run both sides; returns the list of read indices that differ (sequence or status)."""
    import time
    t0 = time.time()
    o_out, o_off, o_st = pair.otab.correct_batch(bases, offs, nthreads=nthreads)
    t1 = time.time()
    g_out, g_off, g_st = pair.ctx.correct(bases, offs)
    pair.last_times = (t1 - t0, time.time() - t1)   # (oracle seconds, HIP path seconds incl. transfers)
    so, sg = seqs_of(o_out, o_off), seqs_of(g_out, g_off)
    bad = [i for i in range(len(so)) if so[i] != sg[i] or int(o_st[i]) != int(g_st[i])]
    if verbose:
        print("reads %d  status(oracle) %s  mismatches %d" % (len(so), np.bincount(o_st, minlength=5).tolist(), len(bad)))
    return bad, (so, o_st), (sg, g_st)


def first_trace_diff(pair, bases, offs, idx):
    """Textual traces of read idx on both sides and the first differing line."""
    seq = bytes(bases[int(offs[idx]):int(offs[idx + 1])]).decode()
    to = pair.otab.trace(seq, steps=bool(os.environ.get("TALC_TRACE_STEPS"))).splitlines()
    b = pair.ctx.batch(np.frombuffer(seq.encode(), dtype=np.uint8), np.array([0, len(seq)], dtype=np.uint64))
    tg = b.trace(0).splitlines()
    b.close()
    n = min(len(to), len(tg))
    for i in range(n):
        if to[i] != tg[i]:
            return i, to[max(0, i - 3):i + 2], tg[max(0, i - 3):i + 2]
    if len(to) != len(tg):
        return n, to[n - 2:n + 2], tg[n - 2:n + 2]
    return None
