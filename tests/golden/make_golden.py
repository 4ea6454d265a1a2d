#!/usr/bin/env python3
"""Generates the golden fixtures in this directory FROM THE ORACLE (the reference ships no
golden vectors — SURVEY.md §4 — and cannot be built here — §8c — so these pin the oracle's
behaviour, "parity unpinned" with respect to a real TALC binary).

Each fixture is data only: the synthetic-input recipe (generator seed + sha256 of the k-mer
dump arrays it must reproduce), the reads as text, the parameters, and the expected outputs
(corrected sequences, per-read status, sha256 of the coverage vectors).  g1 additionally
carries the full dump as text so one case does not depend on the generator at all.

usage: python tests/golden/make_golden.py        (rewrites tests/golden/*.json.gz)
"""
import gzip
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import oracle_lib as O  # noqa: E402
from talc_amd.synth import Synth  # noqa: E402

D = "ACGT"


def unpack(km, k):
    return "".join(D[(int(km) >> (2 * (k - 1 - i))) & 3] for i in range(k))


CASES = {
    "g1_default_k21": dict(k=21, kmers=20_000, seed=101, reads=16, embed_dump=True),
    "g2_junctions_k21": dict(k=21, kmers=60_000, seed=102, reads=24, junctions=True),
    "g3_reverse_k21": dict(k=21, kmers=60_000, seed=103, reads=24, params=dict(reverse=1)),
    "g4_k18": dict(k=18, kmers=60_000, seed=104, reads=20),
    "g5_k30": dict(k=30, kmers=60_000, seed=105, reads=20),
    "g6_k31": dict(k=31, kmers=60_000, seed=106, reads=20),
    "g7_params": dict(k=21, kmers=60_000, seed=107, reads=20,
                      params=dict(min_count=3, window_size=6, max_nb_competing_paths=5, alpha=1.96,
                                  sr_error_rate=0.05, min_inner_score=0.5, min_border_score=0.6)),
    "g8_edge_inputs": dict(k=21, kmers=60_000, seed=108, reads=12, edge=True),
    # paralog families: forks and bubbles in the graph -> several live Trails, gardening, bridge scoring, cycles
    "g9_paralogs_k21": dict(k=21, kmers=80_000, seed=109, reads=24, synth=dict(paralog_frac=0.5, paralog_div=0.03)),
    "g10_paralogs_junctions_maxb4": dict(k=21, kmers=80_000, seed=110, reads=20, junctions=True,
                                         synth=dict(paralog_frac=0.6, paralog_div=0.015),
                                         params=dict(max_nb_competing_paths=4, window_size=7)),
}


def build_case(name, c):
    k = c["k"]
    S = Synth(target_kmers=c["kmers"], k=k, seed=c["seed"], **c.get("synth", {}))
    keys, counts = S.dump_arrays()
    pk = dict(k=k, use_junctions=int(bool(c.get("junctions"))))
    pk.update(c.get("params", {}))
    q = O.params(**pk)
    tab = O.OracleTable(q, O.OracleTable.MAP)
    tab.insert_packed(keys, counts)
    jsha = None
    if c.get("junctions"):
        jk, jc = S.junction_arrays()
        tab.colour_packed(jk, jc)
        jsha = hashlib.sha256(jk.tobytes() + jc.tobytes()).hexdigest()
    tab.decolour()
    bases, offs = S.reads(0, c["reads"])
    reads = [bytes(bases[int(offs[i]):int(offs[i + 1])]).decode() for i in range(c["reads"])]
    if pk.get("reverse"):
        # the table is forward-strand only (k-mers are directional, main.cpp:89): feed reads from
        # the opposite strand so that -rev brings them back onto the table's strand
        comp0 = str.maketrans("ACGTN", "TGCAN")
        reads = [x.translate(comp0)[::-1] for x in reads]
    if c.get("edge"):
        r = reads
        reads = [
            "",                                   # empty read
            r[0][:k],                             # exactly K bases: skipped (main.cpp:262)
            r[0][:k + 1],                         # K+1: two k-mers
            r[1].lower(),                         # lower case accepted by Dna5
            r[2][:400] + "N" + r[2][400:],        # one N
            r[3][:300] + "NNNNNNNNNN" + r[3][300:900] + "RYKM" + r[3][900:],   # IUPAC -> N
            "ACGT" * 300,                         # low-complexity, probably no solid k-mer
            "A" * 500,                            # homopolymer
        ] + r[4:8]
    rb = "".join(reads).encode()
    roffs = np.zeros(len(reads) + 1, dtype=np.uint64)
    roffs[1:] = np.cumsum([len(x) for x in reads])
    out, oo, st = tab.correct_batch(np.frombuffer(rb, dtype=np.uint8) if rb else np.zeros(0, np.uint8), roffs, nthreads=8)
    corrected = [bytes(out[int(oo[i]):int(oo[i + 1])]).decode() for i in range(len(reads))]
    h = hashlib.sha256()
    comp = str.maketrans("ACGTN", "TGCAN")
    for s in reads:
        if pk.get("reverse"):
            s = "".join(ch if ch in "ACGT" else "N" for ch in s.upper()).translate(comp)[::-1]
        cov, jc_, nin = tab.coverage(s)
        h.update(cov.tobytes()); h.update(jc_.tobytes())
    fx = {
        "name": name, "params": pk, "synth": dict({"target_kmers": c["kmers"], "k": k, "seed": c["seed"]}, **c.get("synth", {})),
        "dump_sha256": hashlib.sha256(keys.tobytes() + counts.tobytes()).hexdigest(),
        "junction_sha256": jsha, "table_size": len(tab),
        "reads": reads, "expected": corrected, "status": [int(x) for x in st], "coverage_sha256": h.hexdigest(),
    }
    if c.get("embed_dump"):
        fx["dump_text"] = "".join("%s %d\n" % (unpack(keys[i], k), counts[i]) for i in range(len(keys)))
    with gzip.GzipFile(os.path.join(HERE, name + ".json.gz"), "wb", mtime=0) as f:
        f.write(json.dumps(fx, sort_keys=True).encode())
    print(name, "reads", len(reads), "status", np.bincount(st, minlength=4).tolist(), "ub", O.ub_counters())


if __name__ == "__main__":
    only = sys.argv[1:]   # optional: names of the cases to (re)generate
    for n, c in CASES.items():
        if not only or n in only:
            build_case(n, c)
