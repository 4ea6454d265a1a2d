"""CPU-side checks of the product boundary: the C-ABI library loads without a GPU, exports every
symbol include/talc_hip.h declares, fails loudly (no CPU fallback) when asked to compute, and
its host-side table builder agrees with the oracle's table."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle_lib as O
import parity_util as PU
from talc_amd import lib as T
from talc_amd.synth import Synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "talc_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(talc_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no prototypes found in the header"
    L = T.lib()
    missing = [s for s in sorted(declared) if not hasattr(L, s)]
    assert not missing, missing
    assert set(T.ABI_SYMBOLS) == declared
    assert L.talc_abi_version() == 1


def test_params_default_match_reference_defaults():
    p = T.default_params()
    q = O.params()
    for f in PU.PARAM_FIELDS:
        assert getattr(p, f) == getattr(q, f), f
    # main.cpp:115-116,142-183 / Explorer.cpp:85-97 / Jellyfish.cpp:64 / Read.cpp:361
    assert (p.k, p.min_count, p.alpha, p.window_size, p.sr_error_rate) == (21, 2, 2.57, 9, 0.025)
    assert (p.min_inner_score, p.min_border_score, p.max_nb_competing_paths) == (0.7, 0.7, 7)
    assert (p.min_start_anchors, p.max_start_anchors, p.max_in_count) == (3, 5, 100000)
    assert (p.max_nb_border_paths, p.max_nb_inner_paths, p.check_interval) == (75, 50, 6)
    assert (p.allowed_failure_rate, p.max_nb_border_failures, p.coloured_count_thr, p.max_border_length) == (0.3, 3, 10000, 500)


def test_invalid_parameters_are_rejected():
    for kw in (dict(k=17), dict(k=32), dict(min_count=0), dict(coloured_count_thr=70000)):
        p = T.default_params(**kw)
        with pytest.raises(T.TalcError):
            T.Table.from_arrays(np.zeros(1, np.uint64), np.ones(1, np.uint32), p)


@pytest.mark.skipif(PU.T.device_count() > 0, reason="only meaningful on a host without a GPU")
def test_compute_fails_loudly_without_gpu():
    p = T.default_params()
    t = T.Table.from_arrays(np.arange(10, dtype=np.uint64), np.full(10, 5, np.uint32), p)
    with pytest.raises(T.TalcError):
        t.upload(0)
    with pytest.raises(T.TalcError):
        T.Context(t, p, 0)          # table not uploaded / no device: no silent CPU path
    with pytest.raises(T.TalcError):
        t.lookup(np.arange(3, dtype=np.uint64))


def test_host_table_equals_oracle_table_incl_duplicates_and_filter():
    rng = np.random.default_rng(3)
    p, q = PU.both_params(k=21)
    keys = rng.integers(0, 1 << 42, 50_000, dtype=np.uint64)
    counts = rng.integers(0, 40, 50_000).astype(np.uint32)       # some below MIN_COUNT
    # duplicates with different counts: the first line wins (std::map::insert, Jellyfish.cpp:262)
    keys = np.concatenate([keys, keys[:5000]])
    counts = np.concatenate([counts, (counts[:5000] + 7).astype(np.uint32)])
    ot = O.OracleTable(q, O.OracleTable.MAP)
    ot.insert_packed(keys, counts)
    tt = T.Table.from_arrays(keys, counts, p)
    assert len(ot) == len(tt)
    probe = np.concatenate([keys, rng.integers(0, 1 << 42, 20_000, dtype=np.uint64)])
    oc, oj = ot.lookup_packed(probe)
    tc, tj = tt.lookup_host(probe)
    assert (oc == tc).all() and (oj == tj).all()


def test_host_table_junction_colouring_and_decolouring():
    p, q = PU.both_params(k=21, use_junctions=1)
    S = Synth(target_kmers=60_000, k=21, seed=5)
    keys, counts = S.dump_arrays()
    jk, jc = S.junction_arrays()
    # add reverse-complement-only junction entries and homopolymers
    hom = np.array([0, (1 << 42) - 1, int("01" * 21, 2), int("10" * 21, 2)], dtype=np.uint64)
    keys = np.concatenate([keys, hom])
    counts = np.concatenate([counts, np.full(4, 50, np.uint32)])
    jk = np.concatenate([jk, hom])
    jc = np.concatenate([jc, np.full(4, 77, np.int64)])
    ot = O.OracleTable(q, O.OracleTable.MAP)
    ot.insert_packed(keys, counts)
    ot.colour_packed(jk, jc)
    ot.decolour()
    tt = T.Table.from_arrays(keys, counts, p)
    tt.colour(jk, jc)
    tt.decolour_repeats()

    def rc(km):
        r = 0
        for _ in range(21):
            r = (r << 2) | (3 - (km & 3))
            km >>= 2
        return r
    probe = np.concatenate([keys, jk, np.array([rc(int(x)) for x in jk[:2000]], dtype=np.uint64)])
    oc, oj = ot.lookup_packed(probe)
    tc, tj = tt.lookup_host(probe)
    assert (oc == tc).all() and (oj == tj).all()
    assert (oj > 0).sum() > 100                 # colours were applied
    assert (oj[len(keys) - 4:len(keys)] == 0).all()   # homopolymers un-coloured (utils.cpp:658-669)
    assert (oj < 10000).all()                   # colouredCountThr (Jellyfish.cpp:64,284)


def test_table_from_dump_files_matches_oracle(tmp_path):
    S = Synth(target_kmers=30_000, k=21, seed=6)
    dump, jd = str(tmp_path / "sr.dump"), str(tmp_path / "j.dump")
    S.write_dump(dump)
    S.write_junctions(jd)
    with open(dump, "a") as f:      # malformed / odd lines the reference tolerates
        f.write("ACGTNACGTACGTACGTACGT 9\n")    # N k-mer: never matchable by an ACGT read k-mer
        f.write("acgtacgtacgtacgtacgta 12\n")   # lower case is accepted by Dna5
        f.write("ACGT 5\n")                     # wrong length
        f.write("lonely\n")                     # one token: 'Error when building dBG...'
        f.write("\n")                           # blank line
        f.write("  GATTACAGATTACAGATTACA\t\t 41  trailing words\r\n")   # tabs, leading blanks, CRLF, extra tokens
        f.write("TTTTACAGATTACAGATTACA 7x\n")    # atoi stops at the first non-digit: 7
        f.write("CCCCACAGATTACAGATTACA +3\n")    # explicit sign
        f.write("AAAAACAGATTACAGATTACA 5")       # last line without a newline
    p, q = PU.both_params(k=21, use_junctions=1)
    ot = O.OracleTable(q, O.OracleTable.MAP)
    st_o = ot.build_from_files(dump, jd)
    tt = T.Table.from_files(dump, jd, p)
    assert tt.build_stats[0] == st_o[0] and tt.build_stats[1] == st_o[1]
    keys, counts = S.dump_arrays(release=False)
    lower = np.array([int("".join({"a": "00", "c": "01", "g": "10", "t": "11"}[c] for c in "acgtacgtacgtacgtacgta"), 2)], dtype=np.uint64)
    probe = np.concatenate([keys, lower])
    oc, oj = ot.lookup_packed(probe)
    tc, tj = tt.lookup_host(probe)
    assert (oc == tc).all() and (oj == tj).all()
    assert tc[-1] == 12
    def pk(t):
        return np.array([int("".join({"A": "00", "C": "01", "G": "10", "T": "11"}[c] for c in t), 2)], dtype=np.uint64)
    for text, want in (("GATTACAGATTACAGATTACA", 41), ("TTTTACAGATTACAGATTACA", 7), ("CCCCACAGATTACAGATTACA", 3),
                       ("AAAAACAGATTACAGATTACA", 5)):
        assert tt.lookup_host(pk(text))[0][0] == want == ot.lookup_packed(pk(text))[0][0], text
    # the oracle's map also holds the two unmatchable keys (N k-mer, short k-mer)
    assert len(ot) == len(tt) + 2


def test_synth_is_deterministic_and_sharded():
    a = Synth(target_kmers=40_000, k=21, seed=9)
    b = Synth(target_kmers=40_000, k=21, seed=9)
    ka, ca = a.dump_arrays()
    kb, cb = b.dump_arrays()
    assert (ka == kb).all() and (ca == cb).all()
    ba, oa = a.reads(0, 50)
    b1, o1 = b.reads(0, 20)
    b2, o2 = b.reads(20, 30)        # any rank can generate its own shard
    assert bytes(ba) == bytes(b1) + bytes(b2)
    assert (ca == 1).sum() > 0 and (ca >= 2).sum() > 30_000
    lens = np.diff(oa.astype(np.int64))
    assert lens.max() < 6000 and lens.mean() > 500


def test_bench_workload_is_derived_from_its_arguments():
    """bench.py names the workload it really ran: N = 1 is BASELINE config 2, N > 1 config 3 (the same 1 M reads for
    every N: strong scaling), --config forces one, and any overridden figure turns the label into "custom"."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    w1 = bench.resolve_workload(bench.parse([]), 1)
    assert (w1["config"], w1["reads"], w1["kmers"], w1["k"]) == (2, 100_000, 50_000_000, 21) and w1["label"].startswith("config2:")
    for n in (2, 4, 8):
        w = bench.resolve_workload(bench.parse(["--gpus", str(n)]), n)
        assert (w["config"], w["reads"], w["kmers"], w["k"], w["junctions"]) == (3, 1_000_000, 200_000_000, 21, False)
        assert w["label"].startswith("config3:") and "over %d GPU" % n in w["label"]
    w4 = bench.resolve_workload(bench.parse(["--config", "4"]), 1)
    assert w4["junctions"] and w4["label"].startswith("config4:") and "junction" in w4["label"]
    w5 = bench.resolve_workload(bench.parse(["--config", "5"]), 1)
    assert (w5["k"], w5["kmers"], w5["mixed"]) == (31, 500_000_000, True) and "20 kb" in w5["label"]
    wk = bench.resolve_workload(bench.parse(["--k", "31", "--reads", "5000"]), 1)
    assert wk["label"].startswith("custom (config2 with reads, k changed)") and wk["k"] == 31 and wk["reads"] == 5000
    assert "config2:" not in wk["label"]
