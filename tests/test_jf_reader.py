"""The native reader of a Jellyfish 2 count file (talc_amd/csrc/talc_jf.h) behind talc_table_build: a .jf
given as -SR / -j builds the table its text dump builds (the counts `jellyfish query` would have answered one k-mer at a
time, Jellyfish.cpp:323-379, 415-467), and anything that does not verify is refused, not guessed at.
The files come from tests/jf_writer.py — the same recollected layout; parity with the real tool is unpinned."""
import numpy as np
import pytest

import jf_writer as JW
import oracle_lib as O
import parity_util as PU
from talc_amd import lib as T
from talc_amd.synth import Synth


@pytest.fixture(scope="module")
def files(tmp_path_factory):
    d = tmp_path_factory.mktemp("jf")
    S = Synth(target_kmers=40_000, k=21, seed=16)
    S.write_dump(str(d / "sr.dump"))
    S.write_junctions(str(d / "j.dump"))
    n = JW.dump_to_jf(str(d / "sr.dump"), str(d / "sr.jf"), 21, seed=3)
    JW.dump_to_jf(str(d / "j.dump"), str(d / "j.jf"), 21, seed=4, counter_len=2)
    return d, S, n


def test_table_from_jf_equals_table_from_its_text_dump_and_the_oracle(files):
    d, S, n = files
    p, q = PU.both_params(k=21, use_junctions=1)
    a = T.Table.from_files(str(d / "sr.dump"), str(d / "j.dump"), p)
    b = T.Table.from_files(str(d / "sr.jf"), str(d / "j.jf"), p)
    c = T.Table.from_files(str(d / "sr.jf"), str(d / "j.dump"), p)      # the two kinds mix
    assert len(a) == len(b) == len(c) and len(a) > 0
    assert b.build_stats[0] == n <= a.build_stats[0] and 0 < b.build_stats[1] <= a.build_stats[1] and b.build_stats[2] == 0
    ot = O.OracleTable(q, O.OracleTable.MAP)
    ot.build_from_files(str(d / "sr.dump"), str(d / "j.dump"))
    keys, _ = S.dump_arrays(release=False)
    rng = np.random.default_rng(5)
    probe = np.concatenate([keys, rng.integers(0, 1 << 42, 20_000, dtype=np.uint64)])
    oc, oj = ot.lookup_packed(probe)
    for t in (a, b, c):
        tc, tj = t.lookup_host(probe)
        assert (tc == oc).all() and (tj == oj).all()
    assert (oj > 0).any() and (oc == 1).sum() == 0          # colours present; the MIN_COUNT filter applied to the .jf too


def test_jf_record_widths_and_count_clamp(tmp_path):
    """k = 31 (62 bits in 8 bytes), k = 18 (36 bits in 5), counts of 1, 2, 4 and 8 bytes."""
    for k, cl in ((31, 8), (18, 1), (24, 2), (21, 4)):
        rng = np.random.default_rng(k)
        kms = np.unique(rng.integers(0, 1 << (2 * k), 3000, dtype=np.uint64))
        top = (1 << (8 * cl)) - 1
        cnt = rng.integers(2, min(top, 1 << 20) + 1, len(kms)).tolist()
        cnt[0] = top                                            # the widest count of this width
        path = str(tmp_path / ("w%d.jf" % k))
        JW.write_jf(path, zip(kms.tolist(), cnt), k, counter_len=cl)
        p = T.default_params(k=k)
        t = T.Table.from_files(path, None, p)
        assert len(t) == len(kms)
        got = t.lookup_host(kms)[0]
        assert got.tolist() == [min(c, 0x7fffffff) for c in cnt]


def test_jf_that_does_not_verify_is_refused(tmp_path):
    p = T.default_params(k=21)
    ent = [("ACGTACGTACGTACGTACGTA", 5), ("GATTACAGATTACAGATTACA", 9)]

    def refused(path, word):
        with pytest.raises(T.TalcError) as e:
            T.Table.from_files(path, None, p)
        assert word in str(e.value), str(e.value)

    ok = str(tmp_path / "ok.jf")
    JW.write_jf(ok, ent, 21)
    assert len(T.Table.from_files(ok, None, p)) == 2
    f = str(tmp_path / "k25.jf")
    JW.write_jf(f, [("ACGTACGTACGTACGTACGTACGTA", 5)], 25)
    refused(f, "25-mers")
    f = str(tmp_path / "bloom.jf")
    JW.write_jf(f, ent, 21, fmt="bloomcounter")
    refused(f, "bloomcounter")
    f = str(tmp_path / "text.jf")
    JW.write_jf(f, ent, 21, fmt="text/sorted")
    refused(f, "text/sorted")
    raw = open(ok, "rb").read()
    f = str(tmp_path / "cut.jf")
    open(f, "wb").write(raw[:-3])
    refused(f, "whole number")
    f = str(tmp_path / "pad.jf")
    body = bytearray(raw)
    body[-5] |= 0x80                                            # a bit above the key's 42 in its sixth byte
    open(f, "wb").write(bytes(body))
    refused(f, "padding bits")
    f = str(tmp_path / "zero.jf")
    JW.write_jf(f, [("ACGTACGTACGTACGTACGTA", 0)], 21)
    refused(f, "zero count")
    f = str(tmp_path / "long.jf")
    open(f, "wb").write(b"%09d" % (len(raw) + 100) + raw[9:])
    refused(f, "longer than the file")
    # header only: an empty table, like an empty dump
    f = str(tmp_path / "empty.jf")
    JW.write_jf(f, [], 21)
    assert len(T.Table.from_files(f, None, p)) == 0
