"""-m gpu: BASELINE.json configs at FULL size — too big for an exhaustive oracle run, so they are checked through
size-independent properties plus an oracle spot-check on a random sample of the very same reads:
  config 2   100 k reads (~2 kb), 50 M-entry k=21 dump
  config 3   1 M reads, 200 M-entry k=21 dump
  config 4   the same with --junctions (junction colours: the dual-value probe path)
  config 5   k=31, 500 M-entry dump, 100 k reads of 500 b - 20 kb
Each case prints one line with the table size, whether the walk tables were built, n_retried / n_failed and the
kernel times, so the GPU log of the round records them."""
import os
import time

import numpy as np
import pytest

import oracle_lib as O
import parity_util as PU
from talc_amd import lib as T
from talc_amd.synth import Synth

pytestmark = pytest.mark.gpu


def _sample_check(otab, bases, offs, out, oo, st, idx, nthreads=16):
    """The oracle on reads `idx` of the batch against the records the GPU produced for them."""
    seq_in = bases.tobytes() if hasattr(bases, "tobytes") else bytes(bases)
    reads = [seq_in[int(offs[i]):int(offs[i + 1])] for i in idx]
    sb = np.frombuffer(b"".join(reads), dtype=np.uint8)
    so = np.zeros(len(reads) + 1, dtype=np.uint64)
    so[1:] = np.cumsum([len(x) for x in reads])
    e_out, e_off, e_st = otab.correct_batch(sb, so, nthreads=nthreads)
    exp = PU.seqs_of(e_out, e_off)
    seq_out = out.tobytes()
    for j, i in enumerate(idx):
        assert seq_out[int(oo[i]):int(oo[i + 1])].decode() == exp[j], int(i)
        assert int(st[i]) == int(e_st[j]), int(i)


def _fullsize_case(name, n_kmers, n_reads, k, junctions, synth_kw, n_spot, min_corrected, check_halves=True):
    t0 = time.time()
    S = Synth(target_kmers=n_kmers, k=k, seed=0, **synth_kw)
    keys, counts = S.dump_arrays()
    p, q = PU.both_params(k=k, use_junctions=int(junctions))
    tab = T.Table.from_arrays(keys, counts, p, device=0)          # insert loop of buildCDBG on the GPU
    otab = O.OracleTable(q, O.OracleTable.FLAT)
    otab.insert_packed(keys, counts)
    if junctions:
        jk, jc = S.junction_arrays()
        tab.colour(jk, jc)
        otab.colour_packed(jk, jc)
    tab.decolour_repeats()
    otab.decolour()
    assert len(otab) == len(tab)
    t_tables = time.time() - t0
    # the table itself against the oracle's: stored k-mers (coloured ones included), absent ones, both through the
    # host image and on the device
    rng = np.random.default_rng(7)
    probe = [keys[rng.integers(0, len(keys), 200_000)], rng.integers(0, 1 << (2 * k), 100_000, dtype=np.uint64)]
    if junctions:
        probe.append(jk[rng.integers(0, len(jk), 100_000)])
    probe = np.concatenate(probe)
    oc, oj = otab.lookup_packed(probe)
    hc, hj = tab.lookup_host(probe)
    assert (oc == hc).all() and (oj == hj).all()
    if junctions:
        assert int((oj > 0).sum()) > 10_000
    del keys, counts
    tab.upload(0)
    gc, gj = tab.lookup(probe)
    assert (oc == gc).all() and (oj == gj).all()
    walk = tab.device_bytes > 1.5 * 2 * 32 * (2 * len(tab))      # (two bucket tables at load 0.5 are 128 B per k-mer, the walk tables as much again)
    ctx = T.Context(tab, p, 0)
    bases, offs = S.reads(0, n_reads)
    t1 = time.time()
    b = ctx.batch(bases, offs)
    rc = b.correct()
    out, oo, st = b.fetch_corrected()
    tm = ctx.timing()
    t_gpu = time.time() - t1
    print("\n[fullsize %s] %d k-mers kept (k=%d%s), table %.1f GB on device, walk tables %s, %d reads / %d bases "
          "(longest %d), n_retried %d, n_failed %d, coverage %.2f ms, structure %.2f ms, search %.2f ms, retry %.2f ms, "
          "tables built in %.0f s, batch create+correct+fetch %.2f s"
          % (name, len(tab), k, ", junction colours" if junctions else "", tab.device_bytes / 1e9,
             "built" if walk else "not built", n_reads, len(bases), int(np.diff(offs.astype(np.int64)).max()),
             tm.n_retried, tm.n_failed, tm.coverage_ms, tm.structure_ms, tm.search_ms, tm.retry_ms, t_tables, t_gpu), flush=True)
    assert rc == 0 and tm.n_failed == 0
    hist = np.bincount(st, minlength=5)
    assert hist.sum() == n_reads and hist[0] > min_corrected * n_reads and hist[4] == 0
    # records are over the Dna5 alphabet and their total size is plausible (correction does not change the length by
    # more than a few percent overall)
    assert set(np.unique(out).tolist()) <= set(b"ACGTN")
    assert 0.9 * len(bases) < len(out) < 1.1 * len(bases)
    # pass-through rule (main.cpp:310): reads that were not corrected come back unchanged
    for i in np.nonzero(st != 0)[0][:200]:
        assert np.array_equal(out[int(oo[i]):int(oo[i + 1])], bases[int(offs[i]):int(offs[i + 1])])
    # determinism: a second run of the same batch is bit-identical
    b.correct()
    o3, oo3, st3 = b.fetch_corrected()
    assert np.array_equal(o3, out) and np.array_equal(oo3, oo) and np.array_equal(st3, st)
    b.close()
    if check_halves:
        # batch-composition independence: two halves give the same records as the whole batch
        h = n_reads // 2
        o1, oo1, st1 = ctx.correct(bases[: int(offs[h])], offs[: h + 1].copy())
        o2, oo2, st2 = ctx.correct(bases[int(offs[h]):], (offs[h:] - offs[h]).copy())
        assert np.array_equal(np.concatenate([o1, o2]), out)
        assert np.array_equal(np.concatenate([st1, st2]), st)
    # oracle spot-check on a random sample of the same reads, the longest reads of the batch among them
    lens = np.diff(offs.astype(np.int64))
    idx = set(rng.choice(n_reads, n_spot, replace=False).tolist())
    idx |= set(np.argsort(lens)[-max(8, n_spot // 10):].tolist())
    _sample_check(otab, bases, offs, out, oo, st, sorted(idx))
    ctx.close()
    tab.close()
    otab.close()
    return tm


def test_config2_full_size_properties():
    _fullsize_case("config2", 50_000_000, 100_000, 21, False, {}, 400, 0.97)


def test_config3_full_size_one_million_reads():
    """BASELINE config 3 as it stands: 1 M reads against the 200 M-entry table, no colours (on one GPU here; over N GPUs
    the same reads are dealt in chunks, bench.py --gpus N)."""
    _fullsize_case("config3", 200_000_000, 1_000_000, 21, False, {}, 300, 0.97, check_halves=False)


def test_config4_full_size_junctions():
    """1 M reads against the 200 M-entry table with junction colours (the dual-table probe path of config 4; config 3 is
    this table without the colours)."""
    _fullsize_case("config4", 200_000_000, 1_000_000, 21, True, {}, 300, 0.97, check_halves=False)


def test_config5_full_size_k31_mixed_lengths():
    """K=31, 500 M-entry table (2 x 35 GB of buckets: offsets beyond 4 GB everywhere), reads from 500 b to 20 kb."""
    tm = _fullsize_case("config5", 500_000_000, 100_000, 31, False, dict(mixed_lengths=1), 150, 0.90, check_halves=False)
    assert tm.n_trail_steps > 0
