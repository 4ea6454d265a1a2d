"""-m gpu: BASELINE.json configs[1] at FULL size (100k reads, 50M-entry k-mer dump) — too big for
an exhaustive oracle run, so it is checked through size-independent properties plus an oracle
spot-check on a random sample of the very same reads."""
import numpy as np
import pytest

import oracle_lib as O
import parity_util as PU
from talc_amd import lib as T
from talc_amd.synth import Synth

pytestmark = pytest.mark.gpu

N_READS = 100_000
N_KMERS = 50_000_000


def test_config2_full_size_properties():
    S = Synth(target_kmers=N_KMERS, k=21, seed=0)
    keys, counts = S.dump_arrays()
    p, q = PU.both_params(k=21)
    tab = T.Table.from_arrays(keys, counts, p)
    tab.decolour_repeats()
    tab.upload(0)
    ctx = T.Context(tab, p, 0)
    bases, offs = S.reads(0, N_READS)
    out, oo, st = ctx.correct(bases, offs)
    tm = ctx.timing()
    assert tm.n_failed == 0
    hist = np.bincount(st, minlength=5)
    assert hist.sum() == N_READS and hist[0] > 0.97 * N_READS and hist[4] == 0
    # records are over the Dna5 alphabet and their total size is plausible (correction does not
    # change the length by more than a few percent overall)
    assert set(np.unique(out).tolist()) <= set(b"ACGTN")
    assert 0.9 * len(bases) < len(out) < 1.1 * len(bases)
    # pass-through rule (main.cpp:310): reads that were not corrected come back unchanged
    seq_in = bytes(bases)
    seq_out = bytes(out)
    for i in np.nonzero(st != 0)[0][:200]:
        assert seq_out[int(oo[i]):int(oo[i + 1])] == seq_in[int(offs[i]):int(offs[i + 1])]
    # batch-composition independence: two halves give the same records as the whole batch
    h = N_READS // 2
    o1, oo1, st1 = ctx.correct(bases[: int(offs[h])], offs[: h + 1].copy())
    o2, oo2, st2 = ctx.correct(bases[int(offs[h]):], (offs[h:] - offs[h]).copy())
    assert np.array_equal(np.concatenate([o1, o2]), out)
    assert np.array_equal(np.concatenate([st1, st2]), st)
    # determinism: a second run of the same batch is bit-identical
    o3, oo3, st3 = ctx.correct(bases, offs)
    assert np.array_equal(o3, out) and np.array_equal(oo3, oo)
    # oracle spot-check on a random sample of the same reads (flat table: same values as the map)
    otab = O.OracleTable(q, O.OracleTable.FLAT)
    otab.insert_packed(keys, counts)
    otab.decolour()
    assert len(otab) == len(tab)
    rng = np.random.default_rng(5)
    idx = np.sort(rng.choice(N_READS, 400, replace=False))
    reads = [seq_in[int(offs[i]):int(offs[i + 1])] for i in idx]
    sb = np.frombuffer(b"".join(reads), dtype=np.uint8)
    so = np.zeros(len(reads) + 1, dtype=np.uint64)
    so[1:] = np.cumsum([len(x) for x in reads])
    e_out, e_off, e_st = otab.correct_batch(sb, so, nthreads=16)
    exp = PU.seqs_of(e_out, e_off)
    for j, i in enumerate(idx):
        assert seq_out[int(oo[i]):int(oo[i + 1])].decode() == exp[j], int(i)
        assert int(st[i]) == int(e_st[j])
