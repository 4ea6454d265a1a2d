"""-m gpu: the parity tests proper.  Everything goes through the C ABI (libtalc_hip.so) and is
compared bit for bit with the oracle on the same seeded inputs.  Integer / byte / index work:
the bar is exact equality; the few doubles inside (count model, distances) are compared through
the decisions and orderings they drive."""
import os
import random

import numpy as np
import pytest

import oracle_lib as O
import parity_util as PU
from talc_amd import lib as T

pytestmark = pytest.mark.gpu
COMP = str.maketrans("ACGTN", "TGCAN")


def rc(s):
    return s.translate(COMP)[::-1]


def pack(reads):
    rb = "".join(reads).encode()
    offs = np.zeros(len(reads) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(x) for x in reads])
    return (np.frombuffer(rb, dtype=np.uint8) if rb else np.zeros(0, np.uint8)), offs


def unpack_kmer(km, k):
    return "".join("ACGT"[(int(km) >> (2 * (k - 1 - i))) & 3] for i in range(k))


# ---------------------------------------------------------------- table surface
def test_point_lookups_match_oracle(gpu_pair):
    rng = np.random.default_rng(1)
    q = np.concatenate([gpu_pair.keys[:20000], rng.integers(0, 1 << 42, 20000, dtype=np.uint64)])
    oc, oj = gpu_pair.otab.lookup_packed(q)
    gc, gj = gpu_pair.ttab.lookup(q)
    assert (oc == gc).all() and (oj == gj).all()
    assert (oc >= 2).sum() > 15000 and (oc == 0).sum() > 15000


def test_successor_queries_match_oracle(gpu_pair):
    """getNextCounts (Jellyfish.cpp:308-321) in both directions, order A,C,G,T."""
    rng = np.random.default_rng(2)
    kept = gpu_pair.keys[gpu_pair.counts >= 2][:3000]
    q = np.concatenate([kept, rng.integers(0, 1 << 42, 500, dtype=np.uint64)])
    for direction in (0, 1):
        gc, gj = gpu_pair.ttab.next_counts(q, direction)
        for i in range(0, len(q), 7):
            oc, oj = gpu_pair.otab.next_counts(unpack_kmer(q[i], 21), direction)
            assert oc.tolist() == gc[i].tolist() and oj.tolist() == gj[i].tolist(), (i, direction)
        assert (gc >= 2).sum() > 500


# ---------------------------------------------------------------- coverage kernel (Read::reCoverage)
def test_coverage_edge_cases(gpu_pair):
    base, offs0 = gpu_pair.reads(0, 6)
    r = PU.seqs_of(base, offs0)
    reads = ["", "ACGT", r[0][:21], r[0][:22], r[1].lower(), r[2][:500] + "N" + r[2][500:], "N" * 100,
             r[3][:2048 + 20], r[3][:2048 + 21], r[3][:512 + 20], r[3][:512 + 21], r[4] + r[5] + r[0] + r[1],     # tile boundaries (tiles of 512 positions)
             "RYKMSW" + r[5][:100]]
    bases, offs = pack(reads)
    b = gpu_pair.ctx.batch(bases, offs)
    b.coverage()
    c, j, ko, nin = b.fetch_coverage()
    for i, s in enumerate(reads):
        oc, oj, onin = gpu_pair.otab.coverage(s)
        gc = c[int(ko[i]):int(ko[i + 1])]
        assert len(gc) == max(0, len(s) - 21 + 1)
        if len(s) >= 21:
            assert (oc == gc).all() and (oj == j[int(ko[i]):int(ko[i + 1])]).all(), i
            assert onin == nin[i]
    b.close()


def _clean_reads(pair, first, n, rate):
    """Reads of the pair's own transcriptome (same seed and size: the generator's transcripts do not depend on the error
    rates) with `rate` substitutions / insertions / deletions each instead of 4 %."""
    from talc_amd.synth import Synth
    S = Synth(target_kmers=int(pair.synth.spec.target_kmers), k=pair.synth.k, seed=int(pair.synth.spec.seed),
              sub_rate=rate, ins_rate=rate, del_rate=rate)
    return S.reads(first, n)


def test_coverage_of_dense_reads_and_tile_boundaries(gpu_pair):
    """The device keeps the coverage as hits + bitmap words per tile of 512 positions: error-free reads (every position a
    hit, every tile full, ranks up to 511), reads cut at the tile boundaries, and nearly clean reads whose solid regions
    run across several tiles."""
    base, offs0 = _clean_reads(gpu_pair, 0, 12, 0.0)
    r = [s for s in PU.seqs_of(base, offs0) if len(s) > 1100][:4]
    assert len(r) >= 3
    reads = [r[0], r[0][:511 + 20], r[0][:512 + 20], r[0][:513 + 20], r[1][:1024 + 20], r[1][:1023 + 21], r[2],
             r[2][:300] + "N" + r[2][301:], r[1][:700].lower()]
    b2, o2 = _clean_reads(gpu_pair, 50, 20, 0.003)
    reads += PU.seqs_of(b2, o2)
    bases, offs = pack(reads)
    b = gpu_pair.ctx.batch(bases, offs)
    b.coverage()
    c, j, ko, nin = b.fetch_coverage()
    nfull = 0
    for i, s in enumerate(reads):
        oc, oj, onin = gpu_pair.otab.coverage(s)
        gc = c[int(ko[i]):int(ko[i + 1])]
        assert (oc == gc).all() and (oj == j[int(ko[i]):int(ko[i + 1])]).all() and onin == nin[i], i
        nfull += int((gc > 0).all())
    assert nfull >= 6            # the error-free ones are hits from end to end
    b.close()


def test_correction_of_nearly_clean_reads(gpu_pair):
    """Solid regions of hundreds of k-mers: the anchor search's regions exceed 64 k-mers (no lane prefetch), cross tile
    boundaries (no hit index) and get long anchor lists; few and short gaps."""
    for rate, n in ((0.003, 80), (0.0, 12)):
        bases, offs = _clean_reads(gpu_pair, 200, n, rate)
        bad, (so, ost), (sg, gst) = PU.compare_correction(gpu_pair, bases, offs, verbose=False)
        assert not bad, (rate, bad[:5])
        assert (ost == 0).sum() >= n - 2


def test_coverage_many_reads(gpu_pair):
    bases, offs = gpu_pair.reads(100, 400)
    b = gpu_pair.ctx.batch(bases, offs)
    b.coverage()
    c, j, ko, nin = b.fetch_coverage()
    seqs = PU.seqs_of(bases, offs)
    for i in range(0, 400, 3):
        oc, oj, onin = gpu_pair.otab.coverage(seqs[i])
        if len(seqs[i]) >= 21:
            assert (oc == c[int(ko[i]):int(ko[i + 1])]).all() and onin == nin[i]
    t = gpu_pair.ctx.timing()
    assert t.n_kmers == b.n_kmers and t.coverage_ms > 0
    b.close()


# ---------------------------------------------------------------- device DP primitives
def test_device_alignment_scores_match_oracle(gpu_pair):
    rnd = random.Random(3)
    L = O.lib()
    ctx = gpu_pair.ctx
    for it in range(60):
        n = rnd.choice([0, 1, 5, 40, 63, 64, 65, 130, 400, 1279, 1290, 3000])
        m = rnd.choice([0, 1, 7, 50, 64, 200, 1300])
        a = "".join(rnd.choice("ACGTN") for _ in range(n))
        b = list(a[:m]) if rnd.random() < 0.5 else [rnd.choice("ACGT") for _ in range(m)]
        for _ in range(rnd.randint(0, 6)):
            if b:
                b[rnd.randrange(len(b))] = rnd.choice("ACGT")
        b = "".join(b)
        for (mt, mm, g, fb) in ((0, -1, -1, 0), (4, -3, -2, 1), (1, 0, 0, 0), (4, -3, -2, 0)):
            exp = L.orc_global_alignment(a.encode(), b.encode(), mt, mm, g, fb, fb, 0, 0)
            got = ctx.test_dp(0, a, b, mt, mm, g, fb)
            assert got[5] == 0 and got[0] == exp, (n, m, mt, fb)


def test_device_edit_distance_and_lcs_match_oracle(gpu_pair):
    """edit_and_lcs (wavefront edit / indel distances, with the lane-skewed DP behind them) against the oracle's
    global alignment (0,-1,-1) and LCS (localAlignment(1,0,0)): similar sequences of every size class (1, 2, 4
    diagonals per lane), unrelated ones (the wavefronts give up, the DP takes over), N bases, empty inputs."""
    rnd = random.Random(13)
    L = O.lib()
    ctx = gpu_pair.ctx
    for it in range(110):
        n = rnd.choice([0, 1, 5, 40, 63, 64, 65, 130, 219, 221, 400, 441, 700, 1279, 1800, 2500, 4095, 4097, 4500])
        a = [rnd.choice("ACGTN" if rnd.random() < 0.1 else "ACGT") for _ in range(n)]
        kind = rnd.random()
        if kind < 0.7:
            b = []
            rate = rnd.choice([0.0, 0.02, 0.08, 0.15, 0.3])
            for ch in a:
                x = rnd.random()
                if x < rate / 3:
                    b.append(rnd.choice("ACGT"))
                elif x < 2 * rate / 3:
                    b.append(ch)
                    b.append(rnd.choice("ACGT"))
                elif x >= rate:
                    b.append(ch)
            if rnd.random() < 0.3:
                b = b[: rnd.randrange(len(b) + 1)]
        else:
            b = [rnd.choice("ACGT") for _ in range(rnd.choice([0, 1, 7, 50, 64, 200, 900, 4200]))]
        a, b = "".join(a), "".join(b)
        if rnd.random() < 0.5:
            a, b = b, a
        exp_e = L.orc_global_alignment(a.encode(), b.encode(), 0, -1, -1, 0, 0, 0, 0)
        exp_l = L.orc_global_alignment(a.encode(), b.encode(), 1, 0, 0, 0, 0, 0, 0)
        got = ctx.test_dp(4, a, b)
        assert got[5] == 0 and (got[0], got[1]) == (exp_e, exp_l), (len(a), len(b), got[:2].tolist(), exp_e, exp_l)
        got = ctx.test_dp(4, a, b, p0=1)     # LCS only (the form used when the edit score is not needed)
        assert got[5] == 0 and (got[0], got[1]) == (0, exp_l), (len(a), len(b), got[:2].tolist(), exp_l)
        # threshold form: "is the LCS at least t" may stop early with t itself, and only when that is true
        for t in {max(1, exp_l - 3), exp_l, exp_l + 1, max(1, (7 * max(len(a), len(b)) + 9) // 10), 1}:
            got = ctx.test_dp(4, a, b, p0=2, p1=t)
            assert got[5] == 0 and got[0] == 0
            assert got[1] == exp_l or (got[1] == t and exp_l >= t), (len(a), len(b), t, int(got[1]), exp_l)


def test_device_edit_distance_and_lcs_beyond_4096_columns(gpu_pair):
    """The bit-vector routines take sequences beyond 4096 positions in blocks of 4096 columns (a 20 kb read's gap is
    searched with Trails of up to 24 k bases): block boundaries at, just before and just after multiples of 4096, two to
    four blocks, similar (12-20 % apart: the wavefront forms give up) and unrelated pairs, both argument orders, N bases."""
    rnd = random.Random(131)
    L = O.lib()
    ctx = gpu_pair.ctx
    sizes = [(4096, 4096), (4097, 4096), (8192, 8191), (8193, 5000), (9000, 9400), (12288, 12000), (13001, 700), (16500, 4100)]
    for n, m in sizes:
        a = [rnd.choice("ACGTN" if rnd.random() < 0.05 else "ACGT") for _ in range(n)]
        if rnd.random() < 0.75:
            b = []
            rate = rnd.choice([0.12, 0.2])
            for ch in a:
                x = rnd.random()
                if x < rate / 3:
                    b.append(rnd.choice("ACGT"))
                elif x < 2 * rate / 3:
                    b.append(ch)
                    b.append(rnd.choice("ACGT"))
                elif x >= rate:
                    b.append(ch)
            b = (b + [rnd.choice("ACGT") for _ in range(max(0, m - len(b)))])[:m]
        else:
            b = [rnd.choice("ACGT") for _ in range(m)]
        a, b = "".join(a), "".join(b)
        exp_e = L.orc_global_alignment(a.encode(), b.encode(), 0, -1, -1, 0, 0, 0, 0)
        exp_l = L.orc_global_alignment(a.encode(), b.encode(), 1, 0, 0, 0, 0, 0, 0)
        for x, y in ((a, b), (b, a)):
            got = ctx.test_dp(4, x, y)
            assert got[5] == 0 and (got[0], got[1]) == (exp_e, exp_l), (len(x), len(y), got[:2].tolist(), exp_e, exp_l)
        got = ctx.test_dp(4, a, b, p0=1)
        assert got[5] == 0 and (got[0], got[1]) == (0, exp_l), (n, m, got[:2].tolist(), exp_l)


def test_device_alignment_rows_continued_equal_the_alignment_from_scratch(gpu_pair):
    """scoreBridges (Explorer.cpp:689-706): the device keeps every Trail's last alignment row over the whole reference and
    adds the rows of the Trail's new bases (wave_nw_rows).  References of 30 to 8191 bases (every instance, 2 to 128
    columns per lane), candidates growing by 1..17 bases from a first scoring of 21 or 70-200 bases, truncation windows of
    5-15: the kept row's entry against the
    alignment from scratch (nw_score, itself pinned to the oracle above) at every scoring."""
    rnd = random.Random(97)
    ctx = gpu_pair.ctx
    total = 0
    for n in [30, 64, 65, 128, 129, 200, 260, 390, 520, 770, 1030, 1540, 2047, 2048, 2500, 3072, 3073, 4095, 4096, 5000, 6144, 6145, 8191]:
        ref = [rnd.choice("ACGT") for _ in range(n)]
        cand = []
        for ch in ref[: min(n, 700)]:
            x = rnd.random()
            if x < 0.05:
                cand.append(rnd.choice("ACGT"))
            elif x < 0.10:
                cand.append(ch)
                cand.append(rnd.choice("ACGT"))
            elif x >= 0.15:
                cand.append(ch)
        cand = cand + [rnd.choice("ACGT") for _ in range(15)]
        step = rnd.choice([1, 3, 6, 6, 9]) if n <= 260 else rnd.choice([6, 9, 17])
        for first in (21, rnd.choice([70, 130, 200])):   # rows of the first scoring: within one 64-row block / over several
            if first > len(cand):
                continue
            got = ctx.test_dp(7, "".join(ref), "".join(cand), first, 0, step, rnd.choice([5, 10, 15]))
            assert got[5] == 0 and got[0] > 0
            assert got[1] == 0, (n, len(cand), first, step, "first differing length", int(got[2]))
            total += int(got[0])
    assert total > 1300


def test_device_seed_and_extension_matches_oracle(gpu_pair):
    import ctypes as C
    rnd = random.Random(4)
    L = O.lib()
    ctx = gpu_pair.ctx
    for it in range(160):
        n = rnd.choice([21, 22, 30, 80, 200, 600, 1400])
        ref = [rnd.choice("ACGT") for _ in range(n)]
        cand = list(ref)
        for _ in range(rnd.randint(0, 12)):
            p = rnd.randrange(len(cand))
            x = rnd.random()
            if x < 0.4:
                cand[p] = rnd.choice("ACGT")
            elif x < 0.7:
                cand.insert(p, rnd.choice("ACGT"))
            elif len(cand) > 25:
                del cand[p]
        if rnd.random() < 0.3:
            cand = cand[: max(21, rnd.randrange(len(cand) + 1))]
        elif rnd.random() < 0.3:
            cand = cand + [rnd.choice("ACGT") for _ in range(rnd.randint(1, 40))]
        ref, cand = "".join(ref), "".join(cand)
        # <= ~30: band held in registers; above: LDS anti-diagonals (short segments) or HBM ones (long segments)
        # (up to 127: the phased wavefront; 128-255: its eight-diagonals-per-lane instance; beyond: anti-diagonal sweeps)
        xdrop = rnd.randint(-1, 30) if rnd.random() < 0.5 else rnd.randint(31, 300)
        for direction in (0, 1):
            out = np.zeros(3, dtype=np.int64)
            stop = C.c_int32()
            sc = L.orc_seed_and_extension(ref.encode(), cand.encode(), xdrop, direction, 21, out.ctypes.data, C.byref(stop))
            # growth order: walking LEFT the device sees both sequences reversed
            a, b = (ref, cand) if direction else (ref[::-1], cand[::-1])
            got = ctx.test_dp(1, a, b, xdrop, direction)
            assert got[5] == 0
            assert (got[0], got[1], got[2], got[3], got[4]) == (int(out[0]), int(out[1]), int(out[2]), int(sc), stop.value), \
                (n, len(cand), xdrop, direction)


def test_device_seed_and_extension_beyond_x_255(gpu_pair):
    """x-drops of 256 .. 511 (a Trail of a thousand steps and more, scored every CHECK_INTERVAL steps with an x that grows
    by 2 per scoring: Explorer.cpp:713): the sixteen-diagonals-per-lane phase of the wavefront form, against the oracle's
    anti-diagonal restatement.  Sequences as such a search has them: a nearly clean query of 300 - 1300 bases against a
    database at least as long; beyond 511, and segments the LDS stage does not hold, the anti-diagonal sweep answers."""
    import ctypes as C
    rnd = random.Random(44)
    L = O.lib()
    ctx = gpu_pair.ctx
    for it in range(36):
        n = rnd.choice([400, 700, 1000, 1300, 1700])
        ref = [rnd.choice("ACGT") for _ in range(n)]
        cand = list(ref[: rnd.randint(300, min(n, 1300))])
        for _ in range(rnd.choice([0, 1, 3, 10, 40])):
            p = rnd.randrange(len(cand))
            x = rnd.random()
            if x < 0.4:
                cand[p] = rnd.choice("ACGT")
            elif x < 0.7:
                cand.insert(p, rnd.choice("ACGT"))
            elif len(cand) > 25:
                del cand[p]
        ref, cand = "".join(ref), "".join(cand)
        xdrop = rnd.randint(256, 511) if it % 6 else rnd.randint(512, 700)
        for direction in (0, 1):
            out = np.zeros(3, dtype=np.int64)
            stop = C.c_int32()
            sc = L.orc_seed_and_extension(ref.encode(), cand.encode(), xdrop, direction, 21, out.ctypes.data, C.byref(stop))
            a, b = (ref, cand) if direction else (ref[::-1], cand[::-1])
            got = ctx.test_dp(1, a, b, xdrop, direction)
            assert got[5] == 0
            assert (got[0], got[1], got[2], got[3], got[4]) == (int(out[0]), int(out[1]), int(out[2]), int(sc), stop.value), \
                (n, len(cand), xdrop, direction)


def test_device_multi_xdrop_run_equals_the_single_runs(gpu_pair):
    """findStopPosition asks for x, x-1, x-2, ...: seed_and_extension_multi takes all of [0, x] from one wavefront run.
    On the device, every x of that run against the single-x routine (itself pinned to the oracle above): similar and
    unrelated pairs, both directions, prefixes / overhangs, runs that reach the far corner, x up to 60."""
    rnd = random.Random(44)
    ctx = gpu_pair.ctx
    used = 0
    for it in range(160):
        n = rnd.choice([21, 22, 25, 30, 60, 120, 300, 700])
        ref = [rnd.choice("ACGT") for _ in range(n)]
        cand = list(ref)
        for _ in range(rnd.choice([0, 1, 2, 5, 12, 30])):
            p = rnd.randrange(len(cand))
            x = rnd.random()
            if x < 0.4:
                cand[p] = rnd.choice("ACGT")
            elif x < 0.7:
                cand.insert(p, rnd.choice("ACGT"))
            elif len(cand) > 25:
                del cand[p]
        r = rnd.random()
        if r < 0.3:
            cand = cand[: max(21, rnd.randrange(len(cand) + 1))]
        elif r < 0.6:
            cand = cand + [rnd.choice("ACGT") for _ in range(rnd.randint(1, 40))]
        if rnd.random() < 0.1:
            cand = cand[:21] + [rnd.choice("ACGT") for _ in range(rnd.randint(0, 80))]
        ref, cand = "".join(ref), "".join(cand)
        x_hi = rnd.choice([1, 2, 3, 5, 8, 13, 21, 31, 32, 45, 60])
        for direction in (0, 1):
            a, b = (ref, cand) if direction else (ref[::-1], cand[::-1])
            got = ctx.test_dp(5, a, b, x_hi, direction)
            assert got[5] == 0
            if got[0]:
                used += 1
                assert got[1] == 0, (n, len(cand), x_hi, direction, "first differing x", int(got[2]))
    assert used > 250


def test_device_kept_wavefront_equals_the_extension_from_scratch(gpu_pair):
    """An edge search scores its growing Trail every few steps; the seed extension keeps the last wavefront level that
    had not met the Trail's end and the next scoring resumes from it (talc_wave.h: WfaKeep).  Every scoring of a growing
    candidate through that path must equal the extension from level 0 — for candidates that follow the reference with
    ONT-like errors, that diverge from it, that are identical to it, and for both directions."""
    rnd = random.Random(17)
    K = 21
    total = resumed = 0
    for it in range(60):
        n = rnd.randint(120, 700)
        ref = "".join(rnd.choice("ACGT") for _ in range(n))
        kind = it % 4
        cand = list(ref)
        if kind == 0:      # noisy copy (12 % errors), same anchor
            out = []
            for ch in ref[K:]:
                u = rnd.random()
                if u < 0.04:
                    continue
                if u < 0.08:
                    out.append(rnd.choice("ACGT"))
                out.append(ch if u >= 0.12 else rnd.choice("ACGT"))
            cand = ref[:K] + "".join(out)
        elif kind == 1:    # follows for a while, then random
            cut = rnd.randint(K + 5, n // 2)
            cand = ref[:cut] + "".join(rnd.choice("ACGT") for _ in range(n - cut))
        elif kind == 2:    # identical
            cand = ref
        else:              # few errors
            cand = "".join(ch if rnd.random() > 0.02 else rnd.choice("ACGT") for ch in ref)
            cand = ref[:K] + cand[K:]
        cand = cand[: min(len(cand), 600)]
        step = rnd.choice([1, 3, 6, 6, 6, 11])
        for dir_right in (1, 0):
            out = gpu_pair.ctx.test_dp(8, ref, cand, K + step, dir_right, step, rnd.choice([2, 3, 3, 5]))
            assert out[1] == 0, (it, kind, dir_right, step, out[:4])
            total += out[0]
            resumed += out[3]
    assert total > 2000 and resumed > total // 4, (total, resumed)


def test_device_window_search(gpu_pair):
    rnd = random.Random(5)
    ctx = gpu_pair.ctx
    for it in range(40):
        n = rnd.choice([21, 22, 64, 65, 100, 500])
        s = "".join(rnd.choice("AC") for _ in range(n))
        p = rnd.randrange(0, n - 21 + 1)
        pat = s[p:p + 21] if rnd.random() < 0.8 else "".join(rnd.choice("GT") for _ in range(21))
        assert ctx.test_dp(2, s, pat, p3=0)[0] == s.find(pat)
        assert ctx.test_dp(2, s, pat, p3=1)[0] == s.rfind(pat)


def test_device_tag_next_nodes_matches_oracle(gpu_pair):
    """tagNextNodes (Explorer.cpp:1226-1298) compiled for the device: tags and, for the successors
    that become Trails, the distance bit patterns (IEEE double sqrt / division on the GPU)."""
    import ctypes as C
    import struct
    rnd = random.Random(11)
    L = O.lib()
    ctx = gpu_pair.ctx
    p = gpu_pair.q
    for it in range(3000):
        count = rnd.choice([0, 1, 2, 3, 5, 30, 80, 400, 5000, rnd.randint(0, 100000)])

        def c():
            return rnd.choice([0, 0, 0, 1, 2, 3, count, max(0, count - rnd.randint(0, 10)), count + rnd.randint(0, 10),
                               rnd.randint(0, 50), rnd.randint(0, 20000)])
        cnt = [c(), c(), c(), c()]
        jc = [rnd.choice([0, 0, 0, 5]) for _ in range(4)]
        cx = rnd.random() < 0.3
        got = ctx.test_dp(3, struct.pack("<9I", *cnt, *jc, count), b"", int(cx))
        cn, jn = np.array(cnt, dtype=np.uint32), np.array(jc, dtype=np.uint32)
        t2, d2 = np.zeros(4, np.int32), np.zeros(4, np.float64)
        L.orc_tag_next_nodes(C.byref(p), cn.ctypes.data, jn.ctypes.data, count, int(cx), t2.ctypes.data, d2.ctypes.data)
        assert [int(x) for x in got[:4]] == t2.tolist(), (cnt, jc, count, cx)
        for i in range(4):
            if t2[i] in (0, 7):
                bits = struct.pack("<ii", int(got[4 + 2 * i]), int(got[5 + 2 * i]))
                assert bits == struct.pack("<d", d2[i]), (cnt, count, i)


# ---------------------------------------------------------------- whole hot path
def _check(pair, first, n, nthreads=8):
    bases, offs = pair.reads(first, n)
    bad, (so, ost), (sg, gst) = PU.compare_correction(pair, bases, offs, nthreads=nthreads, verbose=False)
    if bad:
        d = PU.first_trace_diff(pair, bases, offs, bad[0])
        raise AssertionError("reads %s differ (of %d); first trace difference of read %d: %s" % (bad[:8], n, bad[0], d))
    return so, ost


def test_correction_default_parameters(gpu_pair):
    so, st = _check(gpu_pair, 1000, 600)
    assert (st == 0).sum() > 550
    t = gpu_pair.ctx.timing()
    assert t.n_failed == 0 and t.n_trail_steps > 0 and t.n_dp_cells > 0


@pytest.mark.parametrize("kw", [
    dict(k=18), dict(k=25), dict(k=30), dict(k=31),
    dict(min_count=3, window_size=6, max_nb_competing_paths=5),
    dict(alpha=1.2, sr_error_rate=0.08, min_inner_score=0.4, min_border_score=0.55),
    dict(max_nb_competing_paths=12, window_size=15),
], ids=lambda d: ",".join("%s=%s" % kv for kv in d.items()))
def test_correction_parameter_variants(kw):
    k = kw.pop("k", 21)
    pair = PU.Pair(target_kmers=250_000, k=k, seed=20 + k, **kw)
    pair.upload(0)
    _check(pair, 0, 160)


@pytest.mark.parametrize("synth_kw,kw", [
    (dict(paralog_frac=0.5, paralog_div=0.03), dict()),
    (dict(paralog_frac=0.7, paralog_div=0.06), dict(max_nb_competing_paths=4, window_size=7)),
    (dict(paralog_frac=0.5, paralog_div=0.01), dict(max_nb_competing_paths=12, min_count=3)),
], ids=["default", "maxb4", "maxb12"])
def test_correction_on_branching_graphs(synth_kw, kw):
    """Paralog families put forks and bubbles into the graph: most searches then carry several Trails, go through
    scoreBridges / gardening (including its out-of-range read, Explorer.cpp:852) and the generic expansion step —
    the code the unique-sequence transcriptome of the other cases hardly touches."""
    pair = PU.Pair(target_kmers=250_000, k=21, seed=77, synth_kw=synth_kw, **kw)
    pair.upload(0)
    _check(pair, 0, 200)


def test_correction_with_junction_colours():
    pair = PU.Pair(target_kmers=300_000, k=21, seed=31, junctions=True)
    pair.upload(0)
    _check(pair, 0, 300)


def test_correction_reverse_mode():
    """-rev (main.cpp:253,286): reads from the opposite strand are corrected and flipped back;
    reads that are not corrected stay reverse-complemented."""
    pair = PU.Pair(target_kmers=300_000, k=21, seed=32, reverse=1)
    pair.upload(0)
    bases, offs = pair.reads(0, 200)
    reads = [rc(s) for s in PU.seqs_of(bases, offs)]
    b2, o2 = pack(reads)
    bad, (so, ost), (sg, gst) = PU.compare_correction(pair, b2, o2, verbose=False)
    assert not bad
    assert (ost == 0).sum() > 150
    # property: -rev on the opposite strand == forward correction of the original, flipped
    fwd = PU.Pair(target_kmers=300_000, k=21, seed=32)
    fwd.upload(0)
    g_out, g_off, g_st = fwd.ctx.correct(bases, offs)
    f = PU.seqs_of(g_out, g_off)
    for i in range(200):
        if gst[i] == 0:
            assert sg[i] == rc(f[i])


def test_correction_edge_inputs(gpu_pair):
    base, offs0 = gpu_pair.reads(5000, 8)
    r = PU.seqs_of(base, offs0)
    reads = ["", r[0][:21], r[0][:22], r[1].lower(), r[2][:400] + "N" + r[2][400:],
             r[3][:300] + "N" * 10 + r[3][300:900] + "RYKM" + r[3][900:], "ACGT" * 300, "A" * 500,
             "".join(random.Random(1).choice("ACGT") for _ in range(1500)), r[4], r[5][:60], r[6] + r[7]]
    bases, offs = pack(reads)
    bad, (so, ost), (sg, gst) = PU.compare_correction(gpu_pair, bases, offs, verbose=False)
    assert not bad, bad
    assert ost[0] == 1 and ost[1] == 1           # SKIPPED_SHORT (main.cpp:262)
    assert 2 in ost.tolist()                     # NO_SOLID_KMER reached


def test_reads_with_many_N(gpu_pair):
    base, offs0 = gpu_pair.reads(6000, 60)
    rnd = random.Random(7)
    reads = []
    for s in PU.seqs_of(base, offs0):
        s = list(s)
        for _ in range(rnd.randint(1, 25)):
            if s:
                s[rnd.randrange(len(s))] = "N"
        reads.append("".join(s))
    bases, offs = pack(reads)
    bad, _, _ = PU.compare_correction(gpu_pair, bases, offs, verbose=False)
    assert not bad, bad


def test_scratch_retry_pass_gives_identical_records(gpu_pair, monkeypatch):
    """Reads whose per-wave scratch overflows are redone with 8x scratch: same records."""
    bases, offs = gpu_pair.reads(7000, 200)
    ref_out, ref_off, ref_st = gpu_pair.ctx.correct(bases, offs)
    monkeypatch.setenv("TALC_TEST_TINY_CAPS", "1")
    ctx2 = T.Context(gpu_pair.ttab, gpu_pair.p, 0)
    out, oo, st = ctx2.correct(bases, offs)
    t = ctx2.timing()
    assert t.n_retried > 0 and t.n_failed == 0
    assert np.array_equal(out, ref_out) and np.array_equal(oo, ref_off) and np.array_equal(st, ref_st)
    ctx2.close()


def test_retry_stage_that_does_not_fit_is_a_warning_not_an_error(gpu_pair, monkeypatch):
    """The first pass leaves a valid batch; when the (much larger) retry stage cannot be allocated the flagged reads
    are passed through with READ_ERROR and the call returns TALC_WARN_READ_ERRORS — never a negative code that would
    abort a whole run (include/talc_hip.h: talc_batch_correct)."""
    bases, offs = gpu_pair.reads(7000, 200)
    ref_out, ref_off, ref_st = gpu_pair.ctx.correct(bases, offs)
    monkeypatch.setenv("TALC_TEST_TINY_CAPS", "1")
    monkeypatch.setenv("TALC_TEST_FAIL_RETRY_ALLOC", "1")
    ctx2 = T.Context(gpu_pair.ttab, gpu_pair.p, 0)
    b = ctx2.batch(bases, offs)
    assert b.correct() == T.WARN_READ_ERRORS
    out, oo, st = b.fetch_corrected()
    b.close()
    t = ctx2.timing()
    assert t.n_retried > 0 and t.n_failed == t.n_retried
    seqs, got, want = PU.seqs_of(bases, offs), PU.seqs_of(out, oo), PU.seqs_of(ref_out, ref_off)
    nerr = 0
    for i in range(len(seqs)):
        if st[i] == T.READ_ERROR:
            nerr += 1
            assert got[i] == seqs[i]                      # passed through unchanged
        else:
            assert st[i] == ref_st[i] and got[i] == want[i]   # every other record is the real one
    assert nerr == t.n_failed
    ctx2.close()


def _noisy(rnd, seq, rate):
    out = []
    for ch in seq:
        x = rnd.random()
        if x < rate * 0.4:
            out.append(rnd.choice("ACGT"))          # substitution
        elif x < rate * 0.7:
            continue                                # deletion
        elif x < rate:
            out.append(ch)
            out.append(rnd.choice("ACGT"))          # insertion
        else:
            out.append(ch)
    return "".join(out)


def test_correction_across_tandem_repeats_and_cycles():
    """Tandem repeats with a unit longer than K put cycles into the graph: a Trail that walks into the second copy of the
    unit meets a k-mer of its own path, Trail::ThinkIveAlreadyGotThere (Trail.cpp:289-302) says so and oneMoreStep /
    oneMoreStepInTheDark drop it (Explorer.cpp:586, 648-655).  The synthetic transcriptomes of the other cases never take
    that branch (0 of 24 M window searches of the branching bench workload find anything).  Units of K+2 .. 3K bases, 2-6
    copies, a diverged copy in some (a bubble next to the cycle), reads with 8-14 % errors across them."""
    rnd = random.Random(1234)
    k = 21
    transcripts = []
    for t in range(60):
        parts = ["".join(rnd.choice("ACGT") for _ in range(rnd.randint(150, 400)))]
        for _ in range(rnd.randint(1, 3)):
            unit = "".join(rnd.choice("ACGT") for _ in range(rnd.randint(k + 2, 3 * k)))
            copies = rnd.randint(2, 6)
            for c in range(copies):
                u = unit
                if rnd.random() < 0.3:   # a copy with one substitution
                    i = rnd.randrange(len(u))
                    u = u[:i] + rnd.choice("ACGT".replace(u[i], "")) + u[i + 1:]
                parts.append(u)
            parts.append("".join(rnd.choice("ACGT") for _ in range(rnd.randint(100, 350))))
        transcripts.append("".join(parts))
    pair = PU.CustomPair(transcripts, k=k, depth=20)
    pair.upload(0)
    reads = []
    for t in transcripts:
        for _ in range(4):
            a = rnd.randint(0, 60)
            b = len(t) - rnd.randint(0, 60)
            reads.append(_noisy(rnd, t[a:b], rnd.choice([0.08, 0.11, 0.14])))
    bases, offs = pack(reads)
    bad, (so, ost), (sg, gst) = PU.compare_correction(pair, bases, offs, nthreads=8, verbose=False)
    if bad:
        d = PU.first_trace_diff(pair, bases, offs, bad[0])
        raise AssertionError("reads %s differ (of %d); first trace difference of read %d: %s" % (bad[:8], len(reads), bad[0], d))
    assert (ost == 0).sum() > 200


def test_region_whose_last_position_is_not_a_hit():
    """defineStructure2 can leave a solid region whose END k-mer is not in the table.  Round 3's coverage layout flags a
    region 'clean' (its pairs consecutive from its start's hit index) — a test that must look at the end position itself:
    without that, the pivot of such a region read the NEXT hit's count (the first Trail's count, the anchor walk's first
    level).  Found by tools/stress_branching.py (one read in 7500); this is that read and its neighbours."""
    pair = PU.Pair(target_kmers=300_000, k=21, seed=103, synth_kw=dict(paralog_frac=0.8, paralog_div=0.04),
                   max_nb_competing_paths=8, check_interval=4)
    pair.upload(0)
    _check(pair, 1300, 100)


def test_one_context_over_batches_of_changing_shape_on_a_branching_graph():
    """The kept alignment rows of scoreBridges live in a per-wave arena that is never reset: batches of different
    longest reads (the scratch slots move) and searches of different reference lengths (the records' stride changes)
    through ONE context must still give the oracle's records."""
    pair = PU.Pair(target_kmers=250_000, k=21, seed=77, synth_kw=dict(paralog_frac=0.6, paralog_div=0.04))
    pair.upload(0)
    bases, offs = pair.reads(0, 240)
    seqs = PU.seqs_of(bases, offs)
    o_out, o_off, o_st = pair.otab.correct_batch(bases, offs, nthreads=8)
    want = PU.seqs_of(o_out, o_off)
    order = sorted(range(len(seqs)), key=lambda i: len(seqs[i]))
    groups = [order[:60], order[180:], order[60:120], order[120:180], order[::3]]   # short, longest, then mixed again
    for rep in range(2):
        for g in groups:
            b, o = pack([seqs[i] for i in g])
            out, oo, st = pair.ctx.correct(b, o)
            got = PU.seqs_of(out, oo)
            for j, i in enumerate(g):
                assert got[j] == want[i] and int(st[j]) == int(o_st[i]), (rep, i)


def test_batch_composition_and_order_do_not_matter(gpu_pair):
    """Reads are independent units (main.cpp:247): one batch, several batches or a permuted batch
    give the same record per read."""
    bases, offs = gpu_pair.reads(8000, 300)
    seqs = PU.seqs_of(bases, offs)
    out, oo, st = gpu_pair.ctx.correct(bases, offs)
    whole = PU.seqs_of(out, oo)
    parts = []
    for lo, hi in ((0, 7), (7, 150), (150, 300)):
        b, o = pack(seqs[lo:hi])
        po, poo, pst = gpu_pair.ctx.correct(b, o)
        parts += PU.seqs_of(po, poo)
    assert parts == whole
    perm = np.random.default_rng(3).permutation(300)
    b, o = pack([seqs[i] for i in perm])
    po, poo, pst = gpu_pair.ctx.correct(b, o)
    got = PU.seqs_of(po, poo)
    assert [got[j] for j in np.argsort(perm)] == whole
    assert np.array_equal(pst[np.argsort(perm)], st)


def test_read_stats_rows_match_oracle(gpu_pair):
    """The rows Read::outputBasicReadStats would append (Read.cpp:418-433; the reference has the call commented out,
    main.cpp:305): raw length, span and number of the IN regions as the read's last step leaves them, corrected
    length — for corrected reads, reads without a solid k-mer, reads without a structure and reads of K bases or fewer."""
    base, offs0 = gpu_pair.reads(12000, 150)
    r = PU.seqs_of(base, offs0)
    reads = r + ["", r[0][:21], r[1][:22], "ACGT" * 200, "N" * 80,
                 "".join(random.Random(5).choice("ACGT") for _ in range(900))]
    bases, offs = pack(reads)
    o_out, o_off, o_st, o_rows = gpu_pair.otab.correct_batch_stats(bases, offs, nthreads=8)
    b = gpu_pair.ctx.batch(bases, offs)
    b.correct()
    g_out, g_off, g_st = b.fetch_corrected()
    g_rows = b.fetch_read_stats()
    b.close()
    assert np.array_equal(np.asarray(o_st), g_st) and np.array_equal(o_out, g_out)
    assert np.array_equal(o_rows, g_rows), np.nonzero((o_rows != g_rows).any(axis=1))[0][:10]
    assert set(o_st.tolist()) >= {0, 1, 2} and (o_rows[:, 0] == 0).sum() >= 2 and (o_rows[:, 4] > 0).sum() > 120


def test_trace_hook_matches_oracle(gpu_pair):
    bases, offs = gpu_pair.reads(9000, 3)
    for i in range(3):
        assert PU.first_trace_diff(gpu_pair, bases, offs, i) is None


def test_mixed_lengths_config5_shape():
    """BASELINE config 5 shape at small scale: K=31, reads log-uniform from 500 b to 20 kb (the generator draws every
    read from a transcript at least as long, so the long ones are as correctable as the short ones), 160 of them
    against the oracle, 30+ beyond 8 kb."""
    pair = PU.Pair(target_kmers=1_500_000, k=31, seed=41, synth_kw=dict(mixed_lengths=1))
    pair.upload(0)
    so, st = _check(pair, 0, 160, nthreads=16)
    lens = [len(s) for s in so]
    assert max(lens) > 15000 and min(l for l in lens if l > 31) < 1500
    assert sum(1 for l in lens if l > 8000) >= 30
    long_ok = [i for i in range(len(so)) if lens[i] > 8000 and st[i] == 0]
    assert len(long_ok) >= 25                                     # the long reads are corrected, not passed through
    t = pair.ctx.timing()
    assert t.n_failed == 0


def test_long_gaps_go_through_the_retry_pass(monkeypatch):
    """The Trail buffers of a search are cut from one arena with the stride the search's gap asks for; a gap whose
    stride does not fit the first pass's arena (OVF_SEQ), or that runs out of buffers (OVF_TRAILS), sends its read to
    the retry pass, whose arena holds the full pool for the longest read.  With a 600-byte arena every gap beyond
    ~400 bases takes that route (K=31: most reads have one): same records as the oracle, nothing fails."""
    monkeypatch.setenv("TALC_SEQ_ARENA", "600")
    pair = PU.Pair(target_kmers=600_000, k=31, seed=43, synth_kw=dict(mixed_lengths=1))
    pair.upload(0)
    _check(pair, 0, 60, nthreads=16)
    t = pair.ctx.timing()
    assert t.n_retried >= 20 and t.n_failed == 0


# ---------------------------------------------------------------- walk tables (fast-forward accelerator)
def test_walk_tables_do_not_change_results(gpu_pair, monkeypatch):
    """The walk tables (WalkEntry, talc_common.h) are a derived device structure: with them (default, gpu_pair) and
    without them (TALC_WALK=0: the per-step probing form of the fast-forward) every record is the same."""
    bases, offs = gpu_pair.reads(9000, 400)
    ref_out, ref_off, ref_st = gpu_pair.ctx.correct(bases, offs)
    monkeypatch.setenv("TALC_WALK", "0")
    t2 = T.Table.from_arrays(gpu_pair.keys, gpu_pair.counts, gpu_pair.p)
    t2.decolour_repeats()
    t2.upload(0)
    assert 0 < t2.device_bytes < gpu_pair.ttab.device_bytes    # no walk tables in this copy
    ctx2 = T.Context(t2, gpu_pair.p, 0)
    out, oo, st = ctx2.correct(bases, offs)
    assert np.array_equal(out, ref_out) and np.array_equal(oo, ref_off) and np.array_equal(st, ref_st)
    ctx2.close()
    # and the per-step form against the oracle as well
    o_out, o_off, o_st = gpu_pair.otab.correct_batch(bases, offs, nthreads=8)
    assert PU.seqs_of(o_out, o_off) == PU.seqs_of(out, oo) and np.array_equal(np.asarray(o_st), st)


@pytest.mark.parametrize("form", ["walk-clamped", "per-step"])
def test_walk_tables_with_counts_beyond_their_fields(form):
    """A walk level holds the largest count in 13 bits (larger: the walk stops there and the generic step takes over)
    and a flag "exactly one successor >= MIN_COUNT" for the table's MIN_COUNT; a MIN_COUNT that does not fit the
    count field makes the fast-forward use the per-step form instead of the walk tables."""
    from talc_amd.synth import Synth
    S = Synth(target_kmers=200_000, k=21, seed=23)
    keys, counts = S.dump_arrays()
    if form == "walk-clamped":     # counts x 700: about a third of them above 16383, a few percent above 65535
        big = (counts.astype(np.uint64) * 700).astype(np.uint32)
        minc = 2 * 700
        assert minc < 0x1FFF and int((big > 65535).sum()) > 100 and int((big > 0x1FFF).sum()) > 10000
    else:                          # MIN_COUNT itself beyond 14 bits
        big = (counts.astype(np.uint64) * 9000).astype(np.uint32)
        minc = 2 * 9000
        assert minc > 0x1FFF
    p, q = PU.both_params(k=21, min_count=minc)
    otab = O.OracleTable(q, O.OracleTable.FLAT)
    otab.insert_packed(keys, big)
    otab.decolour()
    ttab = T.Table.from_arrays(keys, big, p)
    ttab.decolour_repeats()
    ttab.upload(0)
    ctx = T.Context(ttab, p, 0)
    bases, offs = S.reads(0, 150)
    o_out, o_off, o_st = otab.correct_batch(bases, offs, nthreads=8)
    g_out, g_off, g_st = ctx.correct(bases, offs)
    assert PU.seqs_of(o_out, o_off) == PU.seqs_of(g_out, g_off) and np.array_equal(np.asarray(o_st), g_st)
    assert ctx.timing().n_trail_steps > 0
    ctx.close()


# ---------------------------------------------------------------- device table builder
def _revcomp_packed(km, k):
    s = unpack_kmer(km, k)
    r = rc(s)
    v = 0
    for ch in r:
        v = (v << 2) | "ACGT".index(ch)
    return v


def test_device_built_table_matches_the_oracle_table(tmp_path):
    """buildCDBG on the GPU (talc_table_from_arrays_device / talc_table_build_device: insertion, junction colouring on
    both strands, homopolymer de-colouring, all as kernels on the device-resident image) against the ORACLE's table
    built from the same dump: same size and the same (count, colour) for every stored k-mer — duplicated lines with
    other counts included (first wins), junction lines that reach the same k-mer several times with other colours
    (last wins), colours at and above colouredCountThr, negative ones — for k-mers that are not stored, through the
    host image, on the device, and through the text parser.  The host builder is held to the same answers."""
    from talc_amd.synth import Synth
    S = Synth(target_kmers=400_000, k=21, seed=41)
    keys, counts = S.dump_arrays()
    rng = np.random.default_rng(3)
    # duplicate 5000 entries with other counts somewhere later in the dump, and some below min_count earlier
    dup = rng.integers(0, len(keys), 5000)
    keys2 = np.concatenate([keys[dup[:500]], keys, keys[dup]])
    counts2 = np.concatenate([np.ones(500, np.uint32), counts, (counts[dup] + 7).astype(np.uint32)])
    p, q = PU.both_params(k=21, use_junctions=1)
    jk, jc = S.junction_arrays()
    # junction lines that hit k-mers already coloured (the same line again with another count; the reverse complement
    # of a coloured k-mer as a line of its own), counts at / above the threshold and a negative one
    jd = rng.integers(0, len(jk), 3000)
    jk2 = np.concatenate([jk, jk[jd], np.array([_revcomp_packed(x, 21) for x in jk[jd[:300]]], dtype=np.uint64)])
    jc2 = np.concatenate([jc, (jc[jd] % 9000 + 11), np.full(300, 77, np.int64)]).astype(np.int64)
    jc2[len(jk):len(jk) + 10] = 10000
    jc2[len(jk) + 10:len(jk) + 20] = 9999
    jc2[len(jk) + 20:len(jk) + 25] = -5
    otab = O.OracleTable(q, O.OracleTable.FLAT)
    otab.insert_packed(keys2, counts2)
    otab.colour_packed(jk2, jc2)
    otab.decolour()
    th = T.Table.from_arrays(keys2, counts2, p)
    td = T.Table.from_arrays(keys2, counts2, p, device=0)
    assert len(th) == len(td) == len(otab) > 0
    for t in (th, td):
        t.colour(jk2, jc2)
        t.decolour_repeats()
    hom = np.array([0, (1 << 42) - 1, int("01" * 21, 2), int("10" * 21, 2)], dtype=np.uint64)
    qs = np.concatenate([keys2, rng.integers(0, 1 << 42, 50000, dtype=np.uint64), jk2,
                         np.array([_revcomp_packed(x, 21) for x in jk2[:2000]], dtype=np.uint64), hom])
    oc, oj = otab.lookup_packed(qs)
    assert int((oj > 0).sum()) > 1000
    hc, hj = th.lookup_host(qs)
    assert (hc == oc).all() and (hj == oj).all()
    dc, dj = td.lookup_host(qs)            # the device-resident image, copied back for this call
    assert (dc == oc).all() and (dj == oj).all()
    td.upload(0)                            # adopts the image where it is
    gc, gj = td.lookup(qs)
    assert (gc == oc).all() and (gj == oj).all()
    for direction in (0, 1):                # the LEFT table carries the same colours
        g4c, g4j = td.next_counts(jk2[:400], direction)
        for i in range(0, 400, 5):
            e4c, e4j = otab.next_counts(unpack_kmer(jk2[i], 21), direction)
            assert e4c.tolist() == g4c[i].tolist() and e4j.tolist() == g4j[i].tolist(), (i, direction)
    # through the text parser (buildCDBG + decolourRepeatsFromDBG as main.cpp:231-232 calls them)
    dump = str(tmp_path / "d.txt")
    S.write_dump(dump)
    junc = str(tmp_path / "j.txt")
    S.write_junctions(junc)
    ofile = O.OracleTable(q, O.OracleTable.FLAT)
    ost = ofile.build_from_files(dump, junc)
    fh = T.Table.from_files(dump, junc, p)
    fd = T.Table.from_files(dump, junc, p, device=0)
    assert len(fh) == len(fd) == len(ofile) and (fh.build_stats == fd.build_stats).all()
    assert int(fd.build_stats[0]) == int(ost[0]) and int(fd.build_stats[1]) == int(ost[1])
    ec, ej = ofile.lookup_packed(qs)
    a = fh.lookup_host(qs)
    b = fd.lookup_host(qs)
    assert (a[0] == ec).all() and (a[1] == ej).all() and (b[0] == ec).all() and (b[1] == ej).all()


def test_text_dump_parsed_on_the_device_equals_the_host_parser(tmp_path, capfd, monkeypatch):
    """A dump of a size that matters goes to the GPU as text and is parsed there (talc_kernels_build.h: k_parse_count /
    k_parse_lines): same table, same statistics as the host parser gives — duplicated lines (the first wins: the line number
    is the order), counts below MIN_COUNT, counts of 1-9 digits, a tab as the blank; and any line that is not canonical (a
    lower-case k-mer is canonical; two blanks, a carriage return, a trailing blank, a count of ten digits, no newline at the end are not) sends the
    whole file through the host's tokeniser, with the same answers as before."""
    from talc_amd.synth import Synth
    S = Synth(target_kmers=420_000, k=21, seed=43)
    keys, counts = S.dump_arrays()
    rng = np.random.default_rng(5)
    p, q = PU.both_params(k=21)
    lines = ["%s %d" % (unpack_kmer(k, 21), int(c)) for k, c in zip(keys, counts)]
    dup = rng.integers(0, len(keys), 4000)
    lines += ["%s\t%d" % (unpack_kmer(keys[i], 21).lower(), int(counts[i]) + 5) for i in dup]        # later duplicates lose
    lines = ["%s 1" % unpack_kmer(keys[i], 21) for i in dup[:300]] + lines                           # earlier ones below MIN_COUNT do not count
    lines += ["%s 123456789" % unpack_kmer(int(rng.integers(0, 1 << 42)), 21) for _ in range(50)]   # nine digits
    canonical = str(tmp_path / "canonical.txt")
    with open(canonical, "w") as f:
        f.write("\n".join(lines) + "\n")
    assert os.path.getsize(canonical) > (8 << 20)
    monkeypatch.setenv("TALC_TIMING", "1")
    capfd.readouterr()
    td = T.Table.from_files(canonical, None, p, device=0)
    assert "dump parsed on the device" in capfd.readouterr().err
    monkeypatch.setenv("TALC_HOST_PARSE", "1")
    th = T.Table.from_files(canonical, None, p, device=0)
    assert "dump parsed on the device" not in capfd.readouterr().err
    monkeypatch.delenv("TALC_HOST_PARSE")
    ofile = O.OracleTable(q, O.OracleTable.FLAT)
    ost = ofile.build_from_files(canonical, None)
    assert len(td) == len(th) == len(ofile) and (td.build_stats == th.build_stats).all()
    assert int(td.build_stats[0]) == int(ost[0]) and int(td.build_stats[1]) == int(ost[1])
    qs = np.concatenate([keys, rng.integers(0, 1 << 42, 50000, dtype=np.uint64)])
    ec, ej = ofile.lookup_packed(qs)
    for t in (td, th):
        c, j = t.lookup_host(qs)
        assert (c == ec).all() and (j == ej).all()
    # lines the device parser does not take: the host's tokeniser answers for the whole file
    for name, extra in (("two-blanks", ["%s  7" % unpack_kmer(keys[5], 21)]), ("crlf", ["%s 7\r" % unpack_kmer(keys[8], 21)]), ("trailing-blank", ["%s 7 " % unpack_kmer(keys[6], 21)]),
                        ("no-final-newline", None), ("ten-digits", ["%s 1234567890" % unpack_kmer(keys[7], 21)])):
        path = str(tmp_path / (name + ".txt"))
        with open(path, "w") as f:
            f.write("\n".join(lines[:len(lines) // 2] + (extra or []) + lines[len(lines) // 2:]) + ("" if extra is None else "\n"))
        capfd.readouterr()
        tdev = T.Table.from_files(path, None, p, device=0)
        err = capfd.readouterr().err
        assert "dump parsed on the device" not in err and "parsing on the host" in err, name
        oo = O.OracleTable(q, O.OracleTable.FLAT)
        oo.build_from_files(path, None)
        assert len(tdev) == len(oo), name
        c, j = tdev.lookup_host(qs[:100000])
        e = oo.lookup_packed(qs[:100000])
        assert (c == e[0]).all() and (j == e[1]).all(), name


def test_table_image_export_and_import():
    """Replication across GPUs: the device image leaves one table as two plain byte arrays in caller-owned
    device buffers and becomes a table again on the importing side; same answers, same corrected records."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")     # the HIP runtime libtalc_hip.so itself runs on: plain device buffers from it
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipFree.argtypes = [C.c_void_p]
    pair = PU.Pair(target_kmers=300_000, k=21, seed=52, junctions=True)
    pair.upload(0)
    nb = pair.ttab.image_bytes
    assert nb == pair.ttab.capacity * 32
    br, bl = C.c_void_p(), C.c_void_p()
    assert hip.hipMalloc(C.byref(br), nb) == 0 and hip.hipMalloc(C.byref(bl), nb) == 0
    pair.ttab.export_device(0, br.value, bl.value)
    t2 = T.Table.import_device(pair.p, pair.ttab.capacity, len(pair.ttab), br.value, bl.value, 0)
    # an image carries no parameters: one filtered with MIN_COUNT 2 is refused under MIN_COUNT 3 (its counts of 2 would be
    # "solid" k-mers of every region), and under another K (its keys are 40 bits wide)
    for bad in (dict(k=21, min_count=3), dict(k=19)):
        pb, _ = PU.both_params(use_junctions=1, **bad)
        with pytest.raises(T.TalcError, match="does not belong to these parameters"):
            T.Table.import_device(pb, pair.ttab.capacity, len(pair.ttab), br.value, bl.value, 0)
    hip.hipFree(br)
    hip.hipFree(bl)
    assert len(t2) == len(pair.ttab)
    t2.upload(0)
    rng = np.random.default_rng(9)
    qs = np.concatenate([pair.keys[:30000], rng.integers(0, 1 << 42, 10000, dtype=np.uint64)])
    a, b = pair.ttab.lookup(qs), t2.lookup(qs)
    assert (a[0] == b[0]).all() and (a[1] == b[1]).all()
    ctx2 = T.Context(t2, pair.p, 0)
    bases, offs = pair.reads(0, 120)
    r1 = pair.ctx.correct(bases, offs)
    r2 = ctx2.correct(bases, offs)
    assert all(np.array_equal(x, y) for x, y in zip(r1, r2))
    ctx2.close()
