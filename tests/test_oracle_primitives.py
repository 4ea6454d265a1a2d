"""Known-answer tests that pin the oracle's SeqAn shim and count model (the reference ships no
tests, SURVEY.md §4: every KAT here is hand-derived or checked against an independent plain
Python restatement written in this file)."""
import ctypes as C
import random

import numpy as np
import pytest

import oracle_lib as O

L = O.lib()


def nw(h, v, m, mm, g, top=0, left=0, right=0, bottom=0):
    return L.orc_global_alignment(h.encode(), v.encode(), m, mm, g, top, left, right, bottom)


def py_nw(h, v, m, mm, g, top=False, left=False, right=False, bottom=False):
    n, k = len(h), len(v)
    D = [[0] * (n + 1) for _ in range(k + 1)]
    for j in range(n + 1):
        D[0][j] = 0 if top else j * g
    for i in range(1, k + 1):
        D[i][0] = 0 if left else i * g
        for j in range(1, n + 1):
            D[i][j] = max(D[i - 1][j - 1] + (m if h[j - 1] == v[i - 1] else mm), D[i - 1][j] + g, D[i][j - 1] + g)
    best = D[k][n]
    if right:
        best = max(best, max(D[i][n] for i in range(k + 1)))
    if bottom:
        best = max(best, max(D[k]))
    return best


def py_lcs(a, b):
    D = [[0] * (len(a) + 1) for _ in range(len(b) + 1)]
    for i in range(1, len(b) + 1):
        for j in range(1, len(a) + 1):
            D[i][j] = D[i - 1][j - 1] + 1 if a[j - 1] == b[i - 1] else max(D[i - 1][j], D[i][j - 1])
    return D[len(b)][len(a)]


# ---- hand-derived KATs -------------------------------------------------------------------
def test_global_alignment_edit_distance_kats():
    assert nw("ACGT", "ACGT", 0, -1, -1) == 0
    assert nw("ACGT", "AGT", 0, -1, -1) == -1          # one deletion
    assert nw("AAAA", "TTTT", 0, -1, -1) == -4         # four substitutions
    assert nw("", "ACG", 0, -1, -1) == -3
    assert nw("ACG", "", 0, -1, -1) == -3
    assert nw("ACGTACGT", "ACGACGT", 0, -1, -1) == -1
    assert nw("NNAC", "NNAC", 0, -1, -1) == 0          # N matches N (Dna5 ordinals)


def test_global_alignment_overlap_configs_kats():
    # Score(4,-3,-2): 4 matches = 16; AC-T vs ACGT = 12 - 2
    assert nw("ACGT", "ACGT", 4, -3, -2) == 16
    assert nw("ACGT", "ACT", 4, -3, -2) == 10
    # AlignConfig<true,true,false,false>: leading gaps free, trailing gaps charged
    assert nw("TTACGT", "ACGT", 4, -3, -2, 1, 1, 0, 0) == 16
    assert nw("TTACGT", "ACGT", 4, -3, -2, 0, 0, 0, 0) == 12
    assert nw("ACGTTT", "ACGT", 4, -3, -2, 1, 1, 0, 0) == 12
    # AlignConfig<false,false,true,true>: trailing gaps free, leading gaps charged
    assert nw("ACGTTT", "ACGT", 4, -3, -2, 0, 0, 1, 1) == 16
    assert nw("TTACGT", "ACGT", 4, -3, -2, 0, 0, 1, 1) == 12


def test_local_alignment_is_lcs_kats():
    assert L.orc_local_alignment(b"ACGT", b"ACGT", 1, 0, 0) == 4
    assert L.orc_local_alignment(b"ACGT", b"TGCA", 1, 0, 0) == 1
    assert L.orc_local_alignment(b"AACCGGTT", b"ACGT", 1, 0, 0) == 4
    assert L.orc_local_alignment(b"ACGT", b"", 1, 0, 0) == 0


def test_alignments_against_plain_python():
    rnd = random.Random(5)
    for _ in range(300):
        a = "".join(rnd.choice("ACGTN") for _ in range(rnd.randint(0, 40)))
        b = "".join(rnd.choice("ACGT") for _ in range(rnd.randint(0, 40)))
        for (m, mm, g) in ((0, -1, -1), (4, -3, -2), (1, 0, 0)):
            for flags in ((0, 0, 0, 0), (1, 1, 0, 0), (0, 0, 1, 1)):
                assert nw(a, b, m, mm, g, *flags) == py_nw(a, b, m, mm, g, *map(bool, flags)), (a, b, m, flags)
        if a and b:
            assert L.orc_local_alignment(a.encode(), b.encode(), 1, 0, 0) == py_lcs(a, b)


# ---- gapped x-drop extension ----------------------------------------------------------------
INT_MIN = -(2 ** 31)


def py_xdrop(query, db, right, match, mismatch, gap, xdrop):
    """Independent restatement of SeqAn2's _extendSeedGappedXDropOneDirection.
    Returns (moved, extCols, extRows)."""
    cols, rows = len(query) + 1, len(db) + 1
    if rows == 1 or cols == 1:
        return (False, 0, 0)
    undef = INT_MIN - gap
    d1, d2, d3 = [], [0], ([undef, undef] if -gap > xdrop else [gap, gap])
    o1 = o2 = o3 = 0
    minCol, maxCol, adn, best = 1, 2, 1, 0
    while minCol < maxCol:
        adn += 1
        d1, d2, d3 = d2, d3, d1
        o1, o2, o3 = o2, o3, minCol - 1
        d3 = [undef] * (maxCol + 1 - o3)
        if adn * gap > best - xdrop:
            if o3 == 0:
                d3[0] = adn * gap
            if adn - maxCol == 0:
                d3[maxCol - o3] = adn * gap
        adbest = adn * gap
        for col in range(minCol, maxCol):
            i3, i2, i1 = col - o3, col - o2, col - o1
            if right:
                qp, dp = col - 1, adn - col - 1
            else:
                qp, dp = cols - 1 - col, rows - 1 + col - adn
            tmp = max(d2[i2 - 1], d2[i2]) + gap
            tmp = max(tmp, d1[i1 - 1] + (match if query[qp] == db[dp] else mismatch))
            if tmp < best - xdrop:
                d3[i3] = undef
            else:
                d3[i3] = tmp
                adbest = max(adbest, tmp)
        best = max(best, adbest)
        while (minCol - o3 < len(d3) and d3[minCol - o3] == undef and minCol - o2 - 1 < len(d2)
               and d2[minCol - o2 - 1] == undef):
            minCol += 1
        while maxCol - o3 > 0 and d3[maxCol - o3 - 1] == undef and d2[maxCol - o2 - 1] == undef:
            maxCol -= 1
        maxCol += 1
        minCol = max(minCol, adn + 2 - rows)
        maxCol = min(maxCol, cols)
    lcol = len(d3) + o3 - 2
    lrow = adn - lcol
    lscore = d3[lcol - o3]
    if lscore == undef:
        if d2[len(d2) - 2] != undef:
            lcol = len(d2) + o2 - 2
            lrow = adn - 1 - lcol
            lscore = d2[lcol - o2]
        elif len(d2) > 2 and d2[len(d2) - 3] != undef:
            lcol = len(d2) + o2 - 3
            lrow = adn - 1 - lcol
            lscore = d2[lcol - o2]
    if lscore == undef:
        for i, v in enumerate(d1):
            if v > lscore:
                lscore, lcol, lrow = v, i + o1, adn - 2 - (i + o1)
    if lscore != undef:
        return (True, lcol, lrow)
    return (False, 0, 0)


def extend(database, query, seed, right, xdrop, m=0, mm=-1, g=-1):
    s = np.array(seed, dtype=np.int64)
    L.orc_extend_seed(database.encode(), query.encode(), s.ctypes.data, 1 if right else 0, m, mm, g, xdrop)
    return tuple(int(x) for x in s)


def test_xdrop_hand_traced_kats():
    # traced by hand in the design notes: identical 2-base segments extend fully
    assert extend("AC", "AC", (0, 0, 0, 0), True, 2) == (0, 0, 2, 2)
    # negative drop-off: every cell is cut, the seed does not move
    assert extend("ACGT", "ACGT", (0, 0, 1, 1), True, -1) == (0, 0, 1, 1)
    # nothing to extend into
    assert extend("ACGT", "ACGT", (0, 0, 4, 4), True, 5) == (0, 0, 4, 4)
    # identical sequences, leftwards from the end
    assert extend("ACGTACGT", "ACGTACGT", (6, 6, 8, 8), False, 2) == (0, 0, 8, 8)


def test_xdrop_against_plain_python():
    rnd = random.Random(9)
    for it in range(400):
        n = rnd.randint(1, 60)
        a = [rnd.choice("ACGT") for _ in range(n)]
        b = list(a)
        for _ in range(rnd.randint(0, 8)):          # mutate a copy
            p = rnd.randrange(len(b))
            r = rnd.random()
            if r < 0.4:
                b[p] = rnd.choice("ACGT")
            elif r < 0.7:
                b.insert(p, rnd.choice("ACGT"))
            elif len(b) > 1:
                del b[p]
        a, b = "".join(a), "".join(b)
        xdrop = rnd.randint(-1, 12)
        right = rnd.random() < 0.5
        if right:
            sh, sv = rnd.randint(0, len(a)), rnd.randint(0, len(b))
            got = extend(a, b, (0, 0, sh, sv), True, xdrop)
            moved, ec, er = py_xdrop(b[sv:], a[sh:], True, 0, -1, -1, xdrop)
            assert got == (0, 0, sh + (er if moved else 0), sv + (ec if moved else 0)), (a, b, sh, sv, xdrop)
        else:
            sh, sv = rnd.randint(0, len(a)), rnd.randint(0, len(b))
            got = extend(a, b, (sh, sv, len(a), len(b)), False, xdrop)
            moved, ec, er = py_xdrop(b[:sv], a[:sh], False, 0, -1, -1, xdrop)
            assert got == (sh - (er if moved else 0), sv - (ec if moved else 0), len(a), len(b)), (a, b, sh, sv, xdrop)


def test_seed_and_extension_shapes():
    # Trail.cpp:341-437: identical reference and candidate -> both fully covered, score 0
    ref = "ACGTTGCAAGGCTTAACCGGTTAACG" * 2
    out = np.zeros(3, dtype=np.int64)
    stop = C.c_int32()
    sc = L.orc_seed_and_extension(ref.encode(), ref.encode(), 3, 1, 21, out.ctypes.data, C.byref(stop))
    assert (int(out[0]), int(out[1]), int(out[2]), sc, stop.value) == (len(ref), len(ref), len(ref), 0.0, 0)
    sc = L.orc_seed_and_extension(ref.encode(), ref.encode(), 3, 0, 21, out.ctypes.data, C.byref(stop))
    assert (int(out[0]), int(out[1]), int(out[2]), sc, stop.value) == (len(ref), len(ref), 0, 0.0, 0)
    # walking RIGHT with nothing beyond the K-1 seed: "BUG IN ALIGNMENT" branch, score = -xdrop, stop
    k21 = ref[:20]
    sc = L.orc_seed_and_extension((k21 + "A").encode(), (k21 + "C").encode(), 0, 1, 21, out.ctypes.data, C.byref(stop))
    assert stop.value == 1 and sc == 0.0 and int(out[0]) == 20 and int(out[1]) == 20


# ---- count model (Explorer.cpp:1185-1217) ---------------------------------------------------
def test_count_model_kats():
    p = O.params()
    a = 2.57
    # cc <= 3, UNEXPECTED class: nextc <= cc+0.5 + a*sqrt(cc+0.5)
    for cc in (0, 1, 2, 3):
        thr = (cc + 0.5) + a * (cc + 0.5) ** 0.5
        for nextc in range(0, 12):
            assert L.orc_is_expected_by_model(C.byref(p), nextc, cc, 1) == int(nextc <= thr)
    # cc == 0, EXPECTED class: sqrt(-0.5) is NaN -> always false
    for nextc in range(0, 5):
        assert L.orc_is_expected_by_model(C.byref(p), nextc, 0, 0) == 0
    # cc > 3: Begaud bounds
    for cc in (4, 10, 30, 1000):
        up = (a / 2 + (cc + 0.96) ** 0.5) ** 2
        lo = (a / 2 - (cc + 0.02) ** 0.5) ** 2
        for nextc in (0, 1, int(lo) - 1, int(lo), int(lo) + 1, cc, int(up), int(up) + 1, 5 * cc):
            if nextc < 0:
                continue
            assert L.orc_is_expected_by_model(C.byref(p), nextc, cc, 1) == int(nextc <= up)
            assert L.orc_is_expected_by_model(C.byref(p), nextc, cc, 0) == int(nextc >= lo)
            assert L.orc_is_expected_by_last_node(C.byref(p), nextc, cc) == int(lo <= nextc <= up)


def tag(counts, jc, count, complex_=False, **kw):
    p = O.params(**kw)
    c = np.array(counts, dtype=np.uint32)
    j = np.array(jc, dtype=np.uint32)
    t = np.zeros(4, dtype=np.int32)
    d = np.zeros(4, dtype=np.float64)
    L.orc_tag_next_nodes(C.byref(p), c.ctypes.data, j.ctypes.data, count, int(complex_), t.ctypes.data, d.ctypes.data)
    return t.tolist(), d.tolist()


EXPECTED, UNEXPECTED, BREAKPOINT = 0, 1, 7


def test_tag_next_nodes_kats():
    # no successor in the table: no tags at all (dead end)
    assert tag([0, 0, 0, 1], [0] * 4, 30)[0] == [-1, -1, -1, -1]
    # a single successor is always EXPECTED (counter == 1), whatever its count
    assert tag([0, 500, 0, 0], [0] * 4, 30)[0] == [UNEXPECTED, EXPECTED, UNEXPECTED, UNEXPECTED]
    # two successors, one in the expected interval, the other at noise level: lambda_noise = int(30*0.025)=0 < MIN
    # -> the unexpected one becomes a BREAKPOINT
    t, d = tag([30, 2, 0, 0], [0] * 4, 30)
    assert t == [EXPECTED, BREAKPOINT, UNEXPECTED, UNEXPECTED]
    assert d[0] == 0.0 and abs(d[1] - 28 / 30 ** 0.5) < 1e-12
    # high coverage: lambda_noise = int(400*0.025) = 10 >= MIN; count 3 is plausible noise -> UNEXPECTED,
    # and since exactly one EXPECTED + unexpected ones and not complex, the noise sum (3) is re-tested
    t, _ = tag([400, 3, 0, 0], [0] * 4, 400)
    assert t == [EXPECTED, UNEXPECTED, UNEXPECTED, UNEXPECTED]
    # same but the low successor is a junction k-mer (colour > 0): BREAKPOINT (Explorer.cpp:1258)
    t, _ = tag([400, 3, 0, 0], [0, 7, 0, 0], 400)
    assert t == [EXPECTED, BREAKPOINT, UNEXPECTED, UNEXPECTED]
    # no EXPECTED and exactly one BREAKPOINT: promoted to EXPECTED (:1278-1281)
    t, _ = tag([2, 0, 0, 0], [0] * 4, 400)
    assert t == [EXPECTED, UNEXPECTED, UNEXPECTED, UNEXPECTED]   # counter == 1 path
    t, _ = tag([100, 3, 0, 0], [0] * 4, 400)                      # 100 not expected from 400, 3 is noise
    assert t == [EXPECTED, UNEXPECTED, UNEXPECTED, UNEXPECTED]   # 100 -> BREAKPOINT -> promoted


def test_seq_error_threshold_kats():
    p = O.params()
    # <= 10 IN counts: plain mean with MIN_COUNT added to the sum (Read.cpp:496,511-512)
    c = np.array([5, 0, 1, 7, 9], dtype=np.uint32)
    assert L.orc_seq_error_threshold(C.byref(p), c.ctypes.data, len(c)) == ((2 + 5 + 7 + 9) / 3) * 0.025
    # > 10: trimmed [0.15m, 0.90m)
    c = np.arange(2, 22, dtype=np.uint32)          # 20 IN counts 2..21, sorted
    first, last = int(0.15 * 20), int(0.90 * 20)
    exp = ((2 + sum(range(2 + first, 2 + last))) / (last - first)) * 0.025
    assert L.orc_seq_error_threshold(C.byref(p), c.ctypes.data, len(c)) == exp


def gard(scores, dists, maxb=7):
    p = O.params(max_nb_competing_paths=maxb)
    s = np.array(scores, dtype=np.float64)
    d = np.array(dists, dtype=np.float64)
    kept = np.zeros(len(s) + maxb + 8, dtype=np.uint32)
    cx = C.c_int32()
    n = L.orc_gardening(C.byref(p), s.ctypes.data, d.ctypes.data, len(s), kept.ctypes.data, len(kept), C.byref(cx))
    return kept[:n].tolist(), bool(cx.value)


def test_gardening_kats():
    # <= MAXB paths: everything is kept, in order
    assert gard([1, 5, 3], [0.1, 0.2, 0.3]) == ([0, 1, 2], False)
    # a path that is best on both rankings (rank sum 0) is kept alone
    sc = [10, 9, 8, 7, 6, 5, 4, 3, 2]
    di = [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9]
    assert gard(sc, di) == ([0], False)
    # best score != best distance: top by score, trimmed to MAXB-1 entries (Explorer.cpp:839-851)
    di2 = list(reversed(di))
    kept, cx = gard(sc, di2)
    assert kept == [0, 1, 2, 3, 4, 5] and not cx
    # the MAXB+1 best all tie on score: complex; MAXB by distance rank, then the whole list again (:852-860)
    # (the best distance must belong to a lower-score path, otherwise a rank-sum-0 path exists)
    sc3 = [5, 5, 5, 5, 5, 5, 5, 5, 4]
    di3 = [0.9, 0.2, 0.8, 0.3, 0.7, 0.4, 0.6, 0.5, 0.1]
    kept, cx = gard(sc3, di3)
    assert cx
    assert kept == [1, 3, 5, 7, 6, 4, 2] + [1, 3, 5, 7, 6, 4, 2, 0]
    # all scores equal and the best distance among them: rank sum 0 exists, single survivor
    assert gard([5] * 9, [0.9, 0.1, 0.8, 0.2, 0.7, 0.3, 0.6, 0.4, 0.5]) == ([1], False)


def test_basic_read_stats_row_follows_the_reference():
    """Read::outputBasicReadStats (Read.cpp:418-433): span = sum of end - start + 1 over the IN regions as they stand,
    their number, the raw length and length(m_correction) — which is empty unless correct2 ran; the call sits inside
    `if (getLength() > K)` (main.cpp:262,305), so shorter reads get no row."""
    import parity_util as PU
    pair = PU.Pair(target_kmers=60_000, k=21, seed=5)
    bases, offs = pair.reads(0, 30)
    seqs = PU.seqs_of(bases, offs) + ["ACGTACGTACGTACGTACGTA", "A" * 300, "ACGT"]
    rb = "".join(seqs).encode()
    o = np.zeros(len(seqs) + 1, dtype=np.uint64)
    o[1:] = np.cumsum([len(x) for x in seqs])
    out, oo, st, rows = pair.otab.correct_batch_stats(np.frombuffer(rb, dtype=np.uint8), o, nthreads=2)
    out2, oo2, st2 = pair.otab.correct_batch(np.frombuffer(rb, dtype=np.uint8), o, nthreads=2)
    assert np.array_equal(out, out2) and np.array_equal(st, st2)
    for i, sq in enumerate(seqs):
        if len(sq) <= 21:
            assert rows[i].tolist() == [0, 0, 0, 0, 0]
            continue
        assert rows[i, 0] == 1 and rows[i, 1] == len(sq)
        if st[i] == 0:
            assert rows[i, 4] == int(oo[i + 1] - oo[i]) and rows[i, 3] >= 1 and rows[i, 2] >= rows[i, 3]
            reg, thr, ok = pair.otab.structure(sq)
            assert ok and rows[i, 3] == len(reg)          # correction moves region borders, never their number
        else:
            assert rows[i, 4] == 0
        if st[i] == 2:
            assert rows[i, 2] == 0 and rows[i, 3] == 0
