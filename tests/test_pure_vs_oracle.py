"""Host build of the product's host/device-pure pieces (talc_amd/csrc/talc_pure.h, compiled by
g++ into libtalc_pure.so) checked against the oracle and against libstdc++'s own std::sort."""
import ctypes as C
import os
import random

import numpy as np

import oracle_lib as O
from talc_amd import build as B

ORC = O.lib()


def pure():
    L = C.CDLL(B.build_pure())
    L.pure_is_expected_by_model.argtypes = [C.c_double, C.c_uint32, C.c_uint32, C.c_int]
    L.pure_is_expected_by_last_node.argtypes = [C.c_double, C.c_uint32, C.c_uint32]
    L.pure_tag_next_nodes.argtypes = [C.c_double, C.c_double, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int,
                                      C.c_void_p, C.c_void_p]
    L.pure_gnu_sort_pairs.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.pure_std_sort_pairs.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.pure_gardening.argtypes = [C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]
    L.pure_sort_anchors.argtypes = [C.c_double, C.c_void_p, C.c_void_p, C.c_int]
    L.pure_wfa_xdrop.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_void_p]
    return L


P = pure()
ORC.orc_sort_anchors.argtypes = [C.c_double, C.c_void_p, C.c_void_p, C.c_int]
ORC.orc_xdrop_right.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]


def test_count_model_matches_oracle():
    rnd = random.Random(1)
    for alpha in (2.57, 0.67, 1.96, 3.3):
        p = O.params(alpha=alpha)
        for _ in range(3000):
            cc = rnd.choice([0, 1, 2, 3, 4, 5, rnd.randint(0, 50), rnd.randint(0, 5000), rnd.randint(0, 200000)])
            nextc = rnd.choice([0, 1, 2, cc, max(0, cc - 1), cc + 1, rnd.randint(0, 60), rnd.randint(0, 300000)])
            for cl in (0, 1):
                assert P.pure_is_expected_by_model(alpha, nextc, cc, cl) == ORC.orc_is_expected_by_model(C.byref(p), nextc, cc, cl)
            assert P.pure_is_expected_by_last_node(alpha, nextc, cc) == ORC.orc_is_expected_by_last_node(C.byref(p), nextc, cc)


def test_tag_next_nodes_matches_oracle_bitwise():
    rnd = random.Random(2)
    for it in range(6000):
        minc = rnd.choice([2, 2, 2, 3, 5])
        err = rnd.choice([0.025, 0.01, 0.1])
        alpha = rnd.choice([2.57, 2.57, 1.0])
        p = O.params(min_count=minc, sr_error_rate=err, alpha=alpha)
        count = rnd.choice([0, 1, 2, 3, 5, 30, 80, 400, 5000, rnd.randint(0, 100000)])

        def c():
            return rnd.choice([0, 0, 0, 1, 2, 3, count, max(0, count - rnd.randint(0, 10)), count + rnd.randint(0, 10),
                               rnd.randint(0, 50), rnd.randint(0, 20000)])
        cnt = np.array([c(), c(), c(), c()], dtype=np.uint32)
        jc = np.array([rnd.choice([0, 0, 0, 5]) for _ in range(4)], dtype=np.uint32)
        cx = rnd.random() < 0.3
        t1, d1 = np.zeros(4, np.int32), np.zeros(4, np.float64)
        t2, d2 = np.zeros(4, np.int32), np.zeros(4, np.float64)
        P.pure_tag_next_nodes(alpha, err, minc, cnt.ctypes.data, jc.ctypes.data, count, int(cx), t1.ctypes.data, d1.ctypes.data)
        ORC.orc_tag_next_nodes(C.byref(p), cnt.ctypes.data, jc.ctypes.data, count, int(cx), t2.ctypes.data, d2.ctypes.data)
        assert t1.tolist() == t2.tolist(), (cnt, jc, count, cx)
        # distances must agree bit for bit (NaN == NaN here: compare the raw bits)
        used = [i for i in range(4) if t1[i] in (0, 7)]      # distances only exist for successors that become Trails
        assert d1.view(np.uint64)[used].tolist() == d2.view(np.uint64)[used].tolist(), (cnt, count)


def _sort_both(keys):
    n = len(keys)
    k1 = np.array(keys, dtype=np.int32)
    p1 = np.arange(n, dtype=np.int32)
    k2, p2 = k1.copy(), p1.copy()
    P.pure_gnu_sort_pairs(k1.ctypes.data, p1.ctypes.data, n)
    P.pure_std_sort_pairs(k2.ctypes.data, p2.ctypes.data, n)
    assert k1.tolist() == sorted(keys)
    return p1.tolist(), p2.tolist()


def _median3_killer(n):
    """Musser's median-of-3 killer adapted to even n: drives introsort to its depth limit so the
    heapsort fallback of the restated std::sort is exercised."""
    n -= n % 2
    k = n // 2
    a = [0] * n
    for i in range(1, k + 1):
        if i % 2 == 1:
            a[i - 1] = i
            a[i] = k + i
        a[k + i - 1] = 2 * i
    return a


def test_gnu_sort_is_libstdcxx_sort():
    rnd = random.Random(3)
    for it in range(1500):
        n = rnd.choice([0, 1, 2, 5, 15, 16, 17, 18, 31, 33, 50, 64, 100, 200, 288, 500, rnd.randint(0, 1200)])
        nd = rnd.choice([1, 2, 3, 5, 10, 50, 1000000])       # few distinct keys = many ties
        keys = [rnd.randrange(nd) for _ in range(n)]
        mode = rnd.random()
        if mode < 0.15:
            keys.sort()
        elif mode < 0.3:
            keys.sort(reverse=True)
        elif mode < 0.4 and n > 4:
            keys = [(i % 7) for i in range(n)]
        a, b = _sort_both(keys)
        assert a == b, (n, nd)
    for n in (64, 128, 500, 1000, 4096):
        a, b = _sort_both(_median3_killer(n))
        assert a == b, n
        keys = list(range(n // 2)) + list(range(n // 2))     # organ pipe-ish
        a, b = _sort_both(keys)
        assert a == b


def test_heapsort_fallback_is_libstdcxx_partial_sort():
    for f in ("pure_gnu_heapsort_pairs", "pure_std_heapsort_pairs"):
        getattr(P, f).argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    rnd = random.Random(8)
    for it in range(800):
        n = rnd.choice([0, 1, 2, 3, 4, 17, 18, 33, 100, rnd.randint(0, 600)])
        nd = rnd.choice([1, 2, 4, 20, 100000])
        keys = [rnd.randrange(nd) for _ in range(n)]
        k1 = np.array(keys, dtype=np.int32)
        p1 = np.arange(n, dtype=np.int32)
        k2, p2 = k1.copy(), p1.copy()
        P.pure_gnu_heapsort_pairs(k1.ctypes.data, p1.ctypes.data, n)
        P.pure_std_heapsort_pairs(k2.ctypes.data, p2.ctypes.data, n)
        assert k1.tolist() == sorted(keys) and p1.tolist() == p2.tolist(), (n, nd)


def test_gardening_matches_oracle():
    rnd = random.Random(4)
    for it in range(4000):
        maxb = rnd.choice([7, 7, 5, 10])
        n = rnd.choice([1, 2, 6, 7, 8, 9, 12, 20, 40, 75, 150, 200, rnd.randint(1, 220)])
        ns = rnd.choice([1, 2, 3, 5, 20, 1000])
        nd = rnd.choice([1, 2, 3, 10, 1000])
        scores = np.array([float(-rnd.randrange(ns) * 3) for _ in range(n)])
        dists = np.array([rnd.randrange(nd) * 0.37 for _ in range(n)])
        if rnd.random() < 0.1:
            dists[rnd.randrange(n)] = float("inf")
        p = O.params(max_nb_competing_paths=maxb)
        k1 = np.zeros(n + maxb + 8, np.uint32)
        k2 = np.zeros(n + maxb + 8, np.uint32)
        c1, c2 = C.c_int32(), C.c_int32()
        n1 = P.pure_gardening(maxb, n, scores.ctypes.data, dists.ctypes.data, k1.ctypes.data, C.byref(c1))
        n2 = ORC.orc_gardening(C.byref(p), scores.ctypes.data, dists.ctypes.data, n, k2.ctypes.data, len(k2), C.byref(c2))
        assert n1 == n2 and c1.value == c2.value and k1[:n1].tolist() == k2[:n2].tolist(), (maxb, n, scores, dists)


def test_anchor_ordering_matches_std_sort():
    rnd = random.Random(6)
    for it in range(2000):
        n = rnd.choice([1, 2, 3, 5, 16, 17, 30, 100, rnd.randint(1, 300)])
        cc = rnd.choice([80.0, 1234.5, 12.0, 40000.0])
        pos = np.arange(n, dtype=np.uint32)
        cnt = np.array([rnd.choice([2, 3, 30, 79, 80, 81, rnd.randint(0, 3000)]) for _ in range(n)], dtype=np.uint32)
        p1, c1, p2, c2 = pos.copy(), cnt.copy(), pos.copy(), cnt.copy()
        P.pure_sort_anchors(cc, p1.ctypes.data, c1.ctypes.data, n)
        ORC.orc_sort_anchors(cc, p2.ctypes.data, c2.ctypes.data, n)
        assert p1.tolist() == p2.tolist() and c1.tolist() == c2.tolist()


def _mutate(rnd, s, rate, alphabet):
    out = []
    for ch in s:
        x = rnd.random()
        if x < rate / 3:
            out.append(rnd.choice(alphabet))
        elif x < 2 * rate / 3:
            out.append(ch)
            out.append(rnd.choice(alphabet))
        elif x >= rate:
            out.append(ch)
    return "".join(out)


def test_wavefront_xdrop_equals_antidiagonal_xdrop():
    """talc_wfa.h (the algorithm wave_xdrop_wfa runs on the device) against the oracle's restatement of SeqAn's
    _extendSeedGappedXDropOneDirection for the unit-cost scoring: same decision, same extension, same score —
    on similar and unrelated segments, tiny alphabets (long match runs), segments ending inside / beyond each
    other (every 'longest extension' branch) and x from -1 to far beyond the segment lengths."""
    rnd = random.Random(11)
    seen = {0: 0, 1: 0}
    for _ in range(12000):
        alphabet = rnd.choice(["ACGT", "AC", "A", "ACGT"])
        n = rnd.choice([1, 2, 3, 5, 8, 13, 30, 60, 120, 300])
        q = "".join(rnd.choice(alphabet) for _ in range(n))
        if rnd.random() < 0.7:
            d = _mutate(rnd, q, rnd.choice([0, 0.02, 0.1, 0.2, 0.4]), alphabet)
        else:
            d = "".join(rnd.choice(alphabet) for _ in range(rnd.choice([1, 2, 4, 9, 40, 100])))
        r = rnd.random()
        if r < 0.25:
            d = d[: rnd.randrange(len(d) + 1)]
        elif r < 0.5:
            d = d + "".join(rnd.choice(alphabet) for _ in range(rnd.randint(1, 30)))
        elif r < 0.6:
            q = q[: rnd.randrange(len(q) + 1)]
        if not q or not d:
            continue
        x = rnd.choice([-1, 0, 1, 2, 3, 4, 5, 7, 10, 15, 25, 40, 80, 400])
        want = np.zeros(4, dtype=np.int32)
        got = np.zeros(4, dtype=np.int32)
        ORC.orc_xdrop_right(q.encode(), d.encode(), 0, -1, -1, x, want.ctypes.data)
        P.pure_wfa_xdrop(q.encode(), len(q), d.encode(), len(d), x, got.ctypes.data)
        assert want[0] == got[0], (q, d, x, want.tolist(), got.tolist())
        if want[0]:
            assert (want[1:] == got[1:]).all(), (q, d, x, want.tolist(), got.tolist())
        seen[int(want[0])] += 1
    assert seen[1] > 1000


# ---------------------------------------------------------------- bit-vector LCS / edit distance (wave_lcs_bitpar, wave_edit_bitpar)
def _carries(G, P, W):
    """Carry into each of W words from per-word generate / propagate bits: the formula of talc_wave.h."""
    Y = (G << 1) & ((1 << W) - 1)
    return (((Y + P) & ((1 << (W + 1)) - 1)) ^ P | Y) & ((1 << W) - 1)


def test_cross_word_carry_formula_is_exact():
    """((G << 1) + P) ^ P | (G << 1) gives the carry into every word of a multi-word addition when a word either
    generates a carry or propagates one, never both — exhaustively for 10 words."""
    W = 10
    for G in range(1 << W):
        for P in range(1 << W):
            if G & P:
                continue
            c, exp = 0, 0
            for l in range(W):
                if c:
                    exp |= 1 << l
                c = ((G >> l) & 1) | (((P >> l) & 1) & c)
            assert _carries(G, P, W) == exp, (bin(G), bin(P))


def _multiword_add(V, U, nw):
    """64-bit words, carries resolved with the ballot formula (what the lanes do)."""
    M64 = (1 << 64) - 1
    S = [(V[i] + U[i]) & M64 for i in range(nw)]
    G = sum(1 << i for i in range(nw) if S[i] < V[i])
    P = sum(1 << i for i in range(nw) if S[i] == M64)
    C = _carries(G, P, nw)
    return [(S[i] + ((C >> i) & 1)) & M64 for i in range(nw)]


def _lcs_bitpar(a, b):
    m = len(a)
    nw = (m + 63) // 64
    M64 = (1 << 64) - 1
    pm = {c: [0] * nw for c in "ACGTN"}
    for i, ch in enumerate(a):
        pm[ch][i // 64] |= 1 << (i % 64)
    V = [M64] * nw
    for ch in b:
        Mm = pm[ch]
        U = [V[i] & Mm[i] for i in range(nw)]
        S = _multiword_add(V, U, nw)
        V = [S[i] | (V[i] & ~Mm[i] & M64) for i in range(nw)]
    z = 0
    for i in range(m):
        z += 1 - ((V[i // 64] >> (i % 64)) & 1)
    return z


def _edit_bitpar(a, b):
    m = len(a)
    nw = (m + 63) // 64
    M64 = (1 << 64) - 1
    pm = {c: [0] * nw for c in "ACGTN"}
    for i, ch in enumerate(a):
        pm[ch][i // 64] |= 1 << (i % 64)
    Pv, Mv, score = [M64] * nw, [0] * nw, m
    tw, tb = (m - 1) // 64, (m - 1) % 64
    for ch in b:
        Eq = pm[ch]
        Xv = [Eq[i] | Mv[i] for i in range(nw)]
        S = _multiword_add(Pv, [Eq[i] & Pv[i] for i in range(nw)], nw)
        Xh = [(S[i] ^ Pv[i]) | Eq[i] for i in range(nw)]
        Ph = [Mv[i] | (~(Xh[i] | Pv[i]) & M64) for i in range(nw)]
        Mh = [Pv[i] & Xh[i] for i in range(nw)]
        score += (Ph[tw] >> tb) & 1
        score -= (Mh[tw] >> tb) & 1
        pc, mc = 1, 0                      # the border of a global alignment: +1 per column in row 0
        for i in range(nw):
            npc, nmc = Ph[i] >> 63, Mh[i] >> 63
            Ph[i] = ((Ph[i] << 1) | pc) & M64
            Mh[i] = ((Mh[i] << 1) | mc) & M64
            pc, mc = npc, nmc
        Pv = [Mh[i] | (~(Xv[i] | Ph[i]) & M64) for i in range(nw)]
        Mv = [Ph[i] & Xv[i] for i in range(nw)]
    return score


def test_bit_vector_lcs_and_edit_distance_equal_the_oracle_alignments():
    """The recurrences wave_lcs_bitpar / wave_edit_bitpar run (Hyyro's LCS, Myers' edit distance, words chained by
    the carry formula) restated in Python, against the oracle's localAlignment(1,0,0) and -globalAlignment(0,-1,-1)."""
    rnd = random.Random(77)
    L = O.lib()
    for it in range(60):
        n = rnd.choice([1, 5, 63, 64, 65, 130, 200, 333])
        a = [rnd.choice("ACGTN" if rnd.random() < 0.1 else "ACGT") for _ in range(n)]
        if rnd.random() < 0.7:
            b = []
            for ch in a:
                x = rnd.random()
                if x < 0.05:
                    b.append(rnd.choice("ACGT"))
                elif x < 0.10:
                    b.append(ch)
                    b.append(rnd.choice("ACGT"))
                elif x >= 0.15:
                    b.append(ch)
            b = b or ["A"]
        else:
            b = [rnd.choice("ACGT") for _ in range(rnd.choice([1, 7, 64, 150]))]
        a, b = "".join(a), "".join(b)
        exp_l = L.orc_global_alignment(a.encode(), b.encode(), 1, 0, 0, 0, 0, 0, 0)
        exp_e = -L.orc_global_alignment(a.encode(), b.encode(), 0, -1, -1, 0, 0, 0, 0)
        assert _lcs_bitpar(a, b) == exp_l and _lcs_bitpar(b, a) == exp_l, (len(a), len(b))
        assert _edit_bitpar(a, b) == exp_e and _edit_bitpar(b, a) == exp_e, (len(a), len(b))


def _lcs_blocks(a, b, B):
    """wave_lcs_bitpar's block form with blocks of B columns: block after block over all rows, one carry bit per row
    handed from a block to the next (the carry out of the block's top bit)."""
    m, carry, z = len(a), [0] * len(b), 0
    for c0 in range(0, m, B):
        mb = min(B, m - c0)
        full = (1 << B) - 1
        pm = {c: 0 for c in "ACGTN"}
        for i in range(mb):
            pm[a[c0 + i]] |= 1 << i
        V = full
        last = c0 + B >= m
        for j, ch in enumerate(b):
            Mm = pm[ch]
            S = V + (V & Mm) + carry[j]
            if not last:
                carry[j] = S >> B
            S &= full
            V = S | (V & ~Mm & full)
        z += sum(1 - ((V >> i) & 1) for i in range(mb))
    return z


def _edit_blocks(a, b, B):
    """wave_edit_bitpar's block form (Myers' block formulation): the horizontal delta of a block's last row, +1 / 0 / -1 per
    consumed base, enters the next block's first row; a -1 entering acts like a match in that row."""
    m = len(a)
    hp, hm = [1] * len(b), [0] * len(b)      # the border: +1 per column in row 0
    score = m
    for c0 in range(0, m, B):
        mb = min(B, m - c0)
        full = (1 << B) - 1
        pm = {c: 0 for c in "ACGTN"}
        for i in range(mb):
            pm[a[c0 + i]] |= 1 << i
        last = c0 + B >= m
        top = (m - 1 - c0) if last else (B - 1)
        Pv, Mv = full, 0
        for j, ch in enumerate(b):
            Eq = pm[ch]
            Xv = Eq | Mv
            Eqx = Eq | (1 if hm[j] else 0)
            Xh = ((((Eqx & Pv) + Pv) & full) ^ Pv) | Eqx
            Ph = Mv | (~(Xh | Pv) & full)
            Mh = Pv & Xh
            up, dn = (Ph >> top) & 1, (Mh >> top) & 1
            Ph = ((Ph << 1) | hp[j]) & full
            Mh = ((Mh << 1) | hm[j]) & full
            if last:
                score += up - dn
            else:
                hp[j], hm[j] = up, dn
            Pv = Mh | (~(Xv | Ph) & full)
            Mv = Ph & Xv
    return score


def test_bit_vector_block_hand_over_is_exact():
    """Sequences beyond one 4096-column block: the block forms (blocks of 4, 7 and 64 columns here, so that every case of
    the hand-over occurs many times) against the oracle's alignments, both argument orders."""
    rnd = random.Random(78)
    L = O.lib()
    for it in range(120):
        n = rnd.choice([1, 3, 4, 5, 8, 9, 29, 64, 65, 130, 200])
        a = [rnd.choice("ACGTN" if rnd.random() < 0.1 else "ACGT") for _ in range(n)]
        if rnd.random() < 0.7:
            b = []
            for ch in a:
                x = rnd.random()
                if x < 0.07:
                    b.append(rnd.choice("ACGT"))
                elif x < 0.14:
                    b.append(ch)
                    b.append(rnd.choice("ACGT"))
                elif x >= 0.21:
                    b.append(ch)
            b = b or ["A"]
        else:
            b = [rnd.choice("ACGT") for _ in range(rnd.choice([1, 7, 64, 150]))]
        a, b = "".join(a), "".join(b)
        exp_l = L.orc_global_alignment(a.encode(), b.encode(), 1, 0, 0, 0, 0, 0, 0)
        exp_e = -L.orc_global_alignment(a.encode(), b.encode(), 0, -1, -1, 0, 0, 0, 0)
        for B in (4, 7, 64):
            assert _lcs_blocks(a, b, B) == exp_l and _lcs_blocks(b, a, B) == exp_l, (len(a), len(b), B)
            assert _edit_blocks(a, b, B) == exp_e and _edit_blocks(b, a, B) == exp_e, (len(a), len(b), B)


def _nw_rows(ref, cand, i0, m, row):
    """wave_nw_rows restated: rows i0+1 .. m of the free-begin (4, -3, -2) matrix of `cand` (rows) against the WHOLE
    `ref` (columns), continued from `row` = row i0 (None: row 0, all zero); returns row m."""
    n = len(ref)
    prev = list(row) if row is not None else [0] * (n + 1)
    for i in range(i0 + 1, m + 1):
        cur = [0] * (n + 1)                      # column 0 is free
        for j in range(1, n + 1):
            d = prev[j - 1] + (4 if ref[j - 1] == cand[i - 1] else -3)
            cur[j] = max(d, prev[j] - 2, cur[j - 1] - 2)
        prev = cur
    return prev


def test_alignment_rows_continued_equal_the_alignments_from_scratch():
    """scoreBridges (Explorer.cpp:689-706) aligns a Trail of K + step bases against the first K + step + WINDOW bases of
    the reference every CHECK_INTERVAL steps; the device keeps the Trail's last matrix row over the whole reference and
    only adds the new rows.  A growing candidate, truncations growing with it, rows continued in steps of 1..9 bases:
    the kept row's entry at the truncation's length equals the oracle's alignment (free begin, end gaps charged) of the
    candidate's prefix against the truncated reference, every time — the same for a copy that branches off and grows
    differently."""
    rnd = random.Random(91)
    L = O.lib()
    for it in range(25):
        n = rnd.choice([30, 64, 65, 200, 330])
        ref = "".join(rnd.choice("ACGT") for _ in range(n))
        cand = []
        for ch in ref:
            x = rnd.random()
            if x < 0.05:
                cand.append(rnd.choice("ACGT"))
            elif x < 0.10:
                cand.append(ch)
                cand.append(rnd.choice("ACGT"))
            elif x >= 0.15:
                cand.append(ch)
        cand = "".join(cand) + "".join(rnd.choice("ACGT") for _ in range(20))
        window = rnd.choice([5, 10, 15])
        K = 21
        row, i0 = None, 0
        branch = None
        m = K
        while m <= len(cand):
            row = _nw_rows(ref, cand, i0, m, row)
            i0 = m
            tlen = min(n, m + window)
            exp = L.orc_global_alignment(ref[:tlen].encode(), cand[:m].encode(), 4, -3, -2, 1, 1, 0, 0)
            assert row[tlen] == exp, (n, m, tlen, row[tlen], exp)
            if branch is None and m > len(cand) // 2:      # a copy branches off here: other bases from now on
                branch = (list(row), m, cand[:m] + "".join(rnd.choice("ACGT") for _ in range(40)))
            m += rnd.choice([1, 3, 6, 6, 9])
        if branch:
            brow, bi0, bcand = branch
            for m2 in range(bi0 + 6, len(bcand) + 1, 6):
                brow = _nw_rows(ref, bcand, bi0, m2, brow)
                bi0 = m2
                tlen = min(n, m2 + window)
                exp = L.orc_global_alignment(ref[:tlen].encode(), bcand[:m2].encode(), 4, -3, -2, 1, 1, 0, 0)
                assert brow[tlen] == exp, ("branch", n, m2, tlen)
