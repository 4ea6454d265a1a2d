// Seed extension with an x-drop for the unit-cost scoring (match 0, mismatch -1, gap -1), as a
// furthest-reaching wavefront recurrence.  Scalar statement of the algorithm the device runs one
// diagonal per lane (talc_wave.h: wave_xdrop_wfa); host builds use it to fuzz the derivation against
// the oracle's anti-diagonal x-drop (tests/test_pure_vs_oracle.py).
//
// Why it is the same function as SeqAn's _extendSeedGappedXDropOneDirection (the oracle's
// gappedXDropOneDirection), for these scores:
//  * all scores are <= 0, so `best` stays 0 and a cell is kept iff its value >= -x.  A cell's value is
//    minus the cost D' of the cheapest path from the origin through kept cells, and a path's prefix
//    costs never exceed its total: kept cells = { D' <= x }, where D' is the edit distance in the grid
//    without the border cells (0,a), (a,0) that the x-drop does not initialise (a >= x; a == 1 is
//    initialised iff x >= 1).
//  * the band [minCol, maxCol) is trimmed only over cells whose three predecessors are all dropped, so
//    no kept cell ever lies outside it.
//  * D' is non-decreasing along a diagonal, so the kept cells of diagonal k = col - row are a prefix,
//    described by F[k] = the last anti-diagonal (col + row) still kept: the WFA recurrence
//    F_e[k] = extend(max(F_{e-1}[k] + 2, F_{e-1}[k-1] + 1, F_{e-1}[k+1] + 1)) for e = 1..x.
//  * the loop of the original ends on the first anti-diagonal T >= 2 after which no kept cell has a
//    successor inside the matrix; with A = the last anti-diagonal holding a kept cell that is T = A
//    (the far corner is kept), T = A + 1 (every successor of the cells on A falls outside the
//    matrix) or T = A + 2, and the "longest extension" tests of the original read cells of the
//    anti-diagonals T, T-1, T-2 that are the last kept cells of their diagonals, whose value is minus
//    the first level e that reached them.
#pragma once
#include <climits>
#include <cstdint>
#include <vector>

namespace talc {

struct WfaResult { int moved; int extCols; int extRows; int score; };

// q = query segment (columns), d = database segment (rows)
inline WfaResult wfa_xdrop_scalar(const uint8_t* q, int qlen, const uint8_t* d, int dlen, int x) {
  WfaResult res = {0, 0, 0, 0};
  if (qlen == 0 || dlen == 0) return res;
  const int NEG = -(1 << 29);
  const int bmax = x >= 2 ? x - 1 : (x == 1 ? 1 : 0);
  const int X = x < 0 ? 0 : x;
  const int kmin = -(X < dlen ? X : dlen), kmax = (X < qlen ? X : qlen);
  const int nd = kmax - kmin + 1;
  std::vector<int> F(nd + 2, NEG), G(nd + 2, NEG), E(nd + 2, 0), amin(nd + 2), amax(nd + 2);
  auto idx = [&](int k) { return k - kmin + 1; };
  for (int k = kmin; k <= kmax; ++k) {
    const int ak = k < 0 ? -k : k;
    amin[idx(k)] = ak + (ak > bmax ? 2 : 0);
    const int a1 = 2 * qlen - k, a2 = 2 * dlen + k;
    amax[idx(k)] = a1 < a2 ? a1 : a2;
  }
  auto extend = [&](int a, int k) {
    int c = (a + k) / 2, r = (a - k) / 2;
    while (c < qlen && r < dlen && q[c] == d[r]) { ++c; ++r; }
    return c + r;
  };
  const int corner = qlen + dlen, kc = qlen - dlen;
  F[idx(0)] = x >= 0 ? extend(0, 0) : 0;
  bool cornerHit = (kc >= kmin && kc <= kmax && F[idx(kc)] == corner);
  int cornerE = 0;
  for (int e = 1; e <= x && !cornerHit; ++e) {
    for (int k = kmin; k <= kmax; ++k) {
      const int i = idx(k);
      const int fm = F[i], fl = F[i - 1], fr = F[i + 1];
      int best = NEG;
      const int v[3] = {fm + 2, fl + 1, fr + 1};
      for (int t = 0; t < 3; ++t)
        if (v[t] >= amin[i] && v[t] <= amax[i] && v[t] > best) best = v[t];
      int nv = fm;
      if (best > fm) { nv = extend(best, k); E[i] = e; }
      G[i] = nv;
    }
    F.swap(G);
    if (kc >= kmin && kc <= kmax && F[idx(kc)] == corner) { cornerHit = true; cornerE = e; }
  }
  if (cornerHit) { res.moved = 1; res.extCols = qlen; res.extRows = dlen; res.score = -cornerE; return res; }
  // ---- where the original stops, and which cell it reports
  int A = 0;
  for (int k = kmin; k <= kmax; ++k) if (F[idx(k)] > A) A = F[idx(k)];
  const int cols = qlen + 1, rows = dlen + 1;
  // kept columns of anti-diagonal a: { (a + k) / 2 : k = a (mod 2), F[k] >= a }
  auto minmaxS = [&](int a, bool dropTopBorder, int& mn, int& mx) {
    mn = INT_MAX; mx = INT_MIN;
    if (a < 0) return;
    for (int k = kmin; k <= kmax; ++k) {
      if (((k - a) & 1) || F[idx(k)] < a) continue;
      const int c = (a + k) / 2;
      if (dropTopBorder && c == a) continue;
      if (c < mn) mn = c;
      if (c > mx) mx = c;
    }
  };
  auto kept = [&](int a, int c) {
    const int k = 2 * c - a;
    return c >= 0 && a - c >= 0 && k >= kmin && k <= kmax && F[idx(k)] >= a;
  };
  auto firstMax = [&](int a) {   // first maximum of the anti-diagonal a: least level, then least column
    int bestE = INT_MAX, bestK = 0;
    for (int k = kmin; k <= kmax; ++k) {
      if (((k - a) & 1) || F[idx(k)] != a) continue;
      if (E[idx(k)] < bestE) { bestE = E[idx(k)]; bestK = k; }
    }
    if (bestE == INT_MAX) return;
    res.moved = 1; res.extCols = (a + bestK) / 2; res.extRows = (a - bestK) / 2; res.score = -bestE;
  };
  bool early = false;   // T == A + 1
  if (A + 1 >= 2) {
    int mn, mx; minmaxS(A, false, mn, mx);
    const int a = A + 1;
    const int lo = (1 + mn > a + 2 - rows) ? 1 + mn : a + 2 - rows;
    const int hi = (2 + mx < cols) ? 2 + mx : cols;
    early = lo >= hi;
  }
  if (!early) { firstMax(A); return res; }
  // T = A + 1: maxCol of the anti-diagonal A
  int maxColA;
  if (A >= 2) {
    int mn1, mx1, mn2, mx2;
    minmaxS(A - 1, true, mn1, mx1); minmaxS(A - 2, false, mn2, mx2);
    const int cm = mx1 > mx2 ? mx1 : mx2;
    maxColA = (2 + cm < cols) ? 2 + cm : cols;
  } else {
    maxColA = 1;
  }
  const int c2 = maxColA - 1;
  auto take = [&](int a, int c) {
    const int k = 2 * c - a;
    res.moved = 1; res.extCols = c; res.extRows = a - c; res.score = -E[idx(k)];
  };
  if (kept(A, c2)) { take(A, c2); return res; }
  if (A >= 2 && kept(A, c2 - 1)) { take(A, c2 - 1); return res; }
  firstMax(A - 1);
  return res;
}

}  // namespace talc
