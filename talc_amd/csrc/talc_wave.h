// talc_wave.h — wavefront-cooperative primitives for the path-search kernel (gfx950, wave64).
//
// Execution model: ONE 64-lane wavefront per workgroup works on ONE long read.  Control flow is
// wave-uniform (every lane evaluates the same scalar decisions); the 64 lanes are spent on
//   * dynamic-programming sweeps (Needleman-Wunsch rows lane-skewed, x-drop anti-diagonals),
//   * k-mer window searches (cycle detection), sequence copies, table probes.
// All sequences are in "growth order": index 0 is the anchor end, the path grows at the back,
// whatever the walking direction (for a LEFT walk the buffers hold the reversed text), so one
// set of routines serves both directions.
#pragma once
#include <limits.h>

#include "talc_common.h"

namespace talc {

#define WSYNC() __syncthreads()   /* one wave per workgroup: orders the wave's own LDS/global traffic */

// the lane's index from the hardware (two VALU ops, no input register): threadIdx.x would have to be carried through
// every call in a callee-saved vector register
TALC_D int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
// the same, opaque to the optimiser: the copy loops below are unrolled eight-fold, and with a transparent lane index
// their eight per-lane offsets are hoisted out of whatever loop encloses the (inlined) copy and then spilled
TALC_D int lane_id_here() { int l = lane_id(); asm volatile("" : "+v"(l)); return l; }
// whole-wave reductions as DPP VALU ops (row_shr 1/2/4/8 inside each row of 16 lanes, then row_bcast 15 / 31 across
// the rows: lane 63 ends up with the result) instead of six ds_bpermute round trips
#define TALC_WAVE_REDUCE(v, OP, IDENT)                                                                   \
  do {                                                                                                   \
    v = OP(v, __builtin_amdgcn_update_dpp(IDENT, v, 0x111, 0xF, 0xF, false));                            \
    v = OP(v, __builtin_amdgcn_update_dpp(IDENT, v, 0x112, 0xF, 0xF, false));                            \
    v = OP(v, __builtin_amdgcn_update_dpp(IDENT, v, 0x114, 0xF, 0xF, false));                            \
    v = OP(v, __builtin_amdgcn_update_dpp(IDENT, v, 0x118, 0xF, 0xF, false));                            \
    v = OP(v, __builtin_amdgcn_update_dpp(IDENT, v, 0x142, 0xA, 0xF, false));                            \
    v = OP(v, __builtin_amdgcn_update_dpp(IDENT, v, 0x143, 0xC, 0xF, false));                            \
  } while (0)
TALC_D int wave_max_i32(int v) {
  TALC_WAVE_REDUCE(v, max, INT_MIN);
  return __builtin_amdgcn_readlane(v, 63);
}
TALC_D int talc_or_i32(int a, int b) { return a | b; }
TALC_D int talc_add_i32(int a, int b) { return a + b; }
TALC_D unsigned wave_or_u32(unsigned v) {
  int w = (int)v;
  TALC_WAVE_REDUCE(w, talc_or_i32, 0);
  return (unsigned)__builtin_amdgcn_readlane(w, 63);
}
TALC_D unsigned wave_max_u32(unsigned v) {
  int w = (int)(v ^ 0x80000000u);   // order-preserving map to signed
  TALC_WAVE_REDUCE(w, max, INT_MIN);
  return (unsigned)__builtin_amdgcn_readlane(w, 63) ^ 0x80000000u;
}
TALC_D unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += (unsigned long long)__shfl_xor((long long)v, off, 64);
  return v;
}
TALC_D int bcast_i32(int v, int src) { return __shfl(v, src, 64); }
// whole-wave lane moves as one DPP VALU op (gfx9 wave_shr:1 / wave_ror:1) instead of ds_bpermute:
// lane L receives lane L-1's value; lane 0 keeps `self` (shr) or receives lane 63's (ror).
TALC_D int lane_shr1(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x138, 0xF, 0xF, false); }
// (a rotation gives every lane a source, so the "old" operand is never used: 0 with bound_ctrl lets the compiler drop the
//  copy it would otherwise make for the tied operand — one VALU instruction per rotation, two per wavefront level)
TALC_D int lane_ror1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x13C, 0xF, 0xF, true); }
// value of lane `src` for a wave-uniform src: v_readlane (SGPR result) instead of ds_bpermute
TALC_D int lane_get(int v, int src) { return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(src)); }
// tell the compiler a value is wave-uniform so that everything derived from it runs on the scalar unit
TALC_D int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <class T> TALC_D T* uni_ptr(T* p) {
  const unsigned long long a = (unsigned long long)p;
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
  return (T*)(((unsigned long long)hi << 32) | lo);
}
// lane `lane` of `old` replaced by the wave-uniform `val` (v_writelane_b32; clang has no builtin for it)
extern "C" __device__ int talc_llvm_writelane(int val, int lane, int old) __asm("llvm.amdgcn.writelane.i32");
TALC_D int lane_set(int old, int val, int lane) { return talc_llvm_writelane(val, lane, old); }
TALC_D unsigned long long uni64(unsigned long long v) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}
// (the ballot builtin, not __ballot(): that one compares the predicate widened to an int with zero — a v_cndmask and a v_cmp
//  per call on top of the compare that made the predicate; the builtin hands back the compare's own lane mask)
TALC_D unsigned long long ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// dst[0..n) = src[0..n); both 16-byte aligned, n arbitrary (tail by bytes).
TALC_D void wave_copy(uint8_t* __restrict__ dst_, const uint8_t* __restrict__ src_, uint32_t n) {
  const int l = lane_id_here();
  const uint32_t nv = n >> 4;
  gu8 dst = (gu8)dst_; gcu8 src = (gcu8)src_;
  const v4u32 TALC_AS1* s4 = (const v4u32 TALC_AS1*)src;
  v4u32 TALC_AS1* d4 = (v4u32 TALC_AS1*)dst;
  for (uint32_t i = l; i < nv; i += 64) d4[i] = s4[i];
  for (uint32_t i = (nv << 4) + l; i < n; i += 64) dst[i] = src[i];
}
// unaligned byte copy, optional reversal of the source range: dst[i] = src[rev ? n-1-i : i]
TALC_D void wave_copy_bytes(uint8_t* __restrict__ dst_, const uint8_t* __restrict__ src_, uint32_t n, bool rev) {
  gu8 dst = (gu8)dst_; gcu8 src = (gcu8)src_;
  for (uint32_t i = (uint32_t)lane_id_here(); i < n; i += 64) dst[i] = rev ? src[n - 1 - i] : src[i];
}

// ------------------------------------------------------------------ Needleman-Wunsch score
// globalAlignment(score-only) with Score<int,Simple>(match, mismatch, gap), linear gaps
// (reference call sites: Trail.cpp:166,171,422; Trajectory.cpp:413), and with gap = 0,
// mismatch = 0, match = 1 the Smith-Waterman optimum the reference uses as LCS length
// (localAlignment(1,0,0), Trajectory.cpp:368,525): with non-negative scores and free gaps the
// local optimum equals the global one.
// freeBegin: AlignConfig<true,true,false,false> (first row and column 0) — in growth order this
// serves both Trail::Overlapscore configurations (Trail.cpp:162-173).
// H spans the DP columns, which are dealt to the lanes in blocks of B = ceil(n/64); lane l
// sweeps row (t - l) at time step t and hands its block's last cell to lane l+1 by shuffle.
// `row` holds n+1 ints (LDS or global).  Returns D[m][n]; *cells gets += n*m.
TALC_D int wave_nw(const uint8_t* __restrict__ H_, int n, const uint8_t* __restrict__ V_, int m, int match, int mismatch,
                   int gap, bool freeBegin, int* row, unsigned long long& cells) {
  gcu8 H = (gcu8)H_; gcu8 V = (gcu8)V_;
  const int l = lane_id();
  cells += (unsigned long long)n * (unsigned long long)m;
  if (n == 0) return freeBegin ? 0 : m * gap;
  if (m == 0) return freeBegin ? 0 : n * gap;
  const int B = (n + 63) >> 6;
  const int nl = (n + B - 1) / B;                 // lanes that own at least one column
  for (int j = l; j <= n; j += 64) row[j] = freeBegin ? 0 : j * gap;
  WSYNC();
  const int j0 = l * B + 1;                       // first own column
  const int j1 = min(n, j0 + B - 1);              // last own column
  int lastOut = 0, prevLastOut = (j1 >= j0) ? (freeBegin ? 0 : j1 * gap) : 0;
  // before a lane's first row, its "row i-1" hand-off is the row-0 value of its last column
  lastOut = prevLastOut;
  const int T = m + nl - 1;
  for (int t = 1; t <= T; ++t) {
    // values of the left neighbour: it finished row (t-1)-(l-1) = i in the previous step
    int nbLast = __shfl_up(lastOut, 1, 64);
    int nbPrev = __shfl_up(prevLastOut, 1, 64);
    const int i = t - l;
    if (l < nl && i >= 1 && i <= m) {
      int left, diag;
      if (l == 0) { left = freeBegin ? 0 : i * gap; diag = freeBegin ? 0 : (i - 1) * gap; }
      else { left = nbLast; diag = nbPrev; }
      const uint8_t vb = V[i - 1];
      int v = left;
      for (int j = j0; j <= j1; ++j) {
        const int up = row[j];
        const int d = diag + ((H[j - 1] == vb) ? match : mismatch);
        v = max(d, max(up + gap, left + gap));
        row[j] = v;
        diag = up;
        left = v;
      }
      prevLastOut = diag;   // D[i-1][j1]
      lastOut = v;          // D[i][j1]
    } else if (l < nl && i == 0) {
      // hand-off for the neighbour's first row: row 0 of the last own column
      prevLastOut = lastOut = freeBegin ? 0 : j1 * gap;
    }
  }
  WSYNC();
  return row[n];
}

// ------------------------------------------------------------------ gapped x-drop extension
// SeqAn2 _extendSeedGappedXDropOneDirection (seeds_extension.h) for EXTEND_RIGHT on
// querySeg (DP columns, V dimension) and databaseSeg (DP rows, H dimension), Score(match,
// mismatch, gap), as used by extendSeed(..., Score(0,-1,-1), xdrop, GappedXDrop())
// (Trail.cpp:372-373,390-391; EXTEND_LEFT is the same computation on reversed segments, which
// is what growth order gives us).  Anti-diagonals are rolled through d1,d2,d3 (each >= cols+2
// ints); the lanes span the live columns [minCol, maxCol) of the current anti-diagonal.
// Outputs the "longest extension" (extCols on the query, extRows on the database) and returns
// whether the seed moves.
template <class IP> struct XDropBufT { IP d1; IP d2; IP d3; };
typedef XDropBufT<int*> XDropBuf;
#ifndef LSYNC
#define LSYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
#endif

// IP = int* (HBM arrays, full barrier) or int AS3* (LDS arrays: a wavefront-scope fence orders the wave's own
// LDS traffic, no vmcnt wait).
template <class IP, bool LDS>
TALC_D bool wave_xdrop(const uint8_t* __restrict__ querySeg_, int qlen, const uint8_t* __restrict__ dbSeg_, int dlen,
                       int match, int mismatch, int gapCost, int scoreDropOff, XDropBufT<IP> buf, int& extCols, int& extRows,
                       unsigned long long& cells) {
  gcu8 querySeg = (gcu8)querySeg_; gcu8 dbSeg = (gcu8)dbSeg_;
  const int l = lane_id();
  const int cols = qlen + 1, rows = dlen + 1;
  extCols = extRows = 0;
  if (rows == 1 || cols == 1) return false;
  const int undefined = INT_MIN - gapCost;
  IP antiDiag1 = buf.d1; IP antiDiag2 = buf.d2; IP antiDiag3 = buf.d3;
  int len1 = 0, len2 = 1, len3 = 2;
  int minCol = 1, maxCol = 2;
  int offset1 = 0, offset2 = 0, offset3 = 0;
  if (l == 0) {
    antiDiag2[0] = 0;
    if (-gapCost > scoreDropOff) { antiDiag3[0] = undefined; antiDiag3[1] = undefined; }
    else { antiDiag3[0] = gapCost; antiDiag3[1] = gapCost; }
  }
  if (LDS) LSYNC(); else WSYNC();
  int antiDiagNo = 1;
  int best = 0;
  unsigned long long ncell = 0;
  while (minCol < maxCol) {
    ++antiDiagNo;
    { IP t = antiDiag1; antiDiag1 = antiDiag2; antiDiag2 = antiDiag3; antiDiag3 = t; }
    len1 = len2; len2 = len3;
    offset1 = offset2; offset2 = offset3; offset3 = minCol - 1;
    len3 = maxCol + 1 - offset3;
    const int minScore = best - scoreDropOff;
    if (l == 0) {  // _initAntiDiag3
      int v0 = undefined, vN = undefined;
      if (antiDiagNo * gapCost > minScore) {
        if (offset3 == 0) v0 = antiDiagNo * gapCost;
        if (antiDiagNo - maxCol == 0) vN = antiDiagNo * gapCost;
      }
      antiDiag3[0] = v0;
      antiDiag3[maxCol - offset3] = vN;
    }
    int antiDiagBest = antiDiagNo * gapCost;
    for (int col = minCol + l; col < maxCol; col += 64) {
      const int i3 = col - offset3, i2 = col - offset2, i1 = col - offset1;
      const int queryPos = col - 1, dbPos = antiDiagNo - col - 1;
      int tmp = max(antiDiag2[i2 - 1], antiDiag2[i2]) + gapCost;
      const int s = (querySeg[queryPos] == dbSeg[dbPos]) ? match : mismatch;
      tmp = max(tmp, antiDiag1[i1 - 1] + s);
      if (tmp < minScore) antiDiag3[i3] = undefined;
      else { antiDiag3[i3] = tmp; antiDiagBest = max(antiDiagBest, tmp); }
    }
    ncell += (unsigned long long)(maxCol - minCol);
    antiDiagBest = wave_max_i32(antiDiagBest);
    best = max(best, antiDiagBest);
    if (LDS) LSYNC(); else WSYNC();
    // new minCol / maxCol: uniform scans from the two ends of the band
    while (minCol - offset3 < len3 && antiDiag3[minCol - offset3] == undefined && minCol - offset2 - 1 < len2 &&
           antiDiag2[minCol - offset2 - 1] == undefined)
      ++minCol;
    while (maxCol - offset3 > 0 && antiDiag3[maxCol - offset3 - 1] == undefined &&
           antiDiag2[maxCol - offset2 - 1] == undefined)
      --maxCol;
    ++maxCol;
    minCol = max(minCol, antiDiagNo + 2 - rows);
    maxCol = min(maxCol, cols);
    if (LDS) LSYNC(); else WSYNC();  // the scans above must finish before lane 0 overwrites antiDiag1 (next antiDiag3)
  }
  cells += ncell;
  // longest extension
  int longestExtensionCol = len3 + offset3 - 2;
  int longestExtensionRow = antiDiagNo - longestExtensionCol;
  int longestExtensionScore = antiDiag3[longestExtensionCol - offset3];
  if (longestExtensionScore == undefined) {
    if (antiDiag2[len2 - 2] != undefined) {
      longestExtensionCol = len2 + offset2 - 2;
      longestExtensionRow = antiDiagNo - 1 - longestExtensionCol;
      longestExtensionScore = antiDiag2[longestExtensionCol - offset2];
    } else if (len2 > 2 && antiDiag2[len2 - 3] != undefined) {
      longestExtensionCol = len2 + offset2 - 3;
      longestExtensionRow = antiDiagNo - 1 - longestExtensionCol;
      longestExtensionScore = antiDiag2[longestExtensionCol - offset2];
    }
  }
  if (longestExtensionScore == undefined) {
    for (int i = 0; i < len1; ++i) {
      const int v = antiDiag1[i];
      if (v > longestExtensionScore) {
        longestExtensionScore = v;
        longestExtensionCol = i + offset1;
        longestExtensionRow = antiDiagNo - 2 - longestExtensionCol;
      }
    }
  }
  if (LDS) LSYNC(); else WSYNC();
  if (longestExtensionScore != undefined) { extCols = longestExtensionCol; extRows = longestExtensionRow; return true; }
  return false;
}

// ------------------------------------------------------------------ register-blocked NW (n <= 64*NB)
// Same recurrence and hand-off as wave_nw, but each lane keeps its block of the previous row and
// its H bases in registers (no LDS / memory traffic inside the sweep) and prefetches the V base
// of the next step.
template <int NB>
TALC_D int wave_nw_reg(const uint8_t* __restrict__ H_, int n, const uint8_t* __restrict__ V_, int m, int match, int mismatch,
                       int gap, bool freeBegin, unsigned long long& cells) {
  gcu8 H = (gcu8)uni_ptr(H_); gcu8 V = (gcu8)uni_ptr(V_);
  const int l = lane_id();
  n = uni(n); m = uni(m); match = uni(match); mismatch = uni(mismatch); gap = uni(gap);
  cells += (unsigned long long)n * (unsigned long long)m;
  if (n == 0) return freeBegin ? 0 : m * gap;
  if (m == 0) return freeBegin ? 0 : n * gap;
  const int B = (n + 63) >> 6;
  const int nl = (n + B - 1) / B;
  const int j0 = l * B + 1;
  const int nOwn = max(0, min(B, n - j0 + 1));
  int h[NB], r[NB];
#pragma unroll
  for (int jj = 0; jj < NB; ++jj) {
    const int j = j0 + jj;
    h[jj] = (jj < nOwn) ? (int)H[j - 1] : 255;
    r[jj] = freeBegin ? 0 : j * gap;
  }
  int lastOut = freeBegin ? 0 : (j0 + nOwn - 1) * gap;   // row 0 of the last own column
  int prevLastOut = lastOut;
  const int T = m + nl - 1;
  int vcur = (l == 0) ? (int)V[0] : 0;
  for (int t = 1; t <= T; ++t) {
    int vnext = 0;
    { const int idx = t - l; if (idx >= 0 && idx < m) vnext = (int)V[idx]; }
    const int nbLast = lane_shr1(lastOut);
    const int nbPrev = lane_shr1(prevLastOut);
    const int i = t - l;
    if (l < nl && i >= 1 && i <= m) {
      int left, diag;
      if (l == 0) { left = freeBegin ? 0 : i * gap; diag = freeBegin ? 0 : (i - 1) * gap; }
      else { left = nbLast; diag = nbPrev; }
      int v = left;
#pragma unroll
      for (int jj = 0; jj < NB; ++jj) {
        if (jj < nOwn) {
          const int up = r[jj];
          const int d = diag + ((h[jj] == vcur) ? match : mismatch);
          v = max(d, max(up + gap, left + gap));
          r[jj] = v;
          diag = up;
          left = v;
        }
      }
      prevLastOut = diag;
      lastOut = v;
    }
    vcur = vnext;
  }
  // D[m][n] sits in the lane that owns column n
  const int ln = (n - 1) / B, tj = (n - 1) - ln * B;
  int res = 0;
#pragma unroll
  for (int jj = 0; jj < NB; ++jj) if (jj == tj) res = r[jj];
  return lane_get(res, ln);
}

// ------------------------------------------------------------------ NW continued row by row (n <= 64*NB)
// scoreBridges (Explorer.cpp:689-706) aligns every live Trail against the reference again every CHECK_INTERVAL steps,
// from scratch (Trail::Overlapscore, Trail.cpp:145-173: free begin, end gaps charged).  The matrix cell (i, j) of such an
// alignment depends on the first i bases of the Trail and the first j of the reference only — neither the Trail's later
// bases nor the truncation of the reference change it — so a Trail that keeps the last row it has computed (over the
// WHOLE reference) only needs the rows of its new bases: rows i0+1 .. m of the free-begin matrix of V (rows) against H
// (columns), continued from row i0 in rowIO[1..n] (i0 == 0: row 0 is all zero and rowIO is not read); leaves row m
// there and returns D[m][outCol] = the score of V[0, m) against H[0, outCol).
// ROW BY ROW, every lane on the same row (the lane-skewed sweep of wave_nw_reg needs rows + lanes - 1 steps, and a scoring
// adds CHECK_INTERVAL rows: 64 steps for 6 rows).  The one dependency along a row, v[j] = max(a[j], v[j-1] + gap) with
// a[j] = max(diagonal move, vertical move), is a prefix maximum after the substitution v'[j] = v[j] - j gap
// (v'[j] = max(a'[j], v'[j-1]), v'[0] = 0: column 0 is free): each lane takes the running maximum of its own columns,
// one DPP scan gives every lane the maximum of the lanes before it, and a second pass over the own columns finishes the
// row.  Same integers as the sweep, cell for cell.  The Trail's bases of 64 rows sit in one register (lane k: row k's).
// (two halves: the lane's column bases depend on the reference alone and serve every Trail of a scoring)
template <int NB>
TALC_D void nw_rows_cols(const uint8_t* __restrict__ H_, int n, unsigned (&hp)[(NB + 3) / 4]) {
  gcu8 H = (gcu8)uni_ptr(H_);
  const int l = lane_id();
  n = uni(n);
  const int B = (n + 63) >> 6;
  const int j0 = l * B + 1;
  const int nOwn = max(0, min(B, n - j0 + 1));
#pragma unroll
  for (int q = 0; q < (NB + 3) / 4; ++q) hp[q] = 0xFFFFFFFFu;
#pragma unroll
  for (int jj = 0; jj < NB; ++jj)
    if (jj < nOwn) hp[jj >> 2] = (hp[jj >> 2] & ~(0xFFu << (8 * (jj & 3)))) | ((unsigned)H[j0 + jj - 1] << (8 * (jj & 3)));
}
template <int NB>
TALC_D int nw_rows_run(const unsigned (&hp)[(NB + 3) / 4], int n, const uint8_t* __restrict__ V_, int i0, int m, int match, int mismatch,
                       int gap, int* rowIO_, int outCol, unsigned long long& cells) {
  gcu8 V = (gcu8)uni_ptr(V_);
  int TALC_AS1* rowIO = (int TALC_AS1*)uni_ptr(rowIO_);   // [0] belongs to the caller (column 0 is zero by definition)
  const int l = lane_id();
  n = uni(n); m = uni(m); i0 = uni(i0); outCol = uni(outCol); match = uni(match); mismatch = uni(mismatch); gap = uni(gap);
  const int rows = m - i0;
  cells += (unsigned long long)n * (unsigned long long)max(rows, 0);
  const int B = (n + 63) >> 6;
  const int j0 = l * B + 1;
  const int nOwn = max(0, min(B, n - j0 + 1));
  int r[NB];
#pragma unroll
  for (int jj = 0; jj < NB; ++jj) r[jj] = (jj < nOwn && i0 > 0) ? rowIO[j0 + jj] : 0;
  const int jg0 = j0 * gap;   // j gap of the first own column
  int vreg = (l < rows) ? (int)V[i0 + l] : 0;
  for (int c0 = 0; c0 < rows; c0 += 64) {
    const int vnext = (c0 + 64 + l < rows) ? (int)V[i0 + c0 + 64 + l] : 0;   // (asked for 64 rows ahead)
    const int cr = min(64, rows - c0);
    for (int i = 0; i < cr; ++i) {
      const int vb = __builtin_amdgcn_readlane(vreg, i);
      int lastOwn = 0;   // the row above at the last own column
#pragma unroll
      for (int jj = 0; jj < NB; ++jj) if (jj == nOwn - 1) lastOwn = r[jj];
      int diag = lane_shr1(lastOwn);
      diag = (l == 0) ? 0 : diag;   // (column 0 is free: all zero)
      int pm = INT_MIN / 2;
      int ap[NB];
#pragma unroll
      for (int jj = 0; jj < NB; ++jj) {
        ap[jj] = INT_MIN / 2;
        if (jj < nOwn) {
          const int up = r[jj];
          const int hb = (int)((hp[jj >> 2] >> (8 * (jj & 3))) & 0xFFu);
          const int a = max(diag + ((hb == vb) ? match : mismatch), up + gap);
          diag = up;
          pm = max(pm, a - (jg0 + jj * gap));
          ap[jj] = pm;
        }
      }
      int sc = pm;   // (INT_MIN / 2 in a lane without columns)
      TALC_WAVE_REDUCE(sc, max, INT_MIN);   // inclusive prefix maximum over the lanes
      int e = lane_shr1(sc);
      e = (l == 0) ? 0 : max(e, 0);
#pragma unroll
      for (int jj = 0; jj < NB; ++jj) if (jj < nOwn) r[jj] = max(ap[jj], e) + (jg0 + jj * gap);
    }
    vreg = vnext;
  }
#pragma unroll
  for (int jj = 0; jj < NB; ++jj) if (jj < nOwn) rowIO[j0 + jj] = r[jj];
  const int ln = (outCol - 1) / B, tj = (outCol - 1) - ln * B;
  int res = 0;
#pragma unroll
  for (int jj = 0; jj < NB; ++jj) if (jj == tj) res = r[jj];
  return lane_get(res, ln);
}
template <int NB>
TALC_D int wave_nw_rows(const uint8_t* __restrict__ H_, int n, const uint8_t* __restrict__ V_, int i0, int m, int match, int mismatch,
                        int gap, int* rowIO_, int outCol, unsigned long long& cells) {
  unsigned hp[(NB + 3) / 4];   // the lane's column bases, four per register
  nw_rows_cols<NB>(H_, n, hp);
  return nw_rows_run<NB>(hp, n, V_, i0, m, match, mismatch, gap, rowIO_, outCol, cells);
}

// ------------------------------------------------------------------ fused edit distance + LCS (n <= 64*NB)
// The reference always scores a candidate twice against the same reference: globalAlignment with
// Score(0,-1,-1) (= -edit distance, Trajectory.cpp:413 / Trail.cpp:422) and localAlignment with
// Score(1,0,0) (= LCS length, Trajectory.cpp:368,525).  One lane-skewed sweep computes both
// (shared base loads, shared hand-off structure).
template <int NB>
TALC_D void wave_edit_lcs_reg(const uint8_t* __restrict__ H_, int n, const uint8_t* __restrict__ V_, int m, int& editScore,
                              int& lcsLen, unsigned long long& cells) {
  gcu8 H = (gcu8)uni_ptr(H_); gcu8 V = (gcu8)uni_ptr(V_);
  const int l = lane_id();
  n = uni(n); m = uni(m);
  cells += 2ull * (unsigned long long)n * (unsigned long long)m;
  if (n == 0 || m == 0) { editScore = -(n + m); lcsLen = 0; return; }
  const int B = (n + 63) >> 6;
  const int nl = (n + B - 1) / B;
  const int j0 = l * B + 1;
  const int nOwn = max(0, min(B, n - j0 + 1));
  int h[NB], re[NB], rl[NB];
#pragma unroll
  for (int jj = 0; jj < NB; ++jj) {
    const int j = j0 + jj;
    h[jj] = (jj < nOwn) ? (int)H[j - 1] : 255;
    re[jj] = -j;
    rl[jj] = 0;
  }
  int lastE = -(j0 + nOwn - 1), prevE = lastE, lastL = 0, prevL = 0;
  const int T = m + nl - 1;
  int vcur = (l == 0) ? (int)V[0] : 0;
  for (int t = 1; t <= T; ++t) {
    int vnext = 0;
    { const int idx = t - l; if (idx >= 0 && idx < m) vnext = (int)V[idx]; }
    const int nbLastE = lane_shr1(lastE), nbPrevE = lane_shr1(prevE);
    const int nbLastL = lane_shr1(lastL), nbPrevL = lane_shr1(prevL);
    const int i = t - l;
    if (l < nl && i >= 1 && i <= m) {
      int leftE, diagE, leftL, diagL;
      if (l == 0) { leftE = -i; diagE = -(i - 1); leftL = 0; diagL = 0; }
      else { leftE = nbLastE; diagE = nbPrevE; leftL = nbLastL; diagL = nbPrevL; }
      int vE = leftE, vL = leftL;
#pragma unroll
      for (int jj = 0; jj < NB; ++jj) {
        if (jj < nOwn) {
          const bool eq = (h[jj] == vcur);
          const int upE = re[jj], upL = rl[jj];
          vE = max(diagE + (eq ? 0 : -1), max(upE, leftE) - 1);
          vL = max(diagL + (eq ? 1 : 0), max(upL, leftL));
          re[jj] = vE; rl[jj] = vL;
          diagE = upE; leftE = vE;
          diagL = upL; leftL = vL;
        }
      }
      prevE = diagE; lastE = vE;
      prevL = diagL; lastL = vL;
    }
    vcur = vnext;
  }
  const int ln = (n - 1) / B, tj = (n - 1) - ln * B;
  int resE = 0, resL = 0;
#pragma unroll
  for (int jj = 0; jj < NB; ++jj) if (jj == tj) { resE = re[jj]; resL = rl[jj]; }
  editScore = lane_get(resE, ln);
  lcsLen = lane_get(resL, ln);
}

// ------------------------------------------------------------------ x-drop as a wavefront recurrence
// The unit-cost x-drop extension (match 0, mismatch -1, gap -1 — the only scoring the correction path
// extends seeds with) computed as furthest-reaching points, one diagonal per lane (NR diagonals per
// lane for wide x).  talc_wfa.h states the algorithm and why it returns exactly what the
// anti-diagonal formulation returns; this is the same code, lane-parallel:
//   F[s] of lane l = last kept anti-diagonal of diagonal k = kmin + 64 s + l, E[s] = level that reached it.
// One level costs ~20 VALU ops plus the match-run extension (8 bases per LDS read pair), against one
// full sweep of the band per anti-diagonal in the DP formulation.
// Returns 1 if the seed moves, 0 if not, -1 if x needs more than 64*NR-1 diagonals or the segments do
// not fit the LDS stage (the caller falls back to the anti-diagonal DP).
TALC_D int lane_rol1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x134, 0xF, 0xF, true); }
TALC_D int wave_min_i32(int v) {
  TALC_WAVE_REDUCE(v, min, INT_MAX);
  return __builtin_amdgcn_readlane(v, 63);
}
TALC_D unsigned long long lds_load_u64(const uint8_t TALC_AS3* p) {
  unsigned long long v;
  __builtin_memcpy(&v, (const void TALC_AS3*)p, 8);   // any alignment: the LDS runs in unaligned mode
  return v;
}

// Follow the match runs of the diagonals k = kmin + 64 s + lane from anti-diagonal a[s] (lanes with act[s] set), all
// slots together: a[s] ends on the run's last anti-diagonal.  The segments are staged with a sentinel behind each (the
// sentinels differ from each other and from every base, so every run ends) and 16 bytes of slack behind that.
// Written without branches inside a slot (the number of equal leading bytes of two 8-byte words is a handful of
// VALU ops): the first round compares eight bases — on a diagonal off the alignment a run is rarely longer than one or
// two bases — the later rounds sixteen.
// v_ffbl_b32 as the hardware has it: the position of the lowest set bit, 0xFFFFFFFF for zero (the compiler's count-
// trailing-zeros wraps it in a compare and a select per use to define the zero case; here 0xFFFFFFFF is what is wanted)
TALC_D unsigned ffbl_raw(unsigned x) {
  unsigned r;
  asm("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(x));
  return r;
}
// position of the first differing byte of two 8-byte words x 8, as the hardware's find-first-bit gives it: 0 .. 63 (bits
// below the first differing byte's: >> 3 = equal leading bytes), or 0xFFFFFFFF when the words agree
TALC_D unsigned wfa_first_diff_raw(unsigned long long x, unsigned long long y) {
  const unsigned long long w = x ^ y;
  // an all-equal half gives 0xFFFFFFFF (unchanged by the "| 32"), which loses the min
  return min(ffbl_raw((unsigned)w), ffbl_raw((unsigned)(w >> 32)) | 32u);
}
TALC_D unsigned wfa_equal_prefix8(unsigned long long x, unsigned long long y) {   // number of equal leading (low) bytes, 0..8
  return min(wfa_first_diff_raw(x, y) >> 3, 8u);
}
// (Whether a lane goes on is carried as the raw value t of its last comparison — "t > 63" is a plain compare whose ballot
//  is the compare's own lane mask; a boolean carried through the divergent rounds costs a select and a second compare per
//  test.  The database address is the query address plus a per-lane constant: a + k is even on every cell of diagonal k.)
template <int NR>
TALC_D void wfa_extend(const uint8_t TALC_AS3* stage, int qpad, int kmin, int (&a)[NR], bool (&act)[NR]) {
  const int l = lane_id();
  if constexpr (NR == 1) {
    // the one-diagonal-per-lane instance (two thirds of all levels): as above
    const int k = kmin + l;
    unsigned qa = (unsigned)((a[0] + k) >> 1), da = qa + (unsigned)(qpad - k), t = 0u;
    if (act[0]) {
      t = wfa_first_diff_raw(lds_load_u64(stage + qa), lds_load_u64(stage + da));
      a[0] += 2 * (int)min(t >> 3, 8u); qa += 8; da += 8;
    }
    while (ballot64(t > 63u) != 0ull) {
      const bool go = t > 63u;
      t = 0u;
      if (go) {
        const unsigned long long q0 = lds_load_u64(stage + qa), d0 = lds_load_u64(stage + da);
        const unsigned long long q1 = lds_load_u64(stage + qa + 8), d1 = lds_load_u64(stage + da + 8);
        const unsigned t0 = wfa_first_diff_raw(q0, d0), t1 = wfa_first_diff_raw(q1, d1);
        const unsigned n = (t0 > 63u) ? 8u + min(t1 >> 3, 8u) : (t0 >> 3);
        a[0] += 2 * (int)n; qa += 16; da += 16;
        t = min(t0, t1);   // both words agreed: on to the next sixteen bases
      }
    }
    act[0] = false;
  } else {
    // the wider instances keep the form whose registers they can afford (the raw values of every slot carried through the
    // rounds made the two- and four-wide level loops spill: +5 GB of scratch traffic per config-2 launch)
    unsigned qa[NR], da[NR];
    bool any = false;
#pragma unroll
    for (int s = 0; s < NR; ++s) {
      const int k = kmin + 64 * s + l;
      qa[s] = (unsigned)((a[s] + k) >> 1); da[s] = (unsigned)(qpad + ((a[s] - k) >> 1));
      if (act[s]) {
        const unsigned n = wfa_equal_prefix8(lds_load_u64(stage + qa[s]), lds_load_u64(stage + da[s]));
        a[s] += 2 * (int)n; qa[s] += 8; da[s] += 8;
        act[s] = (n == 8u);
      }
      any |= act[s];
    }
    while (ballot64(any) != 0ull) {
      any = false;
#pragma unroll
      for (int s = 0; s < NR; ++s) {
        if (act[s]) {
          const unsigned long long q0 = lds_load_u64(stage + qa[s]), d0 = lds_load_u64(stage + da[s]);
          const unsigned long long q1 = lds_load_u64(stage + qa[s] + 8), d1 = lds_load_u64(stage + da[s] + 8);
          const unsigned n0 = wfa_equal_prefix8(q0, d0), n1 = wfa_equal_prefix8(q1, d1);
          const unsigned n = (n0 == 8u) ? 8u + n1 : n0;
          a[s] += 2 * (int)n; qa[s] += 16; da[s] += 16;
          act[s] = (n == 16u);
          any |= act[s];
        }
      }
    }
  }
}

// ---- where the anti-diagonal loop of the original stops, and the cell it reports (talc_wfa.h), given the furthest
// anti-diagonal F and its level E of every diagonal k = kmin + 64 s + lane.  Returns 0 when there is nothing to report.
template <int NR>
TALC_D int wfa_select(const int (&F)[NR], const int (&E)[NR], int kmin, int kmax, int qlen, int dlen,
                      int& extCols, int& extRows, int& extScore) {
  const int l = lane_id();
  const int NEG = -(1 << 29);
  int mF = NEG;
#pragma unroll
  for (int s = 0; s < NR; ++s) mF = max(mF, F[s]);
  const int A = wave_max_i32(mF);
  const int cols = qlen + 1, rows = dlen + 1;
  auto minmaxS = [&](int a, bool dropTopBorder, int& mn, int& mx) {
    int lmn = INT_MAX, lmx = INT_MIN;
#pragma unroll
    for (int s = 0; s < NR; ++s) {
      const int k = kmin + 64 * s + l;
      const int c = (a + k) >> 1;
      const bool on = (((k - a) & 1) == 0) & (F[s] >= a) & !(dropTopBorder & (c == a));
      if (on) { lmn = min(lmn, c); lmx = max(lmx, c); }
    }
    mn = wave_min_i32(lmn); mx = wave_max_i32(lmx);
  };
  // value of F / E on the wave-uniform diagonal k (must lie in [kmin, kmax])
  auto at = [&](const int (&r)[NR], int k) -> int {
    const int j = k - kmin, slot = j >> 6;
    int v = r[0];
#pragma unroll
    for (int s = 0; s < NR; ++s) if (s == slot) v = r[s];
    return lane_get(v, j & 63);
  };
  auto kept = [&](int a, int c) -> bool {
    const int k = 2 * c - a;
    if (c < 0 || a - c < 0 || k < kmin || k > kmax) return false;
    return at(F, k) >= a;
  };
  auto take = [&](int a, int c) { extCols = c; extRows = a - c; extScore = -at(E, 2 * c - a); };
  auto firstMax = [&](int a) -> int {   // least level, then least column, among the diagonals ending on a
    int key = INT_MAX;
#pragma unroll
    for (int s = 0; s < NR; ++s) {
      const int j = 64 * s + l, k = kmin + j;
      if ((((k - a) & 1) == 0) & (F[s] == a)) key = min(key, (E[s] << 10) | j);
    }
    const int kk = wave_min_i32(key);
    if (kk == INT_MAX) return 0;
    const int k = kmin + (kk & 1023);
    extCols = (a + k) >> 1; extRows = (a - k) >> 1; extScore = -(kk >> 10);
    return 1;
  };
  bool early = false;   // the loop stops one anti-diagonal after A: every successor is outside the matrix
  if (A + 1 >= 2) {
    int mn, mx;
    minmaxS(A, false, mn, mx);
    const int lo = max(1 + mn, A + 3 - rows), hi = min(2 + mx, cols);
    early = lo >= hi;
  }
  if (!early) return firstMax(A);
  int maxColA = 1;
  if (A >= 2) {
    int mn1, mx1, mn2, mx2;
    minmaxS(A - 1, true, mn1, mx1);
    minmaxS(A - 2, false, mn2, mx2);
    const int cm = max(mx1, mx2);
    maxColA = (cm == INT_MIN) ? cols : min(2 + cm, cols);
  }
  const int c2 = maxColA - 1;
  if (kept(A, c2)) { take(A, c2); return 1; }
  if (A >= 2 && kept(A, c2 - 1)) { take(A, c2 - 1); return 1; }
  return firstMax(A - 1);
}

// dst[0..n) (LDS, 8-byte aligned) = src[0..n) (global, any alignment): 8 bytes per lane per pass — whole words only, so
// nothing beyond src[n) is read; the last n % 8 bytes go one by one
TALC_D void stage_copy(uint8_t TALC_AS3* dst, gcu8 src, int n) {
  const int l = lane_id_here();
  typedef uint64_t __attribute__((aligned(1))) u64u;
  const int nw = n >> 3;
  for (int w = l; w < nw; w += 64) ((uint64_t TALC_AS3*)dst)[w] = *(const u64u TALC_AS1*)(src + 8 * w);
  const int t = (nw << 3) + l;
  if (t < n) dst[t] = src[t];
}

// Phases.  Level e only involves the diagonals |k| <= e, so a run whose x needs 2 or 4 diagonals per lane can take its
// first 31 levels with one diagonal per lane, the next 32 with two, and only the rest at full width: a phase stops
// after level `toLevel`, leaves the state in memory by diagonal (index k + 256), and the next, wider instance
// continues from there (`fromLevel`).  The segments are staged once, by the first phase, for the run's real x.
// Return value 2 = "to be continued"; a phase that reaches the far corner ends the whole run (return 1).
struct WfaPhase {
  int fromLevel;        // -1: first phase (stages the segments, takes level 0)
  int toLevel;          // -1: last phase (runs to level x and selects)
  int* memF; int* memE;
};

#ifdef TALC_PROF
__shared__ uint32_t g_wprof[4];   // staging, levels, selection of wave_xdrop_wfa (category profile build)
#define WPROF_T() __builtin_amdgcn_s_memtime()
#define WPROF_ADD(i, t0) (g_wprof[i] += (uint32_t)(__builtin_amdgcn_s_memtime() - (t0)))
#else
#define WPROF_T() 0ull
#define WPROF_ADD(i, t0) ((void)(t0))
#endif

// A kept wavefront.  An edge search scores its one live Trail every CHECK_INTERVAL steps: the same database segment
// (the read's edge), the query (the Trail) a few bases longer, x a little larger each time.  Level e of the recurrence
// depends on the two segments only up to the furthest points it reaches, on x only through the diagonals |k| = x, and
// on the query's length only once a diagonal has followed the query to its end: the last level of a run BEFORE any
// diagonal touched the query's end (and below the run's x) is level e of every later run of the same pair with a longer
// query and a larger x.  The first phase (one diagonal per lane) keeps that level here, by diagonal, and the next run
// of the same pair — `keepKey` names the pair; whoever changes a Trail buffer's contents resets `owner` — starts from
// it.  A third of an extension's levels on average (the true cost of the Trail against the edge, ~ an eighth of its
// length, against an x of a third of it), all of them from the narrow phase.
// (F as 16 bits — an anti-diagonal of segments the LDS stage holds — with 0xFFFF for "not reached", E as 8: LDS is handed
//  out to the one-wave workgroups of k_search in steps of 1280 bytes and the kernel sits just below one)
struct WfaKeep { uint32_t owner; int level; int qlenAt; int pad_; uint16_t F[64]; uint8_t E[64]; };
__shared__ WfaKeep g_keep;

template <int NR>
TALC_D int wave_xdrop_wfa(const uint8_t* __restrict__ querySeg_, int qlen, const uint8_t* __restrict__ dbSeg_, int dlen, int x,
                          uint8_t TALC_AS3* stage, int stageCap, int& extCols, int& extRows, int& extScore,
                          unsigned long long& cells, const WfaPhase* ph = nullptr, uint32_t keepKey_ = 0) {
  gcu8 querySeg = (gcu8)uni_ptr(querySeg_); gcu8 dbSeg = (gcu8)uni_ptr(dbSeg_);
  const int l = lane_id();
  qlen = uni(qlen); dlen = uni(dlen); x = uni(x);
  extCols = extRows = extScore = 0;
  if (qlen <= 0 || dlen <= 0) return 0;
  const int NEG = -(1 << 29);
  const int X = min(max(x, 0), 1 << 20);
  const int fromLevel = ph ? ph->fromLevel : -1, toLevel = ph ? ph->toLevel : -1;
  const int Xm = (toLevel >= 0) ? min(X, toLevel) : X;   // the diagonals this phase can meet
  const int kmin = -min(Xm, dlen), kmax = min(Xm, qlen);
  const int nd = kmax - kmin + 1;
  if (nd > 64 * NR - 1) return -1;   // one always-empty lane closes the ring of the lane rotations
  // only cells with |col - row| <= x can be kept: stage that much of each segment, then a sentinel
  const int qS = min(qlen, dlen + X), dS = min(dlen, qlen + X);
  const int qpad = (qS + 16) & ~7;
  if (qpad + dS + 16 > stageCap) return -1;
  const unsigned long long _t0 = WPROF_T();
  if (fromLevel < 0) {
    stage_copy(stage, querySeg, qS);
    stage_copy(stage + qpad, dbSeg, dS);
    if (l == 0) { stage[qS] = 0xF0; stage[qpad + dS] = 0xF1; }   // differ from each other and from every base code
  }
  WSYNC();
  WPROF_ADD(0, _t0);
  const unsigned long long _t1 = WPROF_T();
  const int bmax = x >= 2 ? x - 1 : (x == 1 ? 1 : 0);
  const int corner = qlen + dlen;
  int F[NR], E[NR], amax[NR], forb[NR];
#pragma unroll
  for (int s = 0; s < NR; ++s) {
    const int j = 64 * s + l, k = kmin + j, ak = k < 0 ? -k : k;
    amax[s] = (j < nd) ? min(2 * qlen - k, 2 * dlen + k) : NEG;
    forb[s] = (ak == bmax + 1) ? ak : -1;   // the first border cell the x-drop leaves uninitialised
    F[s] = NEG; E[s] = 0;
  }
  auto extend = [&](int (&a)[NR], bool (&act)[NR]) { wfa_extend<NR>(stage, qpad, kmin, a, act); };
  bool cornerHit = false;
  int cornerE = 0;
  int eStart = 1;
  // keep / resume (NR == 1, first phase only)
  const uint32_t keepKey = (NR == 1 && fromLevel < 0) ? (uint32_t)uni((int)keepKey_) : 0u;
  bool tracking = keepKey != 0u;
  const int kOfLane = kmin + l;
  const int aq = 2 * qlen - kOfLane;           // the anti-diagonal on which this lane's diagonal meets the query's end
  bool resumed = false;
  if (NR == 1 && keepKey != 0u) {
    const int lv = uni(g_keep.level), qat = uni(g_keep.qlenAt);
    if ((uint32_t)uni((int)g_keep.owner) == keepKey && lv >= 1 && lv < x && lv <= 31 && lv < qat && qat <= qlen && lv < dlen) {
      const int ak = kOfLane < 0 ? -kOfLane : kOfLane;
      const bool in = (l < nd) & (ak <= lv);
      const uint32_t f16 = g_keep.F[(kOfLane + 32) & 63];
      F[0] = (in && f16 != 0xFFFFu) ? (int)f16 : NEG;
      E[0] = in ? (int)g_keep.E[(kOfLane + 32) & 63] : 0;
      eStart = lv + 1;
      resumed = true;
    }
  }
  auto keep_level = [&](int level) {   // the current F / E are level `level`'s
    if (level >= 1 && level <= 31) {
      if (l < nd) { g_keep.F[(kOfLane + 32) & 63] = (uint16_t)(F[0] >= 0 ? F[0] : 0xFFFF); g_keep.E[(kOfLane + 32) & 63] = (uint8_t)E[0]; }
      if (l == 0) { g_keep.owner = keepKey; g_keep.level = level; g_keep.qlenAt = qlen; }
    }
  };
  if (resumed) {
  } else
  if (fromLevel >= 0) {   // the state the narrower phase left behind
#pragma unroll
    for (int s = 0; s < NR; ++s) {
      const int j = 64 * s + l, k = kmin + j, ak = k < 0 ? -k : k;
      const bool in = (j < nd) & (ak <= fromLevel);
      F[s] = in ? ph->memF[in ? k + 256 : 256] : NEG;
      E[s] = in ? ph->memE[in ? k + 256 : 256] : 0;
    }
    eStart = fromLevel + 1;
  } else {
    int a0[NR]; bool act0[NR];
    const int j0 = -kmin;
#pragma unroll
    for (int s = 0; s < NR; ++s) { a0[s] = 0; act0[s] = (64 * s + l == j0) && x >= 0; }
    extend(a0, act0);
#pragma unroll
    for (int s = 0; s < NR; ++s) if (64 * s + l == j0) F[s] = a0[s];
    unsigned long long hit = 0;
#pragma unroll
    for (int s = 0; s < NR; ++s) hit |= ballot64(F[s] == corner);
    cornerHit = hit != 0ull;
    if (tracking && ballot64(F[0] == aq) != 0ull) tracking = false;   // level 0 runs to the query's end: nothing to keep
  }
  const int eEnd = (toLevel >= 0) ? min(x, toLevel) : x;
  unsigned long long work = 0;
  // the far corner lies on diagonal qlen - dlen: outside the band (an edge's reference is far longer than its Trail) it can
  // never be reached, and the level loop need not look for it
  const int kc = qlen - dlen;
  const bool cornerInBand = (kc >= kmin) & (kc <= kmax);
  int eLast = eStart - 1;
  for (int e = eStart; e <= eEnd && !cornerHit; ++e) {
    if (NR == 1 && tracking && e == x) { keep_level(e - 1); tracking = false; }   // level x is the one x itself shapes (forb)
    const bool forbLevel = (e == bmax + 1);
    int rotR[NR], rotL[NR];
#pragma unroll
    for (int s = 0; s < NR; ++s) { rotR[s] = lane_ror1(F[s]); rotL[s] = lane_rol1(F[s]); }
    // slots whose diagonals lie within [-e, e]
    const int sLo = max(0, -e - kmin) >> 6, sHi = min(nd - 1, e - kmin) >> 6;
    int b[NR]; bool act[NR];
#pragma unroll
    for (int s = 0; s < NR; ++s) {
      b[s] = NEG; act[s] = false;
      if (NR == 1 || (s >= sLo && s <= sHi)) {
        const int fl = (NR > 1 && l == 0) ? rotR[(s + NR - 1) % NR] : rotR[s];   // diagonal k-1
        const int fr = (NR > 1 && l == 63) ? rotL[(s + 1) % NR] : rotL[s];       // diagonal k+1
        int v2 = fl + 1; v2 = (v2 <= amax[s]) ? v2 : NEG;       // gap along the query
        int v3 = fr + 1; v3 = (v3 <= amax[s]) ? v3 : NEG;       // gap along the database
        int v = max(max(v2, v3), min(F[s] + 2, amax[s]));       // mismatch (clamps to F itself at the matrix end)
        // (the border cell the x-drop leaves uninitialised is cell |k| of diagonal |k| = bmax + 1: level e reaches diagonals
        //  |k| <= e only, and a diagonal enters at its cell |k|, so the test can only ever hit at level bmax + 1)
        if (forbLevel) v = (v == forb[s]) ? NEG : v;
        b[s] = v;
        act[s] = (v > F[s]) & (v >= 0);
      }
    }
    bool moved[NR];
#pragma unroll
    for (int s = 0; s < NR; ++s) moved[s] = act[s];
    extend(b, act);
    // (a lane whose candidate did not move it has b <= F < aq while no lane has met the query's end yet: the plain compare says it)
    if (NR == 1 && tracking && ballot64(b[0] == aq) != 0ull) {   // this level meets the query's end: keep the one before
      keep_level(e - 1);
      tracking = false;
    }
    unsigned long long hit = 0;
#pragma unroll
    for (int s = 0; s < NR; ++s) {
      if (moved[s]) { F[s] = b[s]; E[s] = e; }
      if (cornerInBand) hit |= ballot64(F[s] == corner);
    }
    if (hit != 0ull) { cornerHit = true; cornerE = e; }
    eLast = e;
  }
  if (NR == 1 && tracking && !cornerHit && eLast < x) keep_level(eLast);   // (a phase that ends below x untouched)
  if (eLast >= eStart) {   // the levels' diagonals, counted once: sum over e = eStart..eLast of min(nd, 2 e + 1)
    const int eh = min(eLast, (nd - 1) / 2);   // levels whose 2 e + 1 diagonals all fit the band
    const int lo = max(eh, eStart - 1);
    if (eh >= eStart) work += (unsigned long long)((eh - eStart + 1) * (eh + eStart + 1));
    if (eLast > lo) work += (unsigned long long)((eLast - lo) * nd);
  }
  cells += work;
  WPROF_ADD(1, _t1);
#ifdef TALC_PROF
  if (eLast >= eStart) g_wprof[3] += (uint32_t)(eLast - eStart + 1);
#endif
  const unsigned long long _t2 = WPROF_T();
  if (cornerHit) { extCols = qlen; extRows = dlen; extScore = -cornerE; return 1; }
  if (toLevel >= 0 && toLevel < x) {   // hand over to the wider instance
#pragma unroll
    for (int s = 0; s < NR; ++s) {
      const int j = 64 * s + l, k = kmin + j, ak = k < 0 ? -k : k;
      if ((j < nd) & (ak <= toLevel)) { ph->memF[k + 256] = F[s]; ph->memE[k + 256] = E[s]; }
    }
    WSYNC();
    return 2;
  }
  const int rsel = wfa_select<NR>(F, E, kmin, kmax, qlen, dlen, extCols, extRows, extScore);
  WPROF_ADD(2, _t2);
  return rsel;
}

// ---- the same extension for EVERY drop-off x in [0, xHi] from one run (findStopPosition, Trajectory.cpp:482-503,
// asks for x, x-1, x-2, ... until the extension shrinks).  Kept cells are {cost <= x} whatever x is, so the
// wavefronts of level e are those of any run with x >= e; the one thing that depends on x is the border cell the
// x-drop leaves uninitialised (|k| = x at anti-diagonal x, for x >= 2): a run with x = e never starts diagonal
// |k| = e from that cell.  So level e is taken once without that rule (the state larger x continue from), and the
// result for x = e is selected from a copy in which the two diagonals |k| = e are dropped if that cell was their
// only way in.  resCols[x] / resRows[x] / resScore[x] (global memory, xHi + 1 entries) receive what wave_xdrop_wfa(x) reports.
// Returns -1 when the band does not fit (the caller then asks x by x).
template <int NR>
TALC_D int wave_xdrop_wfa_multi(const uint8_t* __restrict__ querySeg_, int qlen, const uint8_t* __restrict__ dbSeg_, int dlen, int xHi,
                                uint8_t TALC_AS3* stage, int stageCap, int* resCols, int* resRows, int* resScore,
                                unsigned long long& cells) {
  gcu8 querySeg = (gcu8)uni_ptr(querySeg_); gcu8 dbSeg = (gcu8)uni_ptr(dbSeg_);
  const int l = lane_id();
  qlen = uni(qlen); dlen = uni(dlen); xHi = uni(xHi);
  if (qlen <= 0 || dlen <= 0 || xHi < 0) return -1;
  const int NEG = -(1 << 29);
  const int X = min(xHi, 1 << 20);
  const int kmin = -min(X, dlen), kmax = min(X, qlen);
  const int nd = kmax - kmin + 1;
  if (nd > 64 * NR - 1) return -1;
  const int qS = min(qlen, dlen + X), dS = min(dlen, qlen + X);
  const int qpad = (qS + 16) & ~7;
  if (qpad + dS + 16 > stageCap) return -1;
  stage_copy(stage, querySeg, qS);
  stage_copy(stage + qpad, dbSeg, dS);
  if (l == 0) { stage[qS] = 0xF0; stage[qpad + dS] = 0xF1; }
  WSYNC();
  const int corner = qlen + dlen;
  int F[NR], E[NR], amax[NR], ak[NR];
#pragma unroll
  for (int s = 0; s < NR; ++s) {
    const int j = 64 * s + l, k = kmin + j;
    ak[s] = k < 0 ? -k : k;
    amax[s] = (j < nd) ? min(2 * qlen - k, 2 * dlen + k) : NEG;
    F[s] = NEG; E[s] = 0;
  }
  auto extend = [&](int (&a)[NR], bool (&act)[NR]) { wfa_extend<NR>(stage, qpad, kmin, a, act); };
  auto hits = [&](const int (&f)[NR]) -> bool {
    unsigned long long hit = 0;
#pragma unroll
    for (int s = 0; s < NR; ++s) hit |= ballot64(f[s] == corner);
    return hit != 0ull;
  };
  // resScore[x]: the score wave_xdrop_wfa(x) reports (minus the cost of the reported cell), or 1 when it reports none
  auto emit_corner_from = [&](int xFrom, int level) {   // every x >= xFrom reaches the far corner, at this level
    for (int x = xFrom + l; x <= xHi; x += 64) { resCols[x] = qlen; resRows[x] = dlen; resScore[x] = -level; }
  };
  auto emit_selected = [&](int x, const int (&f)[NR], const int (&lev)[NR]) {
    int c = 0, r = 0, sc = 0;
    if (!wfa_select<NR>(f, lev, kmin, kmax, qlen, dlen, c, r, sc)) { c = 0; r = 0; sc = 1; }
    if (l == 0) { resCols[x] = c; resRows[x] = r; resScore[x] = sc; }
  };
  {
    int a0[NR]; bool act0[NR];
    const int j0 = -kmin;
#pragma unroll
    for (int s = 0; s < NR; ++s) { a0[s] = 0; act0[s] = (64 * s + l == j0); }
    extend(a0, act0);
#pragma unroll
    for (int s = 0; s < NR; ++s) if (64 * s + l == j0) F[s] = a0[s];
  }
  unsigned long long work = 0;
  if (hits(F)) { emit_corner_from(0, 0); return 1; }
  emit_selected(0, F, E);
  for (int e = 1; e <= xHi; ++e) {
    int rotR[NR], rotL[NR];
#pragma unroll
    for (int s = 0; s < NR; ++s) { rotR[s] = lane_ror1(F[s]); rotL[s] = lane_rol1(F[s]); }
    const int sLo = max(0, -e - kmin) >> 6, sHi = min(nd - 1, e - kmin) >> 6;
    int b[NR]; bool act[NR], borderOnly[NR];
#pragma unroll
    for (int s = 0; s < NR; ++s) {
      b[s] = NEG; act[s] = false; borderOnly[s] = false;
      if (NR == 1 || (s >= sLo && s <= sHi)) {
        const int fl = (NR > 1 && l == 0) ? rotR[(s + NR - 1) % NR] : rotR[s];   // diagonal k-1
        const int fr = (NR > 1 && l == 63) ? rotL[(s + 1) % NR] : rotL[s];       // diagonal k+1
        int v2 = fl + 1; v2 = (v2 <= amax[s]) ? v2 : NEG;
        int v3 = fr + 1; v3 = (v3 <= amax[s]) ? v3 : NEG;
        const int v = max(max(v2, v3), min(F[s] + 2, amax[s]));
        borderOnly[s] = (e >= 2) & (ak[s] == e) & (v == e);   // what a run with x = e turns into "no cell"
        b[s] = v;
        act[s] = (v > F[s]) & (v >= 0);
      }
    }
    bool moved[NR];
#pragma unroll
    for (int s = 0; s < NR; ++s) moved[s] = act[s];
    extend(b, act);
    int Fx[NR], Ex[NR];
#pragma unroll
    for (int s = 0; s < NR; ++s) {
      const int fOld = F[s], eOld = E[s];
      if (moved[s]) { F[s] = b[s]; E[s] = e; }
      Fx[s] = borderOnly[s] ? fOld : F[s];
      Ex[s] = borderOnly[s] ? eOld : E[s];
    }
    work += (unsigned long long)min(nd, 2 * e + 1);
    if (hits(Fx)) { if (l == 0) { resCols[e] = qlen; resRows[e] = dlen; resScore[e] = -e; } }
    else emit_selected(e, Fx, Ex);
    if (hits(F)) { emit_corner_from(e + 1, e); break; }
  }
  cells += work;
  return 1;
}

// ------------------------------------------------------------------ global distances as wavefronts
// Distance between two whole sequences staged in LDS (query at stage[0..qlen), database at
// stage[qpad..qpad+dlen), each followed by its sentinel): with SUBST the edit distance (= minus the
// global alignment score for match 0, mismatch -1, gap -1: Trail.cpp:408-434), without it the indel
// distance qlen + dlen - 2 LCS (LCS = the localAlignment(1,0,0) optimum of Trajectory.cpp:368,525).
// Same lane layout and level step as wave_xdrop_wfa, no drop-off: the levels run until the far corner
// is reached.  Returns -1 if that takes more than 32*NR-1 levels (the diagonals no longer fit the lanes).
// acceptA (without SUBST only): a path that has reached anti-diagonal a with e insertions/deletions has matched
// (a - e) / 2 bases, a lower bound of the LCS; as soon as some diagonal has a - e >= acceptA the routine returns -2
// ("the LCS is at least acceptA / 2") instead of running on to the corner.  INT_MAX = never.
template <int NR, bool SUBST>
TALC_D int wave_wfa_global(const uint8_t TALC_AS3* stage, int qpad, int qlen, int dlen, unsigned long long& cells, int acceptA = INT_MAX) {
  const int l = lane_id();
  const int NEG = -(1 << 29);
  const int E = 32 * NR - 1;
  const int kmin = -min(E, dlen), kmax = min(E, qlen);
  const int nd = kmax - kmin + 1;
  const int kc = qlen - dlen, corner = qlen + dlen;
  if (kc < kmin || kc > kmax) return -1;
  int F[NR], amax[NR];
#pragma unroll
  for (int s = 0; s < NR; ++s) {
    const int j = 64 * s + l, k = kmin + j;
    amax[s] = (j < nd) ? min(2 * qlen - k, 2 * dlen + k) : NEG;
    F[s] = NEG;
  }
  auto extend = [&](int (&a)[NR], bool (&act)[NR]) { wfa_extend<NR>(stage, qpad, kmin, a, act); };
  {
    int a0[NR]; bool act0[NR];
    const int j0 = -kmin;
#pragma unroll
    for (int s = 0; s < NR; ++s) { a0[s] = 0; act0[s] = (64 * s + l == j0); }
    extend(a0, act0);
#pragma unroll
    for (int s = 0; s < NR; ++s) if (64 * s + l == j0) F[s] = a0[s];
  }
  {
    unsigned long long hit = 0, acc = 0;
#pragma unroll
    for (int s = 0; s < NR; ++s) { hit |= ballot64(F[s] == corner); if (!SUBST) acc |= ballot64(F[s] >= acceptA); }
    if (hit != 0ull) return 0;
    if (!SUBST && acc != 0ull) return -2;
  }
  unsigned long long work = 0;
  int result = -1;
  for (int e = 1; e <= E; ++e) {
    int rotR[NR], rotL[NR];
#pragma unroll
    for (int s = 0; s < NR; ++s) { rotR[s] = lane_ror1(F[s]); rotL[s] = lane_rol1(F[s]); }
    const int sLo = max(0, -e - kmin) >> 6, sHi = min(nd - 1, e - kmin) >> 6;
    int b[NR]; bool act[NR];
#pragma unroll
    for (int s = 0; s < NR; ++s) {
      b[s] = NEG; act[s] = false;
      if (NR == 1 || (s >= sLo && s <= sHi)) {
        const int fl = (NR > 1 && l == 0) ? rotR[(s + NR - 1) % NR] : rotR[s];
        const int fr = (NR > 1 && l == 63) ? rotL[(s + 1) % NR] : rotL[s];
        int v2 = fl + 1; v2 = (v2 <= amax[s]) ? v2 : NEG;
        int v3 = fr + 1; v3 = (v3 <= amax[s]) ? v3 : NEG;
        int v = max(v2, v3);
        if (SUBST) v = max(v, min(F[s] + 2, amax[s]));
        b[s] = v;
        act[s] = (v > F[s]) & (v >= 0);
      }
    }
    bool moved[NR];
#pragma unroll
    for (int s = 0; s < NR; ++s) moved[s] = act[s];
    extend(b, act);
    unsigned long long hit = 0, acc = 0;
#pragma unroll
    for (int s = 0; s < NR; ++s) {
      if (moved[s]) F[s] = b[s];
      hit |= ballot64(F[s] == corner);
      if (!SUBST) acc |= ballot64(F[s] - e >= acceptA);
    }
    work += (unsigned long long)min(nd, 2 * e + 1);
    if (hit != 0ull) { result = e; break; }
    if (!SUBST && acc != 0ull) { result = -2; break; }
  }
  cells += work;
  return result;
}

TALC_D unsigned long long load_u64_unaligned_fwd(gcu8 p) {
  typedef unsigned long long __attribute__((aligned(1))) u64u;
  return *(const u64u TALC_AS1*)p;
}

// ------------------------------------------------------------------ bit-parallel LCS (long sequences)
// LCS length of two sequences by the bit-vector recurrence of Crochemore et al. / Hyyro (2004): with the columns of
// the DP matrix as the bits of V (1 = the row's LCS value does not grow at that column),
//     V' = (V + (V & M[b])) | (V & ~M[b])        M[b] = columns whose base equals the row's base b
// row by row, and LCS = number of 0 bits in the final V.  The columns span the wave: lane l holds bits [64 l, 64 l + 63],
// so up to 4096 columns; the one cross-lane dependency is the carry of the addition, resolved per row with two ballots
// and scalar arithmetic: a lane generates a carry (g) or would pass one on (p, its sum is all ones) — never both — and
// the carries into the lanes are ((G << 1) + P) ^ P | (G << 1)  (verified exhaustively; tests/test_pure_vs_oracle.py).
// About 40 instructions per ROW whatever the number of columns, against cells / 64 for the lane-skewed DP and
// levels x diagonals for the wavefront forms: this is the routine for sequences beyond a few hundred bases (K = 31
// gaps, config 5).  Equal codes match (N with N, like Score<int,Simple>).
// this lane's match masks (one per code) of the 64 columns [c0, c0 + 64) of cs[0, m)
TALC_D void bitpar_masks(gcu8 cs, int m, int c0, unsigned long long& pm0, unsigned long long& pm1, unsigned long long& pm2,
                         unsigned long long& pm3, unsigned long long& pm4) {
  pm0 = pm1 = pm2 = pm3 = pm4 = 0;
  for (int w = 0; w < 8; ++w) {
    const int base = c0 + 8 * w;
    if (base >= m) break;
    unsigned long long bytes = 0;
    if (base + 8 <= m) bytes = load_u64_unaligned_fwd(cs + base);
    else for (int i = 0; base + i < m; ++i) bytes |= (unsigned long long)cs[base + i] << (8 * i);
    const int nb = min(8, m - base);
    for (int i = 0; i < nb; ++i) {
      const unsigned c = (unsigned)(bytes >> (8 * i)) & 0xFFu;
      const unsigned long long bit = 1ull << (8 * w + i);
      pm0 |= (c == 0u) ? bit : 0ull; pm1 |= (c == 1u) ? bit : 0ull; pm2 |= (c == 2u) ? bit : 0ull;
      pm3 |= (c == 3u) ? bit : 0ull; pm4 |= (c == 4u) ? bit : 0ull;
    }
  }
}

// Sequences beyond 4096 columns (the gaps of 20 kb reads are searched with Trails of up to 24 k bases): the columns are
// taken in blocks of 4096, block after block over all rows; what a block hands to the next is one carry bit per row,
// kept in `work` ((rows + 63) / 64 words of global memory; nullptr: one block only, -1 beyond).
TALC_D int wave_lcs_bitpar(const uint8_t* __restrict__ a_, int la, const uint8_t* __restrict__ b_, int lb, unsigned long long& cells,
                           unsigned long long* work_ = nullptr) {
  la = uni(la); lb = uni(lb);
  if (la == 0 || lb == 0) return 0;
  unsigned long long* const work = (unsigned long long*)uni_ptr(work_);
  // the cost is per row and block of columns: columns = whichever sequence makes blocks x rows smaller (the longer on a tie)
  const uint8_t* colp = uni_ptr(a_); int m = la; const uint8_t* rowp = uni_ptr(b_); int n = lb;
  {
    const long long costA = (long long)((la + 4095) >> 12) * lb, costB = (long long)((lb + 4095) >> 12) * la;
    if (costB < costA || (costB == costA && lb > la)) { colp = uni_ptr(b_); m = lb; rowp = uni_ptr(a_); n = la; }
  }
  const int nblk = (m + 4095) >> 12;
  if (nblk > 1 && !work) return -1;
  cells += (unsigned long long)la * (unsigned long long)lb;
  gcu8 cs = (gcu8)colp; gcu8 rs = (gcu8)rowp;
  const int l = lane_id_here();
  const unsigned long long laneBit = 1ull << l;
  int z = 0;
  for (int blk = 0; blk < nblk; ++blk) {
    const int c00 = blk << 12;
    unsigned long long pm0, pm1, pm2, pm3, pm4;
    bitpar_masks(cs, m, c00 + 64 * l, pm0, pm1, pm2, pm3, pm4);
    unsigned long long V = ~0ull;
    for (int j0 = 0; j0 < n; j0 += 64) {
      const int jn = min(64, n - j0);
      const int bvec = (l < jn) ? (int)rs[j0 + l] : 0;
      const unsigned long long cinW = (blk > 0) ? uni64(work[j0 >> 6]) : 0ull;   // the carries out of the block before, row by row
      unsigned long long coutW = 0;
      for (int jj = 0; jj < jn; ++jj) {
        const int b = lane_get(bvec, jj);     // the row's base, wave-uniform
        const unsigned long long M = (b == 0) ? pm0 : (b == 1) ? pm1 : (b == 2) ? pm2 : (b == 3) ? pm3 : pm4;
        const unsigned long long U = V & M;
        unsigned long long S = V + U;
        const unsigned long long G = ballot64(S < V), P = ballot64(S == ~0ull);
        const unsigned long long Y = (G << 1) | ((cinW >> jj) & 1ull);
        const unsigned long long C = ((Y + P) ^ P) | Y;
        coutW |= (((G | (P & C)) >> 63) & 1ull) << jj;
        S += (C & laneBit) ? 1ull : 0ull;
        V = S | (V & ~M);
      }
      if (blk + 1 < nblk && l == 0) work[j0 >> 6] = coutW;
    }
    // zeros among this block's column bits
    const int mine = max(0, min(64, m - c00 - 64 * l));
    const unsigned long long colMask = (mine >= 64) ? ~0ull : ((1ull << mine) - 1);
    z += __popcll(~V & colMask);
    if (blk + 1 < nblk) WSYNC();   // lane 0's words are read by every lane
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) z += __shfl_xor(z, off, 64);
  return z;
}

// ------------------------------------------------------------------ bit-parallel edit distance (long sequences)
// Global edit distance (= minus the globalAlignment score with match 0, mismatch -1, gap -1; Trail.cpp:422,
// Trajectory.cpp:413) by Myers' bit-vector algorithm in Hyyro's formulation: the pattern's positions are the bits (lane l
// holds bits [64 l, 64 l + 63] of a block of 4096), the other sequence is consumed base by base; Pv / Mv are the vertical
// +1 / -1 deltas of the current column, the horizontal delta of row 0 is +1 (the matrix border of a global alignment).
// The addition's carry crosses the lanes as in wave_lcs_bitpar, the two shifts by one take the top bit of the lane
// below.  About 60 instructions per base of the consumed sequence and block.  A pattern beyond 4096 positions is taken
// in blocks (Myers' block formulation): a block hands the next the horizontal delta of its last row, +1 / 0 / -1 per
// consumed base, as two bit arrays in `work` (2 x (n + 63) / 64 words; nullptr: one block only, -1 beyond).
TALC_D int wave_edit_bitpar(const uint8_t* __restrict__ a_, int la, const uint8_t* __restrict__ b_, int lb, unsigned long long& cells,
                            unsigned long long* work_ = nullptr) {
  la = uni(la); lb = uni(lb);
  if (la == 0 || lb == 0) return la + lb;
  unsigned long long* const work = (unsigned long long*)uni_ptr(work_);
  const uint8_t* patp = uni_ptr(a_); int m = la; const uint8_t* txtp = uni_ptr(b_); int n = lb;
  {
    const long long costA = (long long)((la + 4095) >> 12) * lb, costB = (long long)((lb + 4095) >> 12) * la;
    if (costB < costA || (costB == costA && lb > la)) { patp = uni_ptr(b_); m = lb; txtp = uni_ptr(a_); n = la; }
  }
  const int nblk = (m + 4095) >> 12;
  if (nblk > 1 && !work) return -1;
  cells += (unsigned long long)la * (unsigned long long)lb;
  gcu8 ps = (gcu8)patp; gcu8 ts = (gcu8)txtp;
  const int l = lane_id_here();
  const unsigned long long laneBit = 1ull << l;
  const int nw = (n + 63) >> 6;
  int score = m;
  for (int blk = 0; blk < nblk; ++blk) {
    const int c00 = blk << 12;
    const bool lastBlk = blk + 1 == nblk;
    unsigned long long pm0, pm1, pm2, pm3, pm4;
    bitpar_masks(ps, m, c00 + 64 * l, pm0, pm1, pm2, pm3, pm4);
    const int topPos = (lastBlk ? m - 1 - c00 : 4095), topLane = topPos >> 6, topBit = topPos & 63;
    unsigned long long Pv = ~0ull, Mv = 0ull;
    for (int j0 = 0; j0 < n; j0 += 64) {
      const int jn = min(64, n - j0);
      const int bvec = (l < jn) ? (int)ts[j0 + l] : 0;
      // horizontal deltas entering this block's first row: the border's +1 for the first block
      const unsigned long long hinP = (blk > 0) ? uni64(work[j0 >> 6]) : ~0ull, hinM = (blk > 0) ? uni64(work[nw + (j0 >> 6)]) : 0ull;
      unsigned long long outP = 0, outM = 0;
      for (int jj = 0; jj < jn; ++jj) {
        const int b = lane_get(bvec, jj);
        const unsigned long long Eq = (b == 0) ? pm0 : (b == 1) ? pm1 : (b == 2) ? pm2 : (b == 3) ? pm3 : pm4;
        const unsigned hp = (unsigned)((hinP >> jj) & 1ull), hm = (unsigned)((hinM >> jj) & 1ull);
        const unsigned long long Xv = Eq | Mv;
        const unsigned long long Eqx = Eq | ((l == 0 && hm) ? 1ull : 0ull);   // a -1 entering row 0 acts like a match there
        const unsigned long long Xa = Eqx & Pv;
        unsigned long long S = Xa + Pv;
        const unsigned long long G = ballot64(S < Pv), P = ballot64(S == ~0ull);
        const unsigned long long Y = G << 1;
        const unsigned long long C = ((Y + P) ^ P) | Y;
        S += (C & laneBit) ? 1ull : 0ull;
        const unsigned long long Xh = (S ^ Pv) | Eqx;
        unsigned long long Ph = Mv | ~(Xh | Pv);
        unsigned long long Mh = Pv & Xh;
        // the delta of the block's last row
        const unsigned long long up = ballot64(l == topLane && ((Ph >> topBit) & 1ull)), dn = ballot64(l == topLane && ((Mh >> topBit) & 1ull));
        if (lastBlk) { score += (up != 0ull) ? 1 : 0; score -= (dn != 0ull) ? 1 : 0; }
        else { outP |= (up != 0ull ? 1ull : 0ull) << jj; outM |= (dn != 0ull ? 1ull : 0ull) << jj; }
        // shift by one across the lanes: lane 0 takes the delta that enters the block
        int pTop = (int)(Ph >> 63), mTop = (int)(Mh >> 63);
        pTop = lane_shr1(pTop); mTop = lane_shr1(mTop);
        if (l == 0) { pTop = (int)hp; mTop = (int)hm; }
        Ph = (Ph << 1) | (unsigned long long)(unsigned)pTop;
        Mh = (Mh << 1) | (unsigned long long)(unsigned)mTop;
        Pv = Mh | ~(Xv | Ph);
        Mv = Ph & Xv;
      }
      if (!lastBlk && l == 0) { work[j0 >> 6] = outP; work[nw + (j0 >> 6)] = outM; }
    }
    if (!lastBlk) WSYNC();
  }
  return score;
}

// ------------------------------------------------------------------ k-mer window search
// Occurrences of pat[0..K) in seq[0..len): returns the first (wantLast=false) or the last
// (wantLast=true) start index, or -1.  Replaces Finder/Pattern<Horspool> (Trail.cpp:295-298).
TALC_D uint64_t load_u64_unaligned(gcu8 p) {
  typedef uint64_t __attribute__((aligned(1))) u64u;
  return *(const u64u TALC_AS1*)p;
}

TALC_D int wave_find_window(const uint8_t* __restrict__ seq_, int len, const uint8_t* __restrict__ pat_, int K, bool wantLast) {
  gcu8 seq = (gcu8)seq_; gcu8 pat = (gcu8)pat_;
  const int l = lane_id();
  const int nwin = len - K + 1;
  if (nwin <= 0) return -1;
  // K >= 18: the first 8 bytes filter all but ~4^-8 of the windows with one unaligned 8-byte load
  const uint64_t p8 = load_u64_unaligned(pat);
  const int nchunk = (nwin + 63) >> 6;
  // four chunks' loads are issued before the first is looked at: the chunks of one search are independent, and a search
  // that stops at the first hit would otherwise wait for one load after the other (a branching graph asks this question
  // for every child whose k-mer any Trail of the search has seen: a fifth of such a launch)
  for (int c0 = 0; c0 < nchunk; c0 += 4) {
    int base[4]; uint64_t w[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = c0 + j;
      base[j] = wantLast ? (nchunk - 1 - c) * 64 : c * 64;
      const int q = base[j] + l;
      w[j] = (c < nchunk && q < nwin) ? load_u64_unaligned(seq + q) : ~p8;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int q = base[j] + l;
      bool eq = (w[j] == p8);
      if (eq) for (int i = 8; i < K; ++i) if (seq[q + i] != pat[i]) { eq = false; break; }
      const unsigned long long m = ballot64(eq);
      if (m) return wantLast ? base[j] + 63 - (int)__clzll((long long)m) : base[j] + (int)__ffsll((long long)m) - 1;
    }
  }
  return -1;
}

}  // namespace talc
