// seqan_shim.hpp — CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE)
//
// Restatement of the handful of SeqAn2 primitives TALC calls on its hot path.
// SeqAn2 (github.com/seqan/seqan, default branch, NOT pinned: reference
// README.md:26, Makefile:3) is absent from /root/reference and from this image,
// so these follow its published algorithms (include/seqan/align/*,
// include/seqan/seeds/seeds_extension.h) as recalled in SURVEY.md Appendix A.
// parity unpinned: pinned only by hand-derived KATs (tests/test_oracle_primitives.py).
//
// Call sites restated:
//   globalAlignment(score-only)  Trail.cpp:166,171,422  Trajectory.cpp:413
//   localAlignment(score-only)   Trajectory.cpp:368,525
//   extendSeed(GappedXDrop)      Trail.cpp:372-373,390-391
//   find(Horspool)               Trail.cpp:295-298
#pragma once
#include <algorithm>
#include <climits>
#include <string>
#include <vector>

namespace talc_oracle {
namespace shim {

// Score<int, Simple>(match, mismatch, gap) with LinearGaps (gap open == gap extend).
struct SimpleScore {
  int match, mismatch, gap;
};

// AlignConfig<TTop, TLeft, TRight, TBottom>:
//   top    -> first DP row initialised with 0 (leading gaps in the vertical sequence free)
//   left   -> first DP column initialised with 0
//   right  -> optimum also searched in the last column
//   bottom -> optimum also searched in the last row
// Returns the optimal score of globalAlignment(...) (Needleman-Wunsch, linear gaps).
// seqH is sequence 0 (horizontal, DP columns), seqV sequence 1 (vertical, DP rows).
inline int globalAlignmentScore(const std::string& seqH, const std::string& seqV, SimpleScore sc,
                                bool top = false, bool left = false, bool right = false,
                                bool bottom = false) {
  const size_t n = seqH.size(), m = seqV.size();
  std::vector<int> prev(n + 1), cur(n + 1);
  for (size_t j = 0; j <= n; ++j) prev[j] = top ? 0 : (int)j * sc.gap;
  int bestLastCol = prev[n];  // max over D[i][n], i = 0..m
  for (size_t i = 1; i <= m; ++i) {
    cur[0] = left ? 0 : (int)i * sc.gap;
    for (size_t j = 1; j <= n; ++j) {
      int d = prev[j - 1] + (seqH[j - 1] == seqV[i - 1] ? sc.match : sc.mismatch);
      int u = prev[j] + sc.gap;
      int l = cur[j - 1] + sc.gap;
      cur[j] = std::max(d, std::max(u, l));
    }
    bestLastCol = std::max(bestLastCol, cur[n]);
    std::swap(prev, cur);
  }
  // prev now holds the last row D[m][*]
  int best = prev[n];
  if (right) best = std::max(best, bestLastCol);
  if (bottom)
    for (size_t j = 0; j <= n; ++j) best = std::max(best, prev[j]);
  return best;
}

// localAlignment(align, Score<int,Simple>, LinearGaps()): Smith-Waterman optimum (>= 0).
inline int localAlignmentScore(const std::string& a, const std::string& b, SimpleScore sc) {
  const size_t n = a.size(), m = b.size();
  std::vector<int> prev(n + 1, 0), cur(n + 1, 0);
  int best = 0;
  for (size_t i = 1; i <= m; ++i) {
    cur[0] = 0;
    for (size_t j = 1; j <= n; ++j) {
      int d = prev[j - 1] + (a[j - 1] == b[i - 1] ? sc.match : sc.mismatch);
      int u = prev[j] + sc.gap;
      int l = cur[j - 1] + sc.gap;
      int v = std::max(0, std::max(d, std::max(u, l)));
      cur[j] = v;
      best = std::max(best, v);
    }
    std::swap(prev, cur);
  }
  return best;
}

// Seed<Simple>(beginH, beginV, endH, endV): end positions are exclusive.
struct Seed {
  long beginH, beginV, endH, endV;
};
enum ExtensionDirection { EXTEND_LEFT, EXTEND_RIGHT };

// _extendSeedGappedXDropOneDirection (seeds_extension.h).  querySeg indexes DP columns
// (V dimension), databaseSeg DP rows (H dimension).  On return extCols/extRows hold the
// "longest extension" found and the function result says whether the seed must be updated.
inline bool gappedXDropOneDirection(const std::string& querySeg, const std::string& databaseSeg,
                                    ExtensionDirection direction, SimpleScore sc, int scoreDropOff,
                                    long& extCols, long& extRows, int& extScore) {
  typedef long TSize;
  const TSize cols = (TSize)querySeg.size() + 1;
  const TSize rows = (TSize)databaseSeg.size() + 1;
  extCols = extRows = 0;
  extScore = 0;
  if (rows == 1 || cols == 1) return false;

  const int gapCost = sc.gap;
  const int undefined = INT_MIN - gapCost;

  std::vector<int> antiDiag1, antiDiag2, antiDiag3;
  TSize minCol = 1, maxCol = 2;
  TSize offset1 = 0, offset2 = 0, offset3 = 0;

  // _initAntiDiags
  antiDiag2.assign(1, 0);
  antiDiag3.assign(2, 0);
  if (-gapCost > scoreDropOff) {
    antiDiag3[0] = undefined;
    antiDiag3[1] = undefined;
  } else {
    antiDiag3[0] = gapCost;
    antiDiag3[1] = gapCost;
  }
  TSize antiDiagNo = 1;
  int best = 0;

  while (minCol < maxCol) {
    ++antiDiagNo;
    // _swapAntiDiags: 1 <- 2, 2 <- 3, 3 <- old 1
    std::vector<int> temp;
    temp.swap(antiDiag1);
    antiDiag1.swap(antiDiag2);
    antiDiag2.swap(antiDiag3);
    antiDiag3.swap(temp);
    offset1 = offset2;
    offset2 = offset3;
    offset3 = minCol - 1;
    // _initAntiDiag3
    {
      const int minScore = best - scoreDropOff;
      antiDiag3.assign((size_t)(maxCol + 1 - offset3), undefined);  // resize; interior cells are all rewritten below
      antiDiag3[0] = undefined;
      antiDiag3[(size_t)(maxCol - offset3)] = undefined;
      if ((int)antiDiagNo * gapCost > minScore) {
        if (offset3 == 0) antiDiag3[0] = (int)antiDiagNo * gapCost;
        if (antiDiagNo - maxCol == 0) antiDiag3[(size_t)(maxCol - offset3)] = (int)antiDiagNo * gapCost;
      }
    }

    int antiDiagBest = (int)antiDiagNo * gapCost;
    for (TSize col = minCol; col < maxCol; ++col) {
      const TSize i3 = col - offset3, i2 = col - offset2, i1 = col - offset1;
      TSize queryPos, dbPos;
      if (direction == EXTEND_RIGHT) {
        queryPos = col - 1;
        dbPos = antiDiagNo - col - 1;
      } else {
        queryPos = cols - 1 - col;
        dbPos = rows - 1 + col - antiDiagNo;
      }
      int tmp = std::max(antiDiag2[(size_t)(i2 - 1)], antiDiag2[(size_t)i2]) + gapCost;
      const int s = (querySeg[(size_t)queryPos] == databaseSeg[(size_t)dbPos]) ? sc.match : sc.mismatch;
      tmp = std::max(tmp, antiDiag1[(size_t)(i1 - 1)] + s);
      if (tmp < best - scoreDropOff) {
        antiDiag3[(size_t)i3] = undefined;
      } else {
        antiDiag3[(size_t)i3] = tmp;
        antiDiagBest = std::max(antiDiagBest, tmp);
      }
    }
    best = std::max(best, antiDiagBest);

    // new minCol
    while (minCol - offset3 < (TSize)antiDiag3.size() && antiDiag3[(size_t)(minCol - offset3)] == undefined &&
           minCol - offset2 - 1 < (TSize)antiDiag2.size() &&
           antiDiag2[(size_t)(minCol - offset2 - 1)] == undefined) {
      ++minCol;
    }
    // new maxCol
    while (maxCol - offset3 > 0 && antiDiag3[(size_t)(maxCol - offset3 - 1)] == undefined &&
           antiDiag2[(size_t)(maxCol - offset2 - 1)] == undefined) {
      --maxCol;
    }
    ++maxCol;

    // end of databaseSeg reached?
    minCol = std::max((int)minCol, (int)antiDiagNo + 2 - (int)rows);
    // end of querySeg reached?
    maxCol = std::min(maxCol, cols);
  }

  // find positions of longest extension
  // reached ends of both segments
  TSize longestExtensionCol = (TSize)antiDiag3.size() + offset3 - 2;
  TSize longestExtensionRow = antiDiagNo - longestExtensionCol;
  int longestExtensionScore = antiDiag3[(size_t)(longestExtensionCol - offset3)];

  if (longestExtensionScore == undefined) {
    if (antiDiag2[antiDiag2.size() - 2] != undefined) {
      // reached end of query segment
      longestExtensionCol = (TSize)antiDiag2.size() + offset2 - 2;
      longestExtensionRow = antiDiagNo - 1 - longestExtensionCol;
      longestExtensionScore = antiDiag2[(size_t)(longestExtensionCol - offset2)];
    } else if (antiDiag2.size() > 2 && antiDiag2[antiDiag2.size() - 3] != undefined) {
      // reached end of database segment
      longestExtensionCol = (TSize)antiDiag2.size() + offset2 - 3;
      longestExtensionRow = antiDiagNo - 1 - longestExtensionCol;
      longestExtensionScore = antiDiag2[(size_t)(longestExtensionCol - offset2)];
    }
  }
  if (longestExtensionScore == undefined) {
    // general case
    for (size_t i = 0; i < antiDiag1.size(); ++i) {
      if (antiDiag1[i] > longestExtensionScore) {
        longestExtensionScore = antiDiag1[i];
        longestExtensionCol = (TSize)i + offset1;
        longestExtensionRow = antiDiagNo - 2 - longestExtensionCol;
      }
    }
  }
  extScore = longestExtensionScore;
  if (longestExtensionScore != undefined) {
    extCols = longestExtensionCol;
    extRows = longestExtensionRow;
    return true;
  }
  return false;
}

// extendSeed(seed, database=seqH, query=seqV, direction, score, xdrop, GappedXDrop())
inline void extendSeed(Seed& seed, const std::string& database, const std::string& query,
                       ExtensionDirection direction, SimpleScore sc, int scoreDropOff) {
  long extCols = 0, extRows = 0;
  int extScore = 0;
  if (direction == EXTEND_LEFT) {
    const std::string databasePrefix = database.substr(0, (size_t)seed.beginH);
    const std::string queryPrefix = query.substr(0, (size_t)seed.beginV);
    if (gappedXDropOneDirection(queryPrefix, databasePrefix, EXTEND_LEFT, sc, scoreDropOff, extCols, extRows,
                                extScore)) {
      seed.beginH -= extRows;
      seed.beginV -= extCols;
    }
  } else {
    const std::string databaseSuffix = database.substr((size_t)seed.endH);
    const std::string querySuffix = query.substr((size_t)seed.endV);
    if (gappedXDropOneDirection(querySuffix, databaseSuffix, EXTEND_RIGHT, sc, scoreDropOff, extCols, extRows,
                                extScore)) {
      seed.endH += extRows;
      seed.endV += extCols;
    }
  }
}

// Finder<Dna5String> + Pattern<CharString,Horspool>: position of the first occurrence, -1 if none.
inline long findFirst(const std::string& haystack, const std::string& needle) {
  if (needle.empty() || needle.size() > haystack.size()) return -1;
  size_t p = haystack.find(needle);
  return p == std::string::npos ? -1 : (long)p;
}

}  // namespace shim
}  // namespace talc_oracle
