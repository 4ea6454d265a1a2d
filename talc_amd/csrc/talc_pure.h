// talc_pure.h — pieces of the path search that are plain arithmetic / serial logic and compile
// for both the host (unit tests, libtalc_pure.so) and the device (one lane of a wavefront).
//
//  * the count model        isExpectedbyMyModel / isExpectedbyMyLastNode   Explorer.cpp:1185-1217
//  * successor tagging      tagNextNodes                                    Explorer.cpp:1226-1298
//  * gnu_sort               a restatement of libstdc++'s std::sort (introsort: median-of-3,
//                           unguarded partition, depth limit 2*lg(n) with heapsort fallback,
//                           final insertion sort, threshold 16).  The reference ranks Trails
//                           and anchors with the unstable std::sort (Explorer.cpp:385,409,
//                           744-768) and the order of equal elements decides which Trails
//                           survive, so the device must reproduce that exact permutation.
//  * gardening              doABitOfGardening                               Explorer.cpp:773-865
//
// Floating point: IEEE double, no contraction (-ffp-contract=off), sqrt and division correctly
// rounded on both sides; std::pow(x,2) of the reference is x*x (g++ folds it).
#pragma once
#include <cstddef>
#include <math.h>

#include "talc_common.h"

namespace talc {

// ------------------------------------------------------------------ count model
TALC_HD bool is_expected_by_model(double ALPHA, uint32_t nextc, uint32_t cc, bool classeUnexpected) {
  if (cc <= 3) {
    if (classeUnexpected) return ((double)nextc <= ((double)(cc + 0.5) + ALPHA * sqrt((double)(cc + 0.5))));
    return ((double)nextc >= ((double)(cc - 0.5) + (1 - ALPHA) * sqrt((double)(cc - 0.5))));
  }
  if (classeUnexpected) {
    const double t = (ALPHA / 2 + sqrt((double)(cc + 0.96)));
    return ((double)nextc <= t * t);
  }
  const double t = (ALPHA / 2 - sqrt((double)(cc + 0.02)));
  return ((double)nextc >= t * t);
}

TALC_HD bool is_expected_by_last_node(double ALPHA, uint32_t nextc, uint32_t cc) {
  bool e = true;
  if (cc <= 3) {
    e &= ((double)nextc <= ((double)(cc + 0.5) + ALPHA * sqrt((double)(cc + 0.5))));
    e &= ((double)nextc >= ((double)(cc - 0.5) + (1 - ALPHA) * sqrt((double)(cc - 0.5))));
  }
  if (cc > 3) {
    const double t1 = (ALPHA / 2 + sqrt((double)(cc + 0.96)));
    const double t2 = (ALPHA / 2 - sqrt((double)(cc + 0.02)));
    e &= ((double)nextc <= t1 * t1);
    e &= ((double)nextc >= t2 * t2);
  }
  return e;
}

// tags use the reference's Status enumerators (utils.hpp:54)
enum : int { TAG_EXPECTED = 0, TAG_UNEXPECTED = 1, TAG_BREAKPOINT = 7, TAG_NONE = -1 };

// Returns the number of tags produced (0 when no successor reaches MIN_COUNT, else 4).
// dist[b] is only computed for successors present in the table (count >= MIN_COUNT): the
// reference only ever reads it for successors that become Trails (Explorer.cpp:570-574,641-646),
// and those all have count >= MIN_COUNT.  dist may be null (the device's steps compute a child's term when they make it).
// The count model enters through `model(nextc, cc, classeUnexpected)` = isExpectedbyMyModel: the formula itself
// (ModelFormula), or its two thresholds for this call's `count` and lambda_noise read from a table (ModelThresholds,
// device: k_build_thresholds) — tagNextNodes only ever asks (x, count, false) and (x, lambda_noise, true).
struct ModelFormula {
  double ALPHA;
  TALC_HD bool operator()(uint32_t nextc, uint32_t cc, bool classeUnexpected) const { return is_expected_by_model(ALPHA, nextc, cc, classeUnexpected); }
};
// minExpected = the smallest nextc with isExpectedbyMyModel(nextc, count, false); belowUnexpected = the number of
// nextc = 0, 1, 2, ... with isExpectedbyMyModel(nextc, lambda_noise, true) (both predicates are monotone in nextc)
struct ModelThresholds {
  uint32_t minExpected, belowUnexpected;
  TALC_HD bool operator()(uint32_t nextc, uint32_t, bool classeUnexpected) const { return classeUnexpected ? (nextc < belowUnexpected) : (nextc >= minExpected); }
};
TALC_HD uint32_t lambda_noise_of(uint32_t count, double ERR) { return (uint32_t)(int)((double)count * ERR); }

template <class Model>
TALC_HD int tag_next_nodes_with(const Model& model, double ERR, uint32_t MINC, const uint32_t cnt[4], const uint32_t jc[4],
                                uint32_t count, bool complex, int tags[4], double* dist) {
  int counter = 0;
  uint32_t lambda_noise = 0, nbExpected = 0, nbBreakpoints = 0, nbUnexpected = 0;
  for (int i = 0; i < 4; ++i) { tags[i] = TAG_NONE; if (dist) dist[i] = 0; if (cnt[i] >= MINC) counter++; }
  if (counter == 0) return 0;
  lambda_noise = lambda_noise_of(count, ERR);
  for (int b = 0; b < 4; ++b) {
    const uint32_t nextc = cnt[b];
    if (nextc >= MINC) {
      if (dist) dist[b] = fabs((double)count - (double)nextc) / sqrt((double)count);
      if (model(nextc, count, false) || (counter == 1)) {
        tags[b] = TAG_EXPECTED; ++nbExpected;
      } else if (lambda_noise >= MINC) {
        if (!model(nextc, lambda_noise, true) || (jc[b] > 0)) { tags[b] = TAG_BREAKPOINT; ++nbBreakpoints; }
        else { tags[b] = TAG_UNEXPECTED; ++nbUnexpected; }
      } else {
        tags[b] = TAG_BREAKPOINT; ++nbBreakpoints;
      }
    } else
      tags[b] = TAG_UNEXPECTED;
  }
  if ((nbExpected == 0) & (nbBreakpoints == 1)) {
    for (int t = 0; t < 4; ++t) if (tags[t] == TAG_BREAKPOINT) tags[t] = TAG_EXPECTED;
  }
  if ((nbExpected == 1) & (nbUnexpected > 0) & !complex) {
    int sum = 0;
    int index = 0;
    for (int i = 0; i < 4; ++i) {
      if (tags[i] == TAG_UNEXPECTED) {
        if (sum == 0) index = i;
        sum += (int)cnt[i];
        if (cnt[index] < cnt[i]) index = i;
      }
    }
    if (!model((uint32_t)sum, lambda_noise, true)) tags[index] = TAG_BREAKPOINT;
  }
  return 4;
}
TALC_HD int tag_next_nodes(double ALPHA, double ERR, uint32_t MINC, const uint32_t cnt[4], const uint32_t jc[4],
                           uint32_t count, bool complex, int tags[4], double dist[4]) {
  return tag_next_nodes_with(ModelFormula{ALPHA}, ERR, MINC, cnt, jc, count, complex, tags, dist);
}

// ------------------------------------------------------------------ libstdc++ std::sort, restated
template <class T>
TALC_HD void gs_swap(T& a, T& b) { T t = a; a = b; b = t; }

template <class T, class Less>
TALC_HD void gs_unguarded_linear_insert(T* a, int last, Less less) {
  T val = a[last];
  int next = last - 1;
  while (less(val, a[next])) { a[last] = a[next]; last = next; --next; }
  a[last] = val;
}
template <class T, class Less>
TALC_HD void gs_insertion_sort(T* a, int first, int last, Less less) {
  if (first == last) return;
  for (int i = first + 1; i != last; ++i) {
    if (less(a[i], a[first])) {
      T val = a[i];
      for (int j = i; j > first; --j) a[j] = a[j - 1];
      a[first] = val;
    } else
      gs_unguarded_linear_insert(a, i, less);
  }
}
template <class T, class Less>
TALC_HD void gs_push_heap(T* a, int first, int holeIndex, int topIndex, T value, Less less) {
  int parent = (holeIndex - 1) / 2;
  while (holeIndex > topIndex && less(a[first + parent], value)) {
    a[first + holeIndex] = a[first + parent];
    holeIndex = parent;
    parent = (holeIndex - 1) / 2;
  }
  a[first + holeIndex] = value;
}
template <class T, class Less>
TALC_HD void gs_adjust_heap(T* a, int first, int holeIndex, int len, T value, Less less) {
  const int topIndex = holeIndex;
  int secondChild = holeIndex;
  while (secondChild < (len - 1) / 2) {
    secondChild = 2 * (secondChild + 1);
    if (less(a[first + secondChild], a[first + (secondChild - 1)])) secondChild--;
    a[first + holeIndex] = a[first + secondChild];
    holeIndex = secondChild;
  }
  if ((len & 1) == 0 && secondChild == (len - 2) / 2) {
    secondChild = 2 * (secondChild + 1);
    a[first + holeIndex] = a[first + (secondChild - 1)];
    holeIndex = secondChild - 1;
  }
  gs_push_heap(a, first, holeIndex, topIndex, value, less);
}
template <class T, class Less>
TALC_HD void gs_heapsort(T* a, int first, int last, Less less) {  // __partial_sort(first, last, last)
  const int len = last - first;
  if (len >= 2) {  // __make_heap
    int parent = (len - 2) / 2;
    while (true) {
      T value = a[first + parent];
      gs_adjust_heap(a, first, parent, len, value, less);
      if (parent == 0) break;
      parent--;
    }
  }
  int l = last;  // __sort_heap
  while (l - first > 1) {
    --l;
    T value = a[l];
    a[l] = a[first];
    gs_adjust_heap(a, first, 0, l - first, value, less);
  }
}
template <class T, class Less>
TALC_HD void gnu_sort(T* a, int n, Less less) {
  if (n <= 0) return;
  // __introsort_loop with an explicit stack (sub-ranges are disjoint, so the processing order of
  // the pending right-hand parts does not change the result)
  int stFirst[64], stLast[64], stDepth[64];
  int sp = 0;
  int lg = 0;
  for (int m = n; m > 1; m >>= 1) ++lg;
  stFirst[0] = 0; stLast[0] = n; stDepth[0] = 2 * lg; sp = 1;
  while (sp > 0) {
    --sp;
    int first = stFirst[sp], last = stLast[sp], depth = stDepth[sp];
    while (last - first > 16) {
      if (depth == 0) { gs_heapsort(a, first, last, less); break; }
      --depth;
      // __unguarded_partition_pivot
      const int mid = first + (last - first) / 2;
      {  // __move_median_to_first(first, first+1, mid, last-1)
        const int ia = first + 1, ib = mid, ic = last - 1;
        if (less(a[ia], a[ib])) {
          if (less(a[ib], a[ic])) gs_swap(a[first], a[ib]);
          else if (less(a[ia], a[ic])) gs_swap(a[first], a[ic]);
          else gs_swap(a[first], a[ia]);
        } else if (less(a[ia], a[ic])) gs_swap(a[first], a[ia]);
        else if (less(a[ib], a[ic])) gs_swap(a[first], a[ic]);
        else gs_swap(a[first], a[ib]);
      }
      int lo = first + 1, hi = last;
      while (true) {  // __unguarded_partition(first+1, last, pivot=first)
        while (less(a[lo], a[first])) ++lo;
        --hi;
        while (less(a[first], a[hi])) --hi;
        if (!(lo < hi)) break;
        gs_swap(a[lo], a[hi]);
        ++lo;
      }
      const int cut = lo;
      // recurse on [cut, last), continue with [first, cut)
      stFirst[sp] = cut; stLast[sp] = last; stDepth[sp] = depth; ++sp;
      last = cut;
    }
  }
  // __final_insertion_sort
  if (n > 16) {
    gs_insertion_sort(a, 0, 16, less);
    for (int i = 16; i != n; ++i) gs_unguarded_linear_insert(a, i, less);
  } else
    gs_insertion_sort(a, 0, n, less);
}

// ------------------------------------------------------------------ gardening (Explorer.cpp:773-865)
struct Rank4 { uint32_t idx, r1, r2, sum; };
struct ValIdx { double v; uint32_t idx; };
struct LessRankR1 { TALC_HD bool operator()(const Rank4& a, const Rank4& b) const { return a.r1 < b.r1; } };
struct LessRankR2 { TALC_HD bool operator()(const Rank4& a, const Rank4& b) const { return a.r2 < b.r2; } };
struct GreaterVal { TALC_HD bool operator()(const ValIdx& a, const ValIdx& b) const { return a.v > b.v; } };
struct LessVal { TALC_HD bool operator()(const ValIdx& a, const ValIdx& b) const { return a.v < b.v; } };

// scores[n], dists[n]: m_lastScore / m_distance of the new competing paths, in order.
// work: n ValIdx + 2n Rank4.  kept: capacity >= n + MAXB.  Returns the number of kept indices.
TALC_HD int gardening(uint32_t MAXB, int n, const double* scores, const double* dists, ValIdx* wv, Rank4* rankings,
                      Rank4* newr, uint32_t* kept, bool* isComplexOut) {
  *isComplexOut = false;
  if (n <= 0) return 0;
  uint32_t nb = (uint32_t)n < MAXB ? (uint32_t)n : MAXB;
  for (int t = 0; t < n; ++t) { rankings[t].idx = (uint32_t)t; rankings[t].r1 = rankings[t].r2 = rankings[t].sum = 0; }
  // dense ranks with ties by score (descending); the tie order of the sort cannot change a dense rank
  for (int t = 0; t < n; ++t) { wv[t].v = scores[t]; wv[t].idx = (uint32_t)t; }
  gnu_sort(wv, n, GreaterVal());
  {
    uint32_t rk = 0;
    rankings[wv[0].idx].r1 = 0;
    for (int t = 1; t < n; ++t) { if (!(wv[t].v == wv[t - 1].v)) rk = rk + 1; rankings[wv[t].idx].r1 = rk; }
  }
  for (int t = 0; t < n; ++t) { wv[t].v = dists[t]; wv[t].idx = (uint32_t)t; }
  gnu_sort(wv, n, LessVal());
  {
    uint32_t rk = 0;
    rankings[wv[0].idx].r2 = 0;
    for (int t = 1; t < n; ++t) { if (!(wv[t].v == wv[t - 1].v)) rk = rk + 1; rankings[wv[t].idx].r2 = rk; }
  }
  int nn = 0;
  for (int t = 0; t < n; ++t) {
    rankings[t].sum = rankings[t].r1 + rankings[t].r2;
    if ((rankings[t].sum == 0) || ((uint32_t)n <= MAXB)) newr[nn++] = rankings[t];
  }
  int nk = 0;
  if (nn == 0) {
    gnu_sort(rankings, n, LessRankR1());
    uint32_t s = 0;
    bool ties = false;
    do {
      if ((s <= nb) || ties) newr[nn++] = rankings[s];
      if (s < (uint32_t)n - 1) ties = (rankings[s + 1].r1 == rankings[s].r1);
      ++s;
    } while (((s <= nb) || ties) & (s < (uint32_t)n));
    if ((uint32_t)nn > MAXB) {
      if (newr[0].r1 != newr[MAXB].r1) {
        --nn;
        ties = true;
        while (((uint32_t)nn >= MAXB) & ties) {
          ties = (newr[nn - 1].r1 == newr[nn - 2].r1);
          ties |= ((uint32_t)nn >= MAXB);
          if (ties) --nn;
        }
      }
      if (((uint32_t)nn > MAXB) && (newr[0].r1 == newr[MAXB].r1)) {
        *isComplexOut = true;
        gnu_sort(newr, nn, LessRankR2());
        for (uint32_t t = 0; t < MAXB; ++t) kept[nk++] = newr[t].idx;
      }
    }
    for (int t = 0; t < nn; ++t) kept[nk++] = newr[t].idx;
  } else {
    for (int t = 0; t < nn; ++t) kept[nk++] = newr[t].idx;
  }
  return nk;
}

// ------------------------------------------------------------------ anchors ordering (Explorer.cpp:402-411)
struct AnchorRec {
  uint64_t kmer;    // packed natural orientation; bases under an N are 0
  uint64_t nmask;   // bit i set <=> base i (0 = first) is N
  uint32_t pos;
  uint32_t count;   // the count the reference records with the anchor (m_coverage[anchor number], Explorer.cpp:454,520)
};
// (24 bytes on purpose: a 32-byte record with the anchor's own coverage count in it made k_search 2 ms slower on config 2
//  — same-box A/B, whatever the stride of the waves' scratch slots — while saving one load per search)
struct LessAnchor {
  double cc;
  TALC_HD bool operator()(const AnchorRec& l, const AnchorRec& r) const {
    int dl = (int)cc - (int)l.count; if (dl < 0) dl = -dl;
    int dr = (int)cc - (int)r.count; if (dr < 0) dr = -dr;
    return dl < dr;
  }
};

}  // namespace talc
