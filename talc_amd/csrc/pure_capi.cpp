// pure_capi.cpp — host build (g++) of the host/device-pure pieces of the product
// (talc_pure.h) plus the host table builder, exported with C linkage so the CPU test-suite can
// check them without a GPU: libtalc_pure.so.  This is product code compiled for the host, not a
// CPU fallback of the correction path (nothing in libtalc_hip.so calls it).
#include <algorithm>
#include <cstdint>
#include <vector>

#include "talc_pure.h"
#include "talc_wfa.h"

using namespace talc;

extern "C" {

int pure_is_expected_by_model(double alpha, uint32_t nextc, uint32_t cc, int classe_unexpected) {
  return is_expected_by_model(alpha, nextc, cc, classe_unexpected != 0) ? 1 : 0;
}
int pure_is_expected_by_last_node(double alpha, uint32_t nextc, uint32_t cc) {
  return is_expected_by_last_node(alpha, nextc, cc) ? 1 : 0;
}
int pure_tag_next_nodes(double alpha, double err, uint32_t minc, const uint32_t* cnt4, const uint32_t* jc4, uint32_t count,
                        int complex, int32_t* tags4, double* dist4) {
  int t[4];
  double d[4];
  int n = tag_next_nodes(alpha, err, minc, cnt4, jc4, count, complex != 0, t, d);
  for (int i = 0; i < 4; ++i) { tags4[i] = t[i]; dist4[i] = d[i]; }
  return n;
}

// gnu_sort (the product's restatement of libstdc++ std::sort) vs the real std::sort on
// (key, payload) pairs ordered by key only: both must produce the same permutation.
struct KP { int32_t key; int32_t payload; };
struct KPLess { bool operator()(const KP& a, const KP& b) const { return a.key < b.key; } };
void pure_gnu_sort_pairs(int32_t* keys, int32_t* payloads, int n) {
  std::vector<KP> v(n);
  for (int i = 0; i < n; ++i) v[i] = KP{keys[i], payloads[i]};
  gnu_sort(v.data(), n, KPLess());
  for (int i = 0; i < n; ++i) { keys[i] = v[i].key; payloads[i] = v[i].payload; }
}
void pure_std_sort_pairs(int32_t* keys, int32_t* payloads, int n) {
  std::vector<KP> v(n);
  for (int i = 0; i < n; ++i) v[i] = KP{keys[i], payloads[i]};
  std::sort(v.begin(), v.end(), KPLess());
  for (int i = 0; i < n; ++i) { keys[i] = v[i].key; payloads[i] = v[i].payload; }
}

// the depth-limit fallback of introsort: std::__partial_sort(first, last, last) == heapsort
void pure_gnu_heapsort_pairs(int32_t* keys, int32_t* payloads, int n) {
  std::vector<KP> v(n);
  for (int i = 0; i < n; ++i) v[i] = KP{keys[i], payloads[i]};
  gs_heapsort(v.data(), 0, n, KPLess());
  for (int i = 0; i < n; ++i) { keys[i] = v[i].key; payloads[i] = v[i].payload; }
}
void pure_std_heapsort_pairs(int32_t* keys, int32_t* payloads, int n) {
  std::vector<KP> v(n);
  for (int i = 0; i < n; ++i) v[i] = KP{keys[i], payloads[i]};
  std::partial_sort(v.begin(), v.end(), v.end(), KPLess());
  for (int i = 0; i < n; ++i) { keys[i] = v[i].key; payloads[i] = v[i].payload; }
}

int pure_gardening(uint32_t maxb, int n, const double* scores, const double* dists, uint32_t* kept, int32_t* isComplex) {
  std::vector<ValIdx> wv(n + 1);
  std::vector<Rank4> r1(n + 1), r2(n + 1);
  bool cx = false;
  int nk = gardening(maxb, n, scores, dists, wv.data(), r1.data(), r2.data(), kept, &cx);
  *isComplex = cx ? 1 : 0;
  return nk;
}

// anchors ordering (sortAnchorsByNearest, Explorer.cpp:402-411) on (pos,count) records
void pure_sort_anchors(double cc, uint32_t* pos, uint32_t* count, int n) {
  std::vector<AnchorRec> a(n);
  for (int i = 0; i < n; ++i) a[i] = AnchorRec{0, 0, pos[i], count[i]};
  gnu_sort(a.data(), n, LessAnchor{cc});
  for (int i = 0; i < n; ++i) { pos[i] = a[i].pos; count[i] = a[i].count; }
}

// wavefront statement of the unit-cost x-drop extension (talc_wfa.h); out4 = moved, extCols, extRows, score
void pure_wfa_xdrop(const uint8_t* q, int qlen, const uint8_t* d, int dlen, int x, int32_t* out4) {
  const WfaResult r = wfa_xdrop_scalar(q, qlen, d, dlen, x);
  out4[0] = r.moved; out4[1] = r.extCols; out4[2] = r.extRows; out4[3] = r.score;
}
}  // extern "C"
