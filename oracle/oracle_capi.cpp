// oracle_capi.cpp — CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).  parity unpinned.
// Plain C entry points over talc_oracle.{hpp,cpp} so tests/ and bench.py's cpu_baseline leg can
// drive the oracle through ctypes.  Nothing in talc_amd/ links or loads this library.
#include <omp.h>

#include <cstdint>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <sstream>
#include <string>
#include <vector>

#include "seqan_shim.hpp"
#include "talc_oracle.hpp"

using namespace talc_oracle;

extern "C" {

// mirror of talc_oracle::Params as a POD (same field order as include/talc_hip.h's talc_params)
struct orc_params {
  uint32_t k;
  uint32_t min_count;
  double alpha;
  uint32_t window_size;
  double sr_error_rate;
  double min_inner_score;
  double min_border_score;
  uint32_t max_nb_competing_paths;
  int32_t use_junctions;
  int32_t reverse;
  uint32_t min_start_anchors;
  uint32_t max_start_anchors;
  uint32_t max_in_count;
  uint32_t max_nb_border_paths;
  uint32_t max_nb_inner_paths;
  uint32_t check_interval;
  double allowed_failure_rate;
  int32_t max_nb_border_failures;
  uint32_t coloured_count_thr;
  uint32_t max_border_length;
};

static Params toParams(const orc_params* p) {
  Params P;
  if (!p) return P;
  P.K = p->k;
  P.gp_MIN_COUNT = p->min_count;
  P.gp_ALPHA = p->alpha;
  P.gp_WINDOW_SIZE = p->window_size;
  P.gp_SR_ERROR_RATE = p->sr_error_rate;
  P.gp_MIN_INNER_SCORE = p->min_inner_score;
  P.gp_MIN_BORDER_SCORE = p->min_border_score;
  P.gp_MAX_NB_COMPETING_PATHS = p->max_nb_competing_paths;
  P.gp_useJunctions = p->use_junctions != 0;
  P.gp_reverse = p->reverse != 0;
  P.p_MIN_START_ANCHORS = p->min_start_anchors;
  P.p_MAX_START_ANCHORS = p->max_start_anchors;
  P.p_MAX_IN_COUNT = p->max_in_count;
  P.p_MAX_NB_OF_BORDER_PATHS = p->max_nb_border_paths;
  P.p_MAX_NB_OF_INNER_PATHS = p->max_nb_inner_paths;
  P.p_CHECK_INTERVAL = p->check_interval;
  P.p_ALLOWED_FAILURE_RATE = p->allowed_failure_rate;
  P.p_MAX_NB_BORDER_FAILURES = p->max_nb_border_failures;
  P.colouredCountThr = p->coloured_count_thr;
  P.maxBorderLength = p->max_border_length;
  return P;
}

void orc_params_default(orc_params* p) {
  Params P;
  p->k = P.K;
  p->min_count = P.gp_MIN_COUNT;
  p->alpha = P.gp_ALPHA;
  p->window_size = P.gp_WINDOW_SIZE;
  p->sr_error_rate = P.gp_SR_ERROR_RATE;
  p->min_inner_score = P.gp_MIN_INNER_SCORE;
  p->min_border_score = P.gp_MIN_BORDER_SCORE;
  p->max_nb_competing_paths = P.gp_MAX_NB_COMPETING_PATHS;
  p->use_junctions = 0;
  p->reverse = 0;
  p->min_start_anchors = P.p_MIN_START_ANCHORS;
  p->max_start_anchors = P.p_MAX_START_ANCHORS;
  p->max_in_count = P.p_MAX_IN_COUNT;
  p->max_nb_border_paths = P.p_MAX_NB_OF_BORDER_PATHS;
  p->max_nb_inner_paths = P.p_MAX_NB_OF_INNER_PATHS;
  p->check_interval = P.p_CHECK_INTERVAL;
  p->allowed_failure_rate = P.p_ALLOWED_FAILURE_RATE;
  p->max_nb_border_failures = P.p_MAX_NB_BORDER_FAILURES;
  p->coloured_count_thr = P.colouredCountThr;
  p->max_border_length = P.maxBorderLength;
}

// ---------------------------------------------------------------- table
void* orc_table_new(int backend) { return new Table(backend == 0 ? Table::MAP : Table::FLAT); }
void orc_table_free(void* t) { delete (Table*)t; }
uint64_t orc_table_size(void* t) { return ((Table*)t)->size(); }

// buildCDBG + decolourRepeatsFromDBG (main.cpp:231-232)
int orc_table_build(void* t, const char* dump, const char* jdump, const orc_params* p, int64_t* stats3) {
  Params P = toParams(p);
  BuildStats st = buildCDBG(*(Table*)t, dump ? dump : "", jdump ? jdump : "", P);
  decolourRepeatsFromDBG(*(Table*)t, P);
  if (stats3) { stats3[0] = st.onlineCounter; stats3[1] = st.actualCounter; stats3[2] = st.badLines; }
  return 0;
}
// arrays of packed k-mers; counts already filtered by the caller or not: the MIN_COUNT rule of
// Jellyfish.cpp:260 is applied here.
void orc_table_insert_packed(void* t, const uint64_t* keys, const uint32_t* counts, uint64_t n,
                             const orc_params* p, int sorted_hint) {
  Params P = toParams(p);
  std::vector<uint64_t> k2; std::vector<uint32_t> c2;
  k2.reserve(n); c2.reserve(n);
  for (uint64_t i = 0; i < n; ++i) if (counts[i] >= P.gp_MIN_COUNT) { k2.push_back(keys[i]); c2.push_back(counts[i]); }
  ((Table*)t)->insertPacked(k2.data(), c2.data(), nullptr, k2.size(), P.K, sorted_hint != 0);
}
static TSeq unpack(uint64_t key, unsigned K) {
  static const char D[4] = {'A', 'C', 'G', 'T'};
  TSeq s(K, 'A');
  for (unsigned i = 0; i < K; ++i) s[K - 1 - i] = D[(key >> (2 * i)) & 3];
  return s;
}
void orc_table_colour_packed(void* t, const uint64_t* jkeys, const int64_t* jcounts, uint64_t n,
                             const orc_params* p) {
  Params P = toParams(p);
  std::vector<std::pair<TSeq, long>> js;
  for (uint64_t i = 0; i < n; ++i) js.push_back(std::make_pair(unpack(jkeys[i], P.K), (long)jcounts[i]));
  colourJunctions(*(Table*)t, js, P);
}
void orc_table_decolour(void* t, const orc_params* p) { decolourRepeatsFromDBG(*(Table*)t, toParams(p)); }
void orc_table_lookup_packed(void* t, const uint64_t* keys, uint64_t n, uint32_t K, uint32_t* counts,
                             uint32_t* jcounts) {
  for (uint64_t i = 0; i < n; ++i) {
    colouredCount c = ((Table*)t)->at(unpack(keys[i], K));
    counts[i] = c.first;
    jcounts[i] = c.second;
  }
}
// getNextCounts (Jellyfish.cpp:299-321) for a k-mer given as text; direction 0 LEFT, 1 RIGHT
void orc_next_counts(void* t, const orc_params* p, const char* kmer, int direction, uint32_t* counts4,
                     uint32_t* jcounts4) {
  Ctx C; C.P = toParams(p); C.dBG = (Table*)t;
  std::vector<colouredCount> v = getNextCounts(C, toDna5(kmer), direction ? RIGHT : LEFT);
  for (int b = 0; b < 4; ++b) { counts4[b] = v[b].first; jcounts4[b] = v[b].second; }
}

// ---------------------------------------------------------------- per-read surface
// Read::reCoverage (Read.cpp:174-195): counts/jcounts get L-K+1 entries; returns m_nbInKmers
int orc_coverage(void* t, const orc_params* p, const char* bases, uint64_t len, uint32_t* counts,
                 uint32_t* jcounts) {
  Ctx C; C.P = toParams(p); C.dBG = (Table*)t;
  TSeq seq = toDna5(std::string(bases, len));
  if (seq.size() < C.P.K) return 0;
  Read r(C, "", seq);
  r.reCoverage();
  const auto& cov = r.getCoverage();
  for (size_t i = 0; i < cov.size(); ++i) { counts[i] = cov[i].first; if (jcounts) jcounts[i] = cov[i].second; }
  return r.nbInKmers();
}

// Read::defineStructure2 inspection: regions (start,end pairs) and the noise threshold
int orc_structure(void* t, const orc_params* p, const char* bases, uint64_t len, uint32_t* regions,
                  uint32_t max_regions, double* thr, int32_t* ok) {
  Ctx C; C.P = toParams(p); C.dBG = (Table*)t;
  TSeq seq = toDna5(std::string(bases, len));
  Read r(C, "", seq);
  *ok = 0; *thr = 0;
  if (seq.size() <= C.P.K) return 0;
  if (!r.reCoverage()) return 0;
  *ok = r.defineStructure2() ? 1 : 0;
  *thr = r.getPriorNoise();
  const auto& reg = r.getRegions();
  uint32_t n = 0;
  for (auto& x : reg) { if (n < max_regions) { regions[2 * n] = std::get<0>(x); regions[2 * n + 1] = std::get<1>(x); } n++; }
  return (int)n;
}

// main.cpp:247-308 over a batch (OpenMP schedule(dynamic) like the reference).
// bases: concatenated raw characters; offsets[n+1].  out: corrected (or passed-through) sequences
// concatenated in input order; out_offsets[n+1].  Returns 0, or -1 if out_cap is too small
// (out_offsets[n] then holds the needed size).
int orc_correct_batch(void* t, const orc_params* p, const char* bases, const uint64_t* offsets, uint32_t n,
                      char* out, uint64_t out_cap, uint64_t* out_offsets, int32_t* status, int nthreads) {
  Ctx C; C.P = toParams(p); C.dBG = (Table*)t;
  std::vector<TSeq> seqs(n);
  if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic) num_threads(nthreads)
  for (long r = 0; r < (long)n; ++r) {
    seqs[r] = toDna5(std::string(bases + offsets[r], offsets[r + 1] - offsets[r]));
    status[r] = (int32_t)correctOneRead(C, "", seqs[r]);
  }
  uint64_t pos = 0;
  for (uint32_t r = 0; r < n; ++r) { out_offsets[r] = pos; pos += seqs[r].size(); }
  out_offsets[n] = pos;
  if (pos > out_cap) return -1;
  for (uint32_t r = 0; r < n; ++r) memcpy(out + out_offsets[r], seqs[r].data(), seqs[r].size());
  return 0;
}

// the same with the rows of Read::outputBasicReadStats (Read.cpp:418-433; call commented out at main.cpp:305):
// stats5[5 r ..] = {row written (L > K), raw length, span of the IN regions, number of IN regions, corrected length}
int orc_correct_batch_stats(void* t, const orc_params* p, const char* bases, const uint64_t* offsets, uint32_t n,
                            char* out, uint64_t out_cap, uint64_t* out_offsets, int32_t* status, int nthreads, int64_t* stats5) {
  Ctx C; C.P = toParams(p); C.dBG = (Table*)t;
  std::vector<TSeq> seqs(n);
  if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic) num_threads(nthreads)
  for (long r = 0; r < (long)n; ++r) {
    seqs[r] = toDna5(std::string(bases + offsets[r], offsets[r + 1] - offsets[r]));
    BasicReadStats bs;
    status[r] = (int32_t)correctOneRead(C, "", seqs[r], nullptr, &bs);
    stats5[5 * r] = bs.written ? 1 : 0; stats5[5 * r + 1] = bs.rawLength; stats5[5 * r + 2] = bs.nbInKmersBefore;
    stats5[5 * r + 3] = bs.nbSReg; stats5[5 * r + 4] = bs.corrLength;
  }
  uint64_t pos = 0;
  for (uint32_t r = 0; r < n; ++r) { out_offsets[r] = pos; pos += seqs[r].size(); }
  out_offsets[n] = pos;
  if (pos > out_cap) return -1;
  for (uint32_t r = 0; r < n; ++r) memcpy(out + out_offsets[r], seqs[r].data(), seqs[r].size());
  return 0;
}

// textual trace of one read (regions, anchors, per-gap results) for divergence hunting
int64_t orc_trace_read(void* t, const orc_params* p, const char* bases, uint64_t len, int steps, char* buf,
                       uint64_t cap) {
  Ctx C; C.P = toParams(p); C.dBG = (Table*)t;
  TSeq seq = toDna5(std::string(bases, len));
  Trace tr; tr.enabled = true; tr.steps = steps != 0;
  ReadStatus st = correctOneRead(C, "", seq, &tr);
  std::ostringstream os;
  os << "STATUS " << (int)st << "\n";
  for (auto& e : tr.ev) {
    os << e.kind << " " << e.a << " " << e.b << " " << e.c << " " << e.d << " ";
    char x[64]; snprintf(x, sizeof x, "%a", e.x); os << x << " " << e.s << "\n";
  }
  os << "OUT " << seq << "\n";
  std::string s = os.str();
  if (s.size() + 1 <= cap) memcpy(buf, s.c_str(), s.size() + 1);
  return (int64_t)s.size() + 1;
}

void orc_ub_counters(int64_t* out3) {
  out3[0] = ubCounters().infixClamped; out3[1] = ubCounters().seedTooShort; out3[2] = ubCounters().gardeningOOB;
}

// ---------------------------------------------------------------- primitives (for KATs and device unit tests)
int orc_global_alignment(const char* h, const char* v, int match, int mismatch, int gap, int top, int left,
                         int right, int bottom) {
  return shim::globalAlignmentScore(h, v, {match, mismatch, gap}, top, left, right, bottom);
}
int orc_local_alignment(const char* a, const char* b, int match, int mismatch, int gap) {
  return shim::localAlignmentScore(a, b, {match, mismatch, gap});
}
// extendSeed on Seed(bH,bV,eH,eV); dir 0 LEFT 1 RIGHT; seed4 updated in place
void orc_extend_seed(const char* database, const char* query, int64_t* seed4, int dir, int match, int mismatch,
                     int gap, int xdrop) {
  shim::Seed s = {(long)seed4[0], (long)seed4[1], (long)seed4[2], (long)seed4[3]};
  shim::extendSeed(s, database, query, dir ? shim::EXTEND_RIGHT : shim::EXTEND_LEFT, {match, mismatch, gap}, xdrop);
  seed4[0] = s.beginH; seed4[1] = s.beginV; seed4[2] = s.endH; seed4[3] = s.endV;
}
// _extendSeedGappedXDropOneDirection, EXTEND_RIGHT, on whole segments; out4 = moved, extCols, extRows, extScore
void orc_xdrop_right(const char* query, const char* database, int match, int mismatch, int gap, int xdrop, int32_t* out4) {
  long ec = 0, er = 0; int es = 0;
  const bool m = shim::gappedXDropOneDirection(query, database, shim::EXTEND_RIGHT, {match, mismatch, gap}, xdrop, ec, er, es);
  out4[0] = m ? 1 : 0; out4[1] = (int32_t)ec; out4[2] = (int32_t)er; out4[3] = es;
}
// getSeedAndExtension (Trail.cpp:341): out3 = {len(refExtension), len(histExtension), posOnRef}; returns score
double orc_seed_and_extension(const char* reference, const char* candidate, int xdrop, int direction,
                              uint32_t seedSize, int64_t* out3, int32_t* stopThere) {
  auto r = getSeedAndExtension(reference, candidate, xdrop, direction ? RIGHT : LEFT, seedSize);
  out3[0] = (int64_t)std::get<0>(r).size(); out3[1] = (int64_t)std::get<1>(r).size(); out3[2] = std::get<2>(r);
  *stopThere = std::get<4>(r) ? 1 : 0;
  return std::get<3>(r);
}
int orc_is_expected_by_model(const orc_params* p, uint32_t nextc, uint32_t cc, int classe_unexpected) {
  return isExpectedbyMyModel(toParams(p), nextc, cc, classe_unexpected ? UNEXPECTED : EXPECTED) ? 1 : 0;
}
int orc_is_expected_by_last_node(const orc_params* p, uint32_t nextc, uint32_t cc) {
  return isExpectedbyMyLastNode(toParams(p), nextc, cc) ? 1 : 0;
}
// tagNextNodes: tags4 (Status enum values, -1 if no tag), dist4
void orc_tag_next_nodes(const orc_params* p, const uint32_t* counts4, const uint32_t* jcounts4, uint32_t count,
                        int complex, int32_t* tags4, double* dist4) {
  std::vector<colouredCount> nc;
  for (int b = 0; b < 4; ++b) nc.push_back(colouredCount(counts4[b], jcounts4[b]));
  std::vector<std::pair<Status, double>> tags;
  tagNextNodes(toParams(p), tags, nc, count, complex != 0);
  for (int b = 0; b < 4; ++b) {
    if ((size_t)b < tags.size()) { tags4[b] = (int32_t)tags[b].first; dist4[b] = tags[b].second; }
    else { tags4[b] = -1; dist4[b] = 0; }
  }
}
// doABitOfGardening on (score, distance) pairs; kept[] receives the indices; returns count, *complex set
int orc_gardening(const orc_params* p, const double* scores, const double* dists, uint32_t n, uint32_t* kept,
                  uint32_t cap, int32_t* isComplex) {
  std::vector<Trail> paths(n);
  for (uint32_t i = 0; i < n; ++i) { paths[i].setLastScore(scores[i]); paths[i].recordDistance(dists[i]); }
  std::vector<unsigned int> idx;
  bool c = doABitOfGardening(toParams(p), idx, paths);
  *isComplex = c ? 1 : 0;
  for (size_t i = 0; i < idx.size() && i < cap; ++i) kept[i] = idx[i];
  return (int)idx.size();
}
// sortAnchorsByNearest (Explorer.cpp:402-411) with the real std::sort on (pos,count) records
void orc_sort_anchors(double cc, uint32_t* pos, uint32_t* count, int n) {
  std::vector<anchorTuple> a;
  for (int i = 0; i < n; ++i) a.push_back(std::make_tuple(TSeq(), pos[i], count[i]));
  std::sort(a.begin(), a.end(), [cc](const anchorTuple& lhs, const anchorTuple& rhs) {
    return abs((int)cc - (int)std::get<2>(lhs)) < abs((int)cc - (int)std::get<2>(rhs));
  });
  for (int i = 0; i < n; ++i) { pos[i] = std::get<1>(a[i]); count[i] = std::get<2>(a[i]); }
}
double orc_seq_error_threshold(const orc_params* p, const uint32_t* counts, uint64_t n) {
  std::vector<colouredCount> c(n);
  for (uint64_t i = 0; i < n; ++i) c[i] = colouredCount(counts[i], 0);
  return computeSeqErrorThreshold(toParams(p), c);
}

}  // extern "C"
