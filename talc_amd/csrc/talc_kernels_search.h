// talc_kernels_search.h — structure + path-search + reassembly kernels.
//
// Device restatement of the reference's per-read correction:
//   k_structure : Read::defineStructure2           Read.cpp:260-276, 440-600, 214-258
//   k_search    : Read::correct2                   Read.cpp:336-386
//                 Explorer::searchBridge/searchEdge Explorer.cpp:868-1081 and everything below them
//                 (anchors :413-543, oneMoreStep :546-612, oneMoreStepInTheDark :615-687,
//                  scoreBridges :689-706, scoreEdges :709-740, gardening :773-865, Trail.cpp,
//                  Trajectory.cpp)
//   k_pack      : Read::updateCorrSeq / getCorrSeq + the -rev undo of main.cpp:286
//
// One wavefront per read (see talc_wave.h).  Candidate paths ("Trails") are flat byte strings
// in growth order in per-wave HBM scratch, in two ping-pong sets of TCAP slots.
#pragma once
#include "talc_common.h"
#include "talc_kernels_probe.h"
#include "talc_pure.h"
#include "talc_wave.h"

namespace talc {

#define TCAP 288          /* slots per Trail set: <= 4*max_inner_paths children, or kept + MAXB */
#define NBUF (2 * TCAP)    /* sequence buffers shared by the two Trail sets (9 x 64) */
#ifndef LDS_DP_CAP
#define LDS_DP_CAP 320    /* ints per DP array held in LDS; longer problems use the HBM arrays */
#endif
#ifndef HOT
#define HOT 8             /* Trail slots per set whose metadata lives in LDS (the common case has 1-8 Trails) */
#endif
#ifndef AIMS_LDS
#define AIMS_LDS 16       /* target anchors kept in LDS for the aim check (a region side rarely has more than 3) */
#endif

enum : uint32_t {
  OVF_ANCHORS = 1, OVF_FULLPATHS = 2, OVF_FULLPOOL = 4, OVF_TRAILS = 8, OVF_SEQ = 16, OVF_OUT = 32, OVF_WEAKPOOL = 64,
  OVF_REGIONS = 128, OVF_DP = 256
};

struct ReadState {
  int32_t status;       // talc_read_status (TALC_READ_CORRECTED == structure defined, to be searched)
  uint32_t nRegions;
  double lambda;        // m_priorLambda_noise (Read.cpp:268-269)
  uint32_t outLen;      // corrected length in codes (growth of the read's out slot)
  uint32_t overflow;    // OVF_* bits: scratch exhausted, read left unchanged
  uint32_t inSpan;      // sum over the IN regions of (end - start + 1), as they stand (Read.cpp:423: the stats row)
  uint32_t costEst;     // k_structure's estimate of the search's work (orders the work queue: heaviest reads first)
  uint32_t costGap;     // ... the part of it that stands for the inner gaps (the queue's key weighs it by the batch's fork share)
#ifdef TALC_PROF
  // profile build only (TALC_PROF_READS=file writes one row per read): the estimate's inputs and what the search took
  uint32_t pfHead, pfTail, pfGapSum, pfFork, pfSolid, pfTicks;   // pfTicks: 100 MHz
  uint32_t pfGapSq, pfGapMax, pfShortReg, pfSteps;
  uint32_t pfBridges, pfBridgeMax;   // start anchors of bridge searches walked, the longest walk (100 MHz)
  uint32_t pfEdgeTicks, pfAnchors, pfAnchorMax, pfStart;   // (pfStart: low 32 bits of the 100 MHz counter when the search took the read)   // 100 MHz ticks inside search_edge; edge anchors searched by this wave, the longest
#endif
};

__host__ __device__ inline uint64_t out_capacity_for(uint64_t L) { return 4 * L + 1024; }

struct FullMeta { uint32_t off, len; int32_t lanc, ranc; double dist; };

// the work queue's counters (uint32 words; 1 KB apart: the waves without a read poll the last two while the others take reads from the first)
static const uint32_t kQueueWords = 768, kQueueFinished = 256, kQueueOpen = 512;

// per-wave scratch layout (byte offsets inside one slot)
struct SearchLimits {   // (the part the search itself consults: a copy lives in the wave's LDS)
  uint32_t seqCap;      // bytes per Trail sequence (host: the longest possible; device: the current search's stride)
  uint32_t seqArena;    // bytes of the arena the Trail buffers are cut from
  uint32_t refCap;      // bytes of currentRefSeq
  uint32_t edgeCap;     // bytes per edge candidate sequence
  uint32_t anchCap;     // anchors per side
  uint32_t fullCap;     // recorded bridges per start anchor
  uint32_t fullPool;    // bytes for their sequences
  uint32_t dpCap;       // ints per HBM DP array
  uint32_t regCap;      // regions per read
  uint32_t weakPool;    // bytes for corrected weak sequences
};
struct SearchCaps : SearchLimits {
  uint64_t slotBytes;
  uint64_t o_setA, o_setB, o_seqPool, o_ref, o_ancL, o_ancR, o_ancPos, o_fullMeta, o_fullPoolB, o_edgeLong,
      o_edgeShort, o_edgeTmp, o_dp, o_gard, o_regS, o_regE, o_wOff, o_wLen, o_weak, o_wideBloom, o_rowPool, o_regH, o_trace;
};

// The cycle filter of a LONG search (a gap of several kb: a Trail of thousands of k-mers saturates the 8192 bits the wave
// has in LDS, and every false alarm costs an exact window search over the whole path): up to 2^20 bits per wave in HBM,
// sized per search at ~32 bits per possible k-mer, read and written by agent-scope atomics (the L2 is their coherence point).
// scoreBridges' alignments continued row by row (wave_nw_rows): the last matrix row of every Trail, over the whole
// reference.  One arena per wave, cut per search into records of refLen + 2 ints (as many as fit, at most one per Trail
// buffer; a Trail whose buffer has no record, and every reference beyond ROW_MAX_REF bases, is aligned from scratch).
// A record's word 0 (column 0 of the matrix is zero by definition) says how many rows (= Trail bases) the record covers
// and in which search of the wave, its last word in which launch (k_search's launchStamp): nothing is reset per search,
// and neither an earlier launch's records nor — the slots move when the batch's longest read changes — another wave's
// are ever taken for this search's.
#ifndef ROW_ARENA_INTS
#define ROW_ARENA_INTS (448 * 1024)
#endif
#define ROW_MAX_REF 8191
#define ROW_COV_BITS 13   /* a record's word 0: rows covered (a Trail of up to 8191 bases) below the search's number */
#define WIDE_BLOOM_WORDS 16384
#ifndef WIDE_BLOOM_MIN_PATH
#define WIDE_BLOOM_MIN_PATH 600   /* Trails that may grow beyond this many bases use it (config 5: search 122 / 113 / 107 / 104 / 103 / 104 ms at 2500 / 1500 / 1000 / 700 / 450 / 300) */
#endif

static inline uint64_t align_up(uint64_t x, uint64_t a) { return (x + a - 1) / a * a; }

// metadata of one Trail set: SoA, TCAP entries each
struct TrailSetLayout {
  static constexpr uint64_t kmer = 0;
  static constexpr uint64_t nmask = kmer + 8ull * TCAP;
  static constexpr uint64_t dist = nmask + 8ull * TCAP;
  static constexpr uint64_t cnt = dist + 8ull * TCAP;
  static constexpr uint64_t score = cnt + 4ull * TCAP;
  static constexpr uint64_t fail = score + 4ull * TCAP;
  static constexpr uint64_t lanc = fail + 4ull * TCAP;
  static constexpr uint64_t ranc = lanc + 4ull * TCAP;
  static constexpr uint64_t buf = ranc + 4ull * TCAP;
  static constexpr uint64_t bytes = buf + 4ull * TCAP;
};

// Scratch of one wave.  A Trail of a search is at most K + PATH_MAXLENGTH = 1.2 gap + 4 K bases long
// (Explorer.cpp:919,1036), known when the search starts: the Trail buffers of a search are cut from one arena with
// that stride (X.C.seqCap is set per search), so a short gap gets the full NBUF buffers and a 20 kb gap still a few
// dozen — nearly every gap of a noisy read is a few hundred bases whatever the read's length, and sizing all NBUF
// buffers for the longest read would cost 14 MB per wave at 20 kb.  A search that needs more buffers than its stride
// leaves (many live Trails in a very long gap) raises OVF_TRAILS / OVF_SEQ and the read goes to the retry passes,
// whose arena holds NBUF buffers of the longest possible Trail and whose counted capacities (anchors, recorded
// bridges) are multiplied by `scale`.  arenaLimit (first pass only; 0 = the default) is a test hook.
static inline SearchCaps make_caps(uint32_t maxLen, uint32_t K, uint32_t scale, uint32_t arenaLimit, bool tiny = false) {
  SearchCaps c = SearchCaps();
  const uint64_t Lm = maxLen;
  c.seqCap = (uint32_t)align_up((uint64_t)(1.2 * (double)Lm) + 4ull * K + 64, 16);   // the longest Trail of any search
  const uint64_t fullArena = (uint64_t)NBUF * c.seqCap;
  uint64_t arena = fullArena;
  if (scale == 1) {
    arena = std::min<uint64_t>(fullArena, std::max<uint64_t>(1ull << 20, 8ull * c.seqCap));
    if (arenaLimit) arena = std::min<uint64_t>(arena, arenaLimit);
  }
  c.seqArena = (uint32_t)align_up(std::min<uint64_t>(arena, 0xFFFFFF00ull), 16);
  c.refCap = (uint32_t)align_up(Lm + 2ull * K + 64, 16);
  c.edgeCap = (uint32_t)align_up((uint64_t)c.seqCap + c.refCap, 16);
  // (a region of n k-mers records at most n positions: beyond that the anchor lists cannot overflow.  2048 in the first
  //  pass already — 100 KB of a 3.4 MB slot: a nearly error-free read is ONE region of a kilobase and more, its lists hold
  //  every forking k-mer of it, and with 256 every such read went to the retry stage's few slots)
  c.anchCap = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(256ull * scale, 2048ull), Lm + 8);
  if (c.anchCap < 8) c.anchCap = 8;
  c.fullCap = 128 * scale;
  c.fullPool = (uint32_t)align_up(std::max<uint64_t>(65536, 4ull * c.seqCap) * scale, 16);
  c.dpCap = (uint32_t)align_up(std::max<uint64_t>(c.edgeCap, c.seqCap) + 8, 4);
  if (c.dpCap < 2048) c.dpCap = 2048;   // (the phased x-drop keeps its hand-over state in these arrays)
  if (tiny) { c.anchCap = 3; c.fullCap = 1; c.fullPool = (uint32_t)align_up((uint64_t)c.seqCap, 16); }  // test hook: force the retry pass
  c.regCap = (uint32_t)(Lm / 2 + 4);
  c.weakPool = (uint32_t)align_up(out_capacity_for(Lm), 16);
  uint64_t o = 0;
  auto take = [&](uint64_t bytes) { uint64_t r = o; o = align_up(o + bytes, 16); return r; };
  c.o_setA = take(TrailSetLayout::bytes);
  c.o_setB = take(TrailSetLayout::bytes);
  c.o_seqPool = take(c.seqArena);
  c.o_ref = take(c.refCap);
  c.o_ancL = take((uint64_t)c.anchCap * sizeof(AnchorRec));
  c.o_ancR = take((uint64_t)c.anchCap * sizeof(AnchorRec));
  c.o_ancPos = take((uint64_t)c.anchCap * 4);
  c.o_fullMeta = take((uint64_t)c.fullCap * sizeof(FullMeta));
  c.o_fullPoolB = take(c.fullPool);
  c.o_edgeLong = take(c.edgeCap);
  c.o_edgeShort = take(c.edgeCap);
  c.o_edgeTmp = take(c.edgeCap);
  c.o_dp = take(3ull * c.dpCap * 4);
  c.o_gard = take((uint64_t)(TCAP + 64) * (8 + 8 + 16 + 32 + 4));
  c.o_regS = take((uint64_t)c.regCap * 4);
  c.o_regE = take((uint64_t)c.regCap * 4);
  c.o_wOff = take((uint64_t)c.regCap * 4);
  c.o_wLen = take((uint64_t)c.regCap * 4);
  c.o_weak = take(c.weakPool);
  c.o_wideBloom = take((uint64_t)WIDE_BLOOM_WORDS * 8);
  c.o_rowPool = take((uint64_t)ROW_ARENA_INTS * 4);
  c.o_regH = take((uint64_t)c.regCap * 4);
  c.o_trace = take(64);
  c.slotBytes = align_up(o, 256);
  return c;
}

#if defined(__HIPCC__)

// ------------------------------------------------------------------ trace (debug hook)
struct TraceRec { int32_t kind; int32_t a, b, c, d; double x; uint32_t soff, slen; };
struct TraceBuf { TraceRec* recs; uint32_t* nrec; uint32_t cap; uint8_t* pool; uint32_t* npool; uint32_t poolCap; int steps; };
enum { TR_REGION = 1, TR_THRESHOLD = 2, TR_SEARCH = 3, TR_ANCHOR = 4, TR_RESULT = 5, TR_STEP = 6 };

// wave_copy_bytes as a real call: two dozen inlined copies of its loop were a quarter of k_search's instructions, and
// every one of them waits on a global round trip anyway (a leaf with a handful of registers: nothing to save)
TALC_DN void copy_bytes(uint8_t* dst, const uint8_t* src, uint32_t n, bool rev) {
  wave_copy_bytes(uni_ptr(dst), uni_ptr(src), (uint32_t)uni((int)n), uni((int)rev) != 0);
}

// ------------------------------------------------------------------ small wave helpers on reads
// packed natural-orientation k-mer (+ N mask) of s[0..K): lanes 0..K-1 load one base each
TALC_D void wave_kmer_at(const uint8_t* __restrict__ s, int K, uint64_t& kmer, uint64_t& nmask) {
  const int l = lane_id();
  uint32_t c = (l < K) ? (uint32_t)((gcu8)s)[l] : 0u;
  nmask = ballot64((l < K) && (c > 3u));
  uint64_t v = (l < K) ? ((uint64_t)(c & 3u) << (2 * (K - 1 - l))) : 0ull;
  if (c > 3u) v = 0;
  v = ((uint64_t)wave_or_u32((uint32_t)(v >> 32)) << 32) | wave_or_u32((uint32_t)v);
  kmer = v;
}

// the same for one position per lane (K byte loads each).  A rare path (k_coverage leaves the degrees of every table
// k-mer next to its count), so it is a real function: its registers do not weigh on its callers.
TALC_DN void lane_kmer_at(const uint8_t* __restrict__ s, int K, uint64_t& kmer, uint64_t& nmask) {
  gcu8 g = (gcu8)s;
  uint64_t v = 0, nm = 0;
  // eight byte loads in flight at a time (K <= 31; the index is clamped, never out of the read)
#pragma unroll
  for (int j0 = 0; j0 < 32; j0 += 8) {
    uint32_t c[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) c[j] = (uint32_t)g[(j0 + j) < K ? (j0 + j) : K - 1];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (j0 + j < K) {
        v = (v << 2) | (uint64_t)((c[j] > 3u) ? 0u : c[j]);
        nm |= (uint64_t)(c[j] > 3u) << (j0 + j);
      }
    }
  }
  kmer = v; nmask = nm;
}

// getOutDegree (Jellyfish.cpp:383-393) of a packed k-mer: one bucket probe (per lane, or uniform)
TALC_D int dev_out_degree(const TableView& T, uint32_t MINC, uint64_t kmer, uint64_t nmask, int dirRight) {
  const uint32_t K = T.k;
  const uint64_t succN = dirRight ? (nmask >> 1) : (nmask & ((1ULL << (K - 1)) - 1));
  if (succN) return 0;  // every successor contains an N: absent from the table
  uint32_t c[4], j[4];
  dev_next_counts(T, kmer, dirRight, c, j);
  int d = 0;
#pragma unroll
  for (int b = 0; b < 4; ++b) d += (c[b] >= MINC) ? 1 : 0;
  return d;
}

// ------------------------------------------------------------------ the coverage of one read (talc_common.h: CovWord)
struct CovRead {
  const uint2 TALC_AS1* hits;       // the read's dense slot: the hits of tile t packed from index t * TALC_COV_TILE
  const CovWord TALC_AS1* words;    // one word per 64 positions
};
TALC_D CovRead cov_read(const uint2* covAll, const CovWord* wordsAll, uint64_t koff, uint32_t r) {
  CovRead c;
  c.hits = (const uint2 TALC_AS1*)(covAll + koff);
  c.words = (const CovWord TALC_AS1*)(wordsAll + cov_word_base(koff, r));
  return c;
}
TALC_D CovWord cov_word(const CovRead& cr, uint32_t w) {
  const v4u32 q = *(const v4u32 TALC_AS1*)(cr.words + w);
  CovWord cw; cw.bits = ((uint64_t)q.y << 32) | q.x; cw.rank = q.z; cw.pad = 0;
  return cw;
}
// {count, colour | degrees} of the k-mer at position p (per lane or uniform): (0, 0) when it is not in the table
TALC_D uint2 cov_at(const CovRead& cr, uint32_t p) {
  const CovWord cw = cov_word(cr, p >> 6);
  uint2 v = make_uint2(0u, 0u);
  if ((cw.bits >> (p & 63u)) & 1ull) {
    const uint32_t idx = (p & ~(uint32_t)(TALC_COV_TILE - 1)) + cw.rank + (uint32_t)__popcll(cw.bits & ((1ull << (p & 63u)) - 1ull));
    const v2u32 e = *(const v2u32 TALC_AS1*)(cr.hits + idx);
    v.x = e.x; v.y = e.y;
  }
  return v;
}
static const uint32_t kRegClean = 1u << 31;   // flag in a region's hit index (k_structure)
static const int kHeadCov = 16;               // per read: dense counts of its first positions (k_structure -> build_anchors)
// index (in the read's hit array) of the pair of position p, which must be a hit
TALC_D uint32_t cov_hit_index(const CovRead& cr, uint32_t p) {
  const CovWord cw = cov_word(cr, p >> 6);
  return (p & ~(uint32_t)(TALC_COV_TILE - 1)) + cw.rank + (uint32_t)__popcll(cw.bits & ((1ull << (p & 63u)) - 1ull));
}
// ... and of a position p >= rs inside a RUN of hits that starts at rs (a solid region: every position of it is a hit),
// rsIdx = cov_hit_index(rs): the pairs of a run are consecutive inside a tile, and a tile the run entered from the left
// holds the run's pairs from its first slot — no word needs reading
TALC_D uint32_t cov_run_index(uint32_t rs, uint32_t rsIdx, uint32_t p) {
  return ((p ^ rs) >= (uint32_t)TALC_COV_TILE) ? p : rsIdx + (p - rs);
}
// f(count, colourWord) for every hit of the read, lane-strided.  The tiles' hit counts (rank + population of each tile's
// last word) are fetched 64 tiles at a time, one per lane: one memory round trip for a read of up to 32 k positions, then
// the tiles' pairs stream.
template <class F>
TALC_D void cov_for_hits(const CovRead& cr, uint32_t n, F&& f) {
  const int l = lane_id();
  const uint32_t nTiles = (n + (uint32_t)TALC_COV_TILE - 1u) / (uint32_t)TALC_COV_TILE;
  for (uint32_t tb = 0; tb < nTiles; tb += 64) {
    const uint32_t t = tb + (uint32_t)l;
    uint32_t nhMine = 0;
    if (t < nTiles) {
      const uint32_t lastW = (min(n, (t + 1u) * (uint32_t)TALC_COV_TILE) - 1u) >> 6;
      const CovWord cw = cov_word(cr, lastW);
      nhMine = cw.rank + (uint32_t)__popcll(cw.bits);
    }
    const uint32_t cnt = min(64u, nTiles - tb);
    for (uint32_t j = 0; j < cnt; ++j) {
      const uint32_t nh = (uint32_t)lane_get((int)nhMine, (int)j), t0 = (tb + j) * (uint32_t)TALC_COV_TILE;
      for (uint32_t i = (uint32_t)l; i < nh; i += 64) { const v2u32 e = *(const v2u32 TALC_AS1*)(cr.hits + t0 + i); f(e.x, e.y); }
    }
  }
}

// ==================================================================== k_structure
// Read::defineStructure2 for one read per wave.
// S(r) = sum of the r smallest IN counts (order statistics without sorting: the reference's sort only
// feeds a trimmed sum, Read.cpp:505-512).  Counts below STRUCT_HBINS go through an LDS histogram (one
// pass over the coverage, then a scan over the bins); larger ones through a bisection on the value.
constexpr int STRUCT_HBINS = 1024, STRUCT_DEG_CAP = 512;   // (5 KB of LDS per wave: 32 waves per CU)
TALC_D unsigned long long trimmed_prefix_sum(const CovRead& cov, uint32_t n, uint32_t MINC, uint32_t r, uint32_t vmax) {
  if (r == 0) return 0ull;
  // smallest v with #{x >= MINC, x <= v} >= r
  uint32_t lo = 0, hi = vmax;
  while (lo < hi) {
    const uint32_t mid = lo + (hi - lo) / 2;
    unsigned long long cnt = 0;
    cov_for_hits(cov, n, [&](uint32_t x, uint32_t) { cnt += (x >= MINC && x <= mid) ? 1 : 0; });
    cnt = wave_sum_u64(cnt);
    if (cnt >= r) hi = mid; else lo = mid + 1;
  }
  const uint32_t v = lo;
  unsigned long long sumLess = 0, cntLess = 0;
  cov_for_hits(cov, n, [&](uint32_t x, uint32_t) { if (x >= MINC && x < v) { sumLess += x; cntLess += 1; } });
  sumLess = wave_sum_u64(sumLess);
  cntLess = wave_sum_u64(cntLess);
  return sumLess + ((unsigned long long)r - cntLess) * (unsigned long long)v;
}
// the same from the histogram: lane l owns bins [32 l, 32 l + 32); cBefore / sBefore = count and sum of the bins of
// the lanes before it
TALC_D unsigned long long hist_prefix_sum(const uint32_t* hist, unsigned long long cBefore, unsigned long long sBefore,
                                          unsigned long long cMine, uint32_t r) {
  if (r == 0) return 0ull;
  const int l = lane_id();
  const bool owner = (cBefore < (unsigned long long)r) && ((unsigned long long)r <= cBefore + cMine);
  unsigned long long res = 0;
  if (owner) {
    unsigned long long c = cBefore, sm = sBefore;
    for (int b = 0; b < STRUCT_HBINS / 64; ++b) {
      const uint32_t v = (uint32_t)(l * (STRUCT_HBINS / 64) + b), h = hist[v];
      if (c + h >= (unsigned long long)r) { res = sm + ((unsigned long long)r - c) * v; break; }
      c += h; sm += (unsigned long long)h * v;
    }
  }
  const unsigned long long m = ballot64(owner);
  const int src = m ? (int)__builtin_ctzll(m) : 0;
  return ((unsigned long long)(uint32_t)lane_get((int)(uint32_t)(res >> 32), src) << 32) | (uint32_t)lane_get((int)(uint32_t)res, src);
}

__global__ void __launch_bounds__(64)
k_structure(DevParams P, TableView T, const uint8_t* __restrict__ codes, const uint64_t* __restrict__ offsets,
            const uint64_t* __restrict__ koff, const uint2* __restrict__ covAll, const CovWord* __restrict__ covWords,
            const int32_t* __restrict__ n_in,
            ReadState* __restrict__ state, uint32_t* __restrict__ regions, const uint64_t* __restrict__ regoff,
            uint32_t* __restrict__ headCov, uint32_t n_reads, TraceBuf trace, uint32_t traceRead, uint32_t* __restrict__ batchStats) {
  const uint32_t r = blockIdx.x;
  if (r >= n_reads) return;
  const int l = lane_id();
  const uint32_t K = P.K, MINC = P.MIN_COUNT;
  const uint8_t* read = codes + offsets[r];
  const uint32_t L = (uint32_t)(offsets[r + 1] - offsets[r]);
  ReadState st;
  st.status = TALC_READ_CORRECTED; st.nRegions = 0; st.lambda = (double)MINC; st.outLen = 0; st.overflow = 0; st.inSpan = 0; st.costEst = 0; st.costGap = 0;
  if (!(L > K)) { st.status = TALC_READ_SKIPPED_SHORT; if (l == 0) state[r] = st; return; }      // main.cpp:262
  if (!(n_in[r] > 0)) { st.status = TALC_READ_NO_SOLID_KMER; if (l == 0) state[r] = st; return; }  // Read.cpp:194
  const uint32_t n = L - K + 1;
  const CovRead cov = cov_read(covAll, covWords, koff[r], r);
  uint32_t* regS = regions + 3 * regoff[r];   // per read: starts, ends, hit indices of the starts (kRegClean: below)
  const uint32_t regCap = (uint32_t)(regoff[r + 1] - regoff[r]);
  uint32_t* regE = regS + regCap;

  // ---- findINRegions (Read.cpp:440-489): maximal runs of count >= MIN_COUNT (needs n > 1)
  uint32_t R = 0, Rends = 0;
  if (n > 1) {
    // (a k-mer is IN iff it is in the table: every stored count is >= MIN_COUNT, talc_common.h — so the runs are the runs
    //  of the hit bitmap, 64 positions per word, the words of one pass fetched one per lane)
    const uint32_t nW = (n + 63u) >> 6;
    unsigned long long prevBits = 0ull;
    for (uint32_t base = 0; base < n; base += 64) {
      const uint32_t i = base + l;
      const uint32_t w = base >> 6;
      unsigned long long bitsW, nextW = 0ull;
      {
        // lanes 0 / 1 fetch this word and the next; both are made uniform
        const uint32_t wi = w + (uint32_t)(l & 1);
        const unsigned long long mine = (l < 2 && wi < nW) ? cov_word(cov, wi).bits : 0ull;
        bitsW = ((unsigned long long)(uint32_t)lane_get((int)(uint32_t)(mine >> 32), 0) << 32) | (uint32_t)lane_get((int)(uint32_t)mine, 0);
        nextW = ((unsigned long long)(uint32_t)lane_get((int)(uint32_t)(mine >> 32), 1) << 32) | (uint32_t)lane_get((int)(uint32_t)mine, 1);
      }
      const unsigned long long ms = bitsW & ~((bitsW << 1) | (prevBits >> 63));
      const unsigned long long me = bitsW & ~((bitsW >> 1) | (nextW << 63));
      prevBits = bitsW;
      const bool isStart = ((ms >> l) & 1ull) != 0ull, isEnd = ((me >> l) & 1ull) != 0ull;
      // the k-th start pairs with the k-th end
      const unsigned long long below = (l == 0) ? 0ull : (~0ull >> (64 - l));
      if (isStart) { const uint32_t k = R + (uint32_t)__popcll(ms & below); if (k < regCap) regS[k] = i; }
      if (isEnd) { const uint32_t k = Rends + (uint32_t)__popcll(me & below); if (k < regCap) regE[k] = i; }
      R += (uint32_t)__popcll(ms);
      Rends += (uint32_t)__popcll(me);
    }
  }
  if (R > regCap) { st.overflow |= OVF_REGIONS; R = regCap; }
  WSYNC();

  // ---- computeSeqErrorThreshold (Read.cpp:493-518)
  __shared__ uint32_t s_hist[STRUCT_HBINS];
  __shared__ uint8_t s_degS[STRUCT_DEG_CAP], s_degE[STRUCT_DEG_CAP];
  unsigned long long m = 0;
  uint32_t vmax = 0;
  // one pass over the hits: their number, their largest count and — optimistically — the histogram of the counts below
  // STRUCT_HBINS (used when vmax turns out to be below it, which it nearly always is)
  for (int b = l; b < STRUCT_HBINS; b += 64) s_hist[b] = 0;
  WSYNC();
  unsigned long long nFork = 0;   // solid k-mers with more than one successor or predecessor in the graph (the cost estimate)
  cov_for_hits(cov, n, [&](uint32_t x, uint32_t y) {
    if (x >= MINC) { m += 1; vmax = max(vmax, x); if (x < (uint32_t)STRUCT_HBINS) atomicAdd(&s_hist[x], 1u); }
    nFork += ((x >= MINC) && (y & kCovDegKnown) && ((((y >> kCovDegRShift) & 7u) > 1u) || (((y >> kCovDegLShift) & 7u) > 1u))) ? 1u : 0u;
  });
  m = wave_sum_u64(m);
  nFork = wave_sum_u64(nFork);
  vmax = wave_max_u32(vmax);
  uint32_t first = 0, last = (uint32_t)m;
  if (m > 10) { first = (uint32_t)(0.15 * (double)m); last = (uint32_t)(0.90 * (double)m); }
  unsigned long long sum;
  if (vmax < (uint32_t)STRUCT_HBINS) {
    WSYNC();
    unsigned long long cMine = 0, sMine = 0;
    for (int b = 0; b < STRUCT_HBINS / 64; ++b) { const uint32_t v = (uint32_t)(l * (STRUCT_HBINS / 64) + b), h = s_hist[v]; cMine += h; sMine += (unsigned long long)h * v; }
    unsigned long long cIncl = cMine, sIncl = sMine;   // inclusive scan over the lanes
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned long long c2 = (unsigned long long)__shfl_up((long long)cIncl, off, 64), s2 = (unsigned long long)__shfl_up((long long)sIncl, off, 64);
      if (l >= off) { cIncl += c2; sIncl += s2; }
    }
    sum = hist_prefix_sum(s_hist, cIncl - cMine, sIncl - sMine, cMine, last) - hist_prefix_sum(s_hist, cIncl - cMine, sIncl - sMine, cMine, first);
  } else {
    sum = trimmed_prefix_sum(cov, n, MINC, last, vmax) - trimmed_prefix_sum(cov, n, MINC, first, vmax);
  }
  double robMean = (double)((unsigned long long)MINC + sum);
  robMean /= (double)(last - first);
  const double thr = robMean * P.ERR;
  st.lambda = thr;

  // ---- out-degrees of every region's first k-mer (towards LEFT) and last k-mer (towards RIGHT), one region per
  // lane: the walk below needs them one after the other, and each is a dependent probe
  const uint32_t Rdeg = min(R, (uint32_t)STRUCT_DEG_CAP);
  for (uint32_t base = 0; base < Rdeg; base += 64) {
    const uint32_t reg = base + l;
    if (reg < Rdeg) {   // region ends are IN k-mers, i.e. k-mers of the table: k_coverage left their degrees in cov[].y
      const uint32_t ys = cov_at(cov, regS[reg]).y, ye = cov_at(cov, regE[reg]).y;
      uint64_t km, nm;
      if (ys & kCovDegKnown) s_degS[reg] = (uint8_t)((ys >> kCovDegLShift) & 7u);
      else { lane_kmer_at(read + regS[reg], (int)K, km, nm); s_degS[reg] = (uint8_t)dev_out_degree(T, MINC, km, nm, 0); }
      if (ye & kCovDegKnown) s_degE[reg] = (uint8_t)((ye >> kCovDegRShift) & 7u);
      else { lane_kmer_at(read + regE[reg], (int)K, km, nm); s_degE[reg] = (uint8_t)dev_out_degree(T, MINC, km, nm, 1); }
    }
  }
  WSYNC();

  // out-degree of the k-mer at a wave-uniform position: from the coverage word when k_coverage knew the k-mer
  auto degree_of = [&](uint32_t pos, int dirRight) -> int {
    const uint32_t y = (uint32_t)uni((int)cov_at(cov, pos).y);
    if (y & kCovDegKnown) return (int)((y >> (dirRight ? kCovDegRShift : kCovDegLShift)) & 7u);
    uint64_t km, nm;
    wave_kmer_at(read + pos, (int)K, km, nm);
    return dev_out_degree(T, MINC, km, nm, dirRight);
  };

  // ---- analyzeINRegions (Read.cpp:524-600): uniform serial walk over the regions
  // new list is written in place behind a write cursor (nNew <= reg always)
  uint32_t nNew = 0;
  {
    const uint32_t solidThr = (uint32_t)thr;
    bool startMoved = false;
    for (uint32_t reg = 0; reg < R; ++reg) {
      int span = 0;
      bool OK = true;
      uint32_t new_start_pos = regS[reg];
      const uint32_t regEnd = regE[reg];
      const bool lastAndNone = ((R == reg + 1) & (nNew == 0));
      int degStart;
      if (reg < Rdeg && !startMoved) degStart = (int)s_degS[reg];
      else degStart = degree_of(new_start_pos, 0);
      startMoved = false;
      if (!lastAndNone & (degStart == 0) & (new_start_pos != 0)) {
        OK = false;
        while ((new_start_pos < regEnd) & !OK) {
          ++new_start_pos;
          if (degree_of(new_start_pos, 0) > 1) OK = true;
        }
      }
      uint32_t new_end_pos = regEnd;
      if (OK & !lastAndNone) {
        int degEnd;
        if (reg < Rdeg) degEnd = (int)s_degE[reg];
        else degEnd = degree_of(new_end_pos, 1);
        if ((degEnd == 0) & (new_end_pos != n - 1)) {
          OK = false;
          while ((new_end_pos > regS[reg]) & !OK) {
            --new_end_pos;
            if (degree_of(new_end_pos, 1) > 1) OK = true;
          }
        }
      }
      if (OK) {
        if (reg + 1 < R) span = ((int)regS[reg + 1] - (int)(new_end_pos + K));
        if (span < 0) {
          if ((int)regE[reg + 1] + span >= (int)regS[reg + 1]) { if (l == 0) regS[reg + 1] = regS[reg + 1] - (uint32_t)span; }
          else { if (l == 0) regS[reg + 1] = new_start_pos; OK = false; }
          startMoved = true;   // the next region no longer starts where its degree was taken
          WSYNC();
        }
        if (OK) {
          uint32_t c = 0;
          for (uint32_t i = new_start_pos + l; i <= new_end_pos; i += 64) c = max(c, cov_at(cov, i).x);
          c = wave_max_u32(c);
          const bool expected = !is_expected_by_model(P.ALPHA, c, solidThr, true);
          if (expected) {
            // kept regions are compacted to the front; slot nNew <= reg has already been consumed
            WSYNC();
            if (l == 0) { regS[nNew] = new_start_pos; regE[nNew] = new_end_pos; }
            WSYNC();
            ++nNew;
          }
        }
      }
    }
  }
  // Read.cpp:599: the original list is kept when nothing qualified.  The in-place compaction above
  // destroys it only if nNew > 0, but the span fix-ups (:582,:585) already edited the original list
  // in place in the reference too, so when nNew == 0 regS/regE hold exactly what the reference keeps.
  uint32_t Rfinal = (nNew > 0) ? nNew : R;
  bool checok = (R > 0);   // findINRegions' return value (Read.cpp:488)

  // ---- setInitialStructure (Read.cpp:214-258): only its length check decides anything here
  if (Rfinal > 0) {
    unsigned long long len = 0;
    if (regS[0] > 0) len += regS[0];
    const uint32_t eLast = regE[Rfinal - 1];
    if (eLast + 1 < n) len += (unsigned long long)L - ((unsigned long long)eLast + K);
    unsigned long long part = 0;
    for (uint32_t i = l; i + 1 < Rfinal; i += 64) {
      part += (unsigned long long)regE[i] + K - regS[i];
      if (regS[i + 1] > regE[i] + K) part += (unsigned long long)regS[i + 1] - ((unsigned long long)regE[i] + K);
    }
    len += wave_sum_u64(part);
    len += (unsigned long long)eLast + K - regS[Rfinal - 1];
    checok &= (len == (unsigned long long)L);
  }
  st.nRegions = Rfinal;
  // the counts of the read's first kHeadCov positions, dense (one 64-byte line per read): the count the reference records
  // with anchor number a of a region is m_coverage[a] — a position of the READ (Explorer.cpp:454,520) — and every anchor
  // list of the read asks for it
  if (l < kHeadCov) headCov[(uint64_t)r * kHeadCov + (uint32_t)l] = ((uint32_t)l < n) ? cov_at(cov, (uint32_t)l).x : 0u;
  {   // where each region's pairs start in the read's hit array, for the anchor search (k_search): the index of the
      // region's first pair, and kRegClean when every position of the region is a hit of one tile, i.e. when the pair of
      // position p is simply at index + (p - start) — nearly always; a region merged over a gap or one that crosses a tile
      // is addressed position by position through the bitmap words
    WSYNC();
    uint32_t* regH = regS + 2 * regCap;
    for (uint32_t i = l; i < Rfinal; i += 64) {
      const uint32_t s0 = regS[i], e0 = regE[i];
      const uint32_t hs = cov_hit_index(cov, s0);
      // (the end itself must be a hit too: cov_hit_index of a position that is none counts the hits BELOW it, and a region
      //  whose last position is no hit — defineStructure2 leaves such ends — passed the difference test alone; its pivot's
      //  count was then read from the NEXT hit's pair: one read in 7500 of a branching transcriptome, tools/stress_branching.py)
      const bool endHit = ((cov_word(cov, e0 >> 6).bits >> (e0 & 63u)) & 1ull) != 0ull;
      const bool clean = ((s0 ^ e0) < (uint32_t)TALC_COV_TILE) && (e0 >= s0) && endHit && (cov_hit_index(cov, e0) - hs == e0 - s0);
      regH[i] = (hs & ~kRegClean) | (clean ? kRegClean : 0u);
    }
  }
  {   // Read.cpp:423 on the regions as defineStructure2 leaves them (k_search redoes it for the reads it corrects)
    unsigned long long part = 0;
    for (uint32_t i = l; i < Rfinal; i += 64) part += (unsigned long long)regE[i] - regS[i] + 1;
    st.inSpan = (uint32_t)wave_sum_u64(part);
  }
  if (!checok) st.status = TALC_READ_NO_STRUCTURE;   // main.cpp:290
  if (checok && Rfinal > 0) {
    // What the path search will cost, roughly: an edge (head / tail of at most MAX_BORDER_LEN bases) is searched from up
    // to five anchors, each for 1.2 x its length steps with a seed extension every CHECK_INTERVAL steps whose x grows
    // with the path — quadratic in the edge's length, and the reason why a 1.2 kb read with a 480-base head takes ten
    // times as long as a 3 kb read without one; an inner gap is walked once and evaluated once.  Only the order of the
    // work queue depends on this number (measured: with the queue ordered by read length alone the waves were busy 63 %
    // of the launch, waiting for a few late heavy reads).
    // (in wave-cycles, from the category profile of config 2: a step of the walk ~ 460, a wavefront level ~ 1500,
    //  a level per ~ 11 bases of path at 12 % error: an inner gap of g bases ~ 870 g + 20 000, an edge of h bases
    //  ~ 5 anchors x (550 h + 27 h^2)).
    // ... plus q x min(sum of g^2, 900^2): not a cost but a RISK.  Per base a long gap is no dearer than a short one
    // (0.9 ms per kb of gap for 150-base and for 900-base gaps alike), but one read in a hundred with a 400-600 base gap
    // takes five times that — and replaying the per-read times of a launch through the queue, those were the reads that
    // finished last, started late because they looked like 1.5 ms reads.  The term starts every read with a long gap
    // early; it is bounded because beyond one 900-base gap it says nothing new (on config 5, K = 31, nearly every read
    // has one, and the unbounded sum put single-gap reads before the truly heavy many-gap ones: 88.8 -> 103 ms);
    // q = 10 + 200 x the share of the read's solid k-mers with more than one successor or predecessor: where the graph
    // forks, a long gap's walk carries several Trails (0.6 % of the solid k-mers on config 5, 1 % on config 2, 3.6 % on the
    // branching workload).  DevParams.costGap*; swept on one box in profiles/r03/cost_sweep.txt.
    unsigned long long part = 0, sq = 0;
    for (uint32_t i = l; i + 1 < Rfinal; i += 64) {
      const unsigned long long g = (regS[i + 1] > regE[i] + K) ? (unsigned long long)(regS[i + 1] - (regE[i] + K)) : 0ull;
      part += 870ull * g + 20000ull;
      sq += g * g;
    }
    const unsigned long long gq = (unsigned long long)P.costGapQuad + ((unsigned long long)P.costGapFork * nFork) / max(m, 1ull);
    unsigned long long cost = wave_sum_u64(part) + gq * min(wave_sum_u64(sq), (unsigned long long)P.costGapCap * P.costGapCap);
    st.costGap = (uint32_t)min(cost >> 6, 0xFFFFFFFFull);
    // the batch's sums of forking / solid k-mers (64 pairs of counters, one per read number mod 64; k_order_scale)
    if (l == 0) { atomicAdd(&batchStats[2 * (r & 63u)], (uint32_t)min(nFork, 0xFFFFull)); atomicAdd(&batchStats[2 * (r & 63u) + 1], (uint32_t)min(m, 0xFFFFFull)); }
    const unsigned long long head = regS[0], eLast = regE[Rfinal - 1];
    const unsigned long long tail = (eLast + 1 < n) ? (unsigned long long)L - (eLast + K) : 0ull;
    if (head > 0 && head <= P.MAX_BORDER_LEN) cost += (unsigned long long)P.costEdgeLin * head + (unsigned long long)P.costEdgeQuad * head * head;
    if (tail > 0 && tail <= P.MAX_BORDER_LEN) cost += (unsigned long long)P.costEdgeLin * tail + (unsigned long long)P.costEdgeQuad * tail * tail;
    cost += 50ull * (unsigned long long)L;
#ifdef TALC_PROF
    {
      unsigned long long gs = 0;
      for (uint32_t i = l; i + 1 < Rfinal; i += 64) gs += (regS[i + 1] > regE[i] + K) ? (unsigned long long)(regS[i + 1] - (regE[i] + K)) : 0ull;
      st.pfGapSum = (uint32_t)wave_sum_u64(gs);
      unsigned long long g2 = 0, sr = 0; uint32_t gm = 0;   // (sr: sum of min(g, 640)^2, a candidate term of the estimate)
      for (uint32_t i = l; i < Rfinal; i += 64) {
        const unsigned long long g = (i + 1 < Rfinal && regS[i + 1] > regE[i] + K) ? (unsigned long long)(regS[i + 1] - (regE[i] + K)) : 0ull;
        g2 += g * g; gm = max(gm, (uint32_t)g); sr += min(g, 640ull) * min(g, 640ull);
      }
      st.pfGapSq = (uint32_t)min(wave_sum_u64(g2), 0xFFFFFFFFull); st.pfGapMax = wave_max_u32(gm); st.pfShortReg = (uint32_t)wave_sum_u64(sr); st.pfSteps = 0;
      st.pfHead = (uint32_t)head; st.pfTail = (uint32_t)tail; st.pfFork = (uint32_t)nFork; st.pfSolid = (uint32_t)m; st.pfTicks = 0;
    }
#endif
    cost >>= 6;   // (fits 32 bits for any read)
    st.costEst = (uint32_t)min(cost, 0xFFFFFFFFull);
  }
  if (l == 0) state[r] = st;
  if (trace.recs && r == traceRead && l == 0) {
    uint32_t k = atomicAdd(trace.nrec, 1u);
    if (k < trace.cap) trace.recs[k] = TraceRec{TR_THRESHOLD, 0, 0, 0, 0, thr, 0, 0};
    for (uint32_t i = 0; i < Rfinal; ++i) {
      k = atomicAdd(trace.nrec, 1u);
      if (k < trace.cap) trace.recs[k] = TraceRec{TR_REGION, (int)regS[i], (int)regE[i], 0, 0, 0.0, 0, 0};
    }
  }
}

// ==================================================================== k_search
struct TrailSet {
  uint64_t* kmer; uint64_t* nmask; double* dist; uint32_t* cnt; int32_t* score; uint32_t* fail; int32_t* lanc;
  int32_t* ranc; uint32_t* buf;   // buf: index of the sequence buffer in the pool
};
TALC_D TrailSet make_set(uint8_t* meta) {
  TrailSet s;
  s.kmer = (uint64_t*)(meta + TrailSetLayout::kmer); s.nmask = (uint64_t*)(meta + TrailSetLayout::nmask);
  s.dist = (double*)(meta + TrailSetLayout::dist); s.cnt = (uint32_t*)(meta + TrailSetLayout::cnt);
  s.score = (int32_t*)(meta + TrailSetLayout::score); s.fail = (uint32_t*)(meta + TrailSetLayout::fail);
  s.lanc = (int32_t*)(meta + TrailSetLayout::lanc); s.ranc = (int32_t*)(meta + TrailSetLayout::ranc);
  s.buf = (uint32_t*)(meta + TrailSetLayout::buf);
  return s;
}

// one Trail's metadata (Trail.hpp:97-108 minus the sequence)
struct TrailRec { uint64_t kmer, nmask; double dist; uint32_t cnt; int32_t score; uint32_t fail; int32_t lanc, ranc; uint32_t buf; };

// LDS traffic of a wave's own lanes is executed in order: a wavefront-scope fence (no vmcnt wait)
// is enough to order lane 0's ds_write before the other lanes' ds_read.
#define LSYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

// Per-wave LDS (one wavefront per workgroup).  File-scope so that every access is a ds_* op.
__shared__ TrailRec g_hot[2 * HOT];                 // metadata of the first HOT slots of both Trail sets
__shared__ uint64_t g_aimK[AIMS_LDS], g_aimN[AIMS_LDS];
__shared__ uint32_t g_aimPos[AIMS_LDS];             // first AIMS_LDS target anchors
#define BLOOM_WORDS 128
__shared__ unsigned long long g_bloom[BLOOM_WORDS];  // 8192-bit k-mer Bloom filter of the current search (a K = 31 gap is
                                                     // ~600 k-mers: at 4096 bits one step in twelve was a false alarm)
__shared__ __attribute__((aligned(16))) int g_dp[3 * LDS_DP_CAP];                // x-drop stage / short DP arrays

struct EdgeCand {   // best candidate of m_longPaths / m_shortPaths kept online (findBestBORDER is a fold)
  bool have; double score; double dist; double idscore; uint32_t len; uint32_t lanc, ranc;
};

enum { PF_PROBE = 0, PF_CHILD, PF_AIMS, PF_CYCLE, PF_FFWD, PF_SCOREBR, PF_GARDEN, PF_EVALFULL, PF_XDROP, PF_EXTNW,
       PF_EDGEMISC, PF_ANCHORS, PF_ASSEMBLE, PF_STEPB, PF_STEPE, PF_SRCHB, PF_SRCHE, PF_PROLOG, PF_INITTR, PF_TOTAL, PF_NCALLS, PF_NSTEPS,
       PF_FFLOAD, PF_FFREC, PF_FFFLUSH, PF_FFENTRY, PF_NRECS, PF_RD0, PF_RD1, PF_RD2, PF_RD3, PF_RD4, PF_RD5, PF_RDMAX,
       PF_XSTAGE, PF_XLEV, PF_XSEL, PF_REFB, PF_RESULT, PF_CYQ, PF_CYX, PF_CYHIT, PF_CYFILL,
       PF_SB11, PF_SB12, PF_SB21, PF_SB10, PF_SBOTHER, PF_SEGEN, PF_XCALLS, PF_XNLEV, PF_FSFORK, PF_FSDEAD, PF_FSFILT, PF_FSLIM, PF_FK1, PF_FK2, PF_FKBAIL, PF_FORK, PF_ANCCALLS, PF_ANCITER, PF_TPUB, PF_TOWN, PF_TSTOLEN, PF_TWAIT, PF_TRUN, PF_EA0, PF_EA1, PF_EA2, PF_EA3, PF_EA4, PF_EASUM3, PF_EASUM4, PF_RANCH, PF_RANCHMAX, PF_RBR, PF_RBRMAX, PF_N };
#define TALC_PF_NAMES {"probe", "child", "aims", "cycle", "ffwd", "scorebr", "garden", "evalfull", "xdrop", "extnw", "edgemisc", \
                       "anchors", "assemble", "stepb*", "stepe*", "srchb*", "srche*", "prolog", "inittr", "total", "#ffcalls", "#ffsteps", \
                       "ff.load", "ff.record", "ff.flush", "ff.entry", "#ffrecords", "#reads<0.25ms", "#reads<1ms", "#reads<4ms", \
                       "#reads<16ms", "#reads<64ms", "#reads>=64ms", "maxread(10ns)", "x.stage", "x.levels", "x.select", "b.ref", "b.result", \
                       "#cyc.query", "#cyc.exact", "#cyc.found", "cyc.fill%sum", \
                       "#stepb 1->1", "#stepb 1->2", "#stepb 2->1", "#stepb 1->0", "#stepb other", "#stepe generic", "#xdrop calls", "#xdrop levels", \
                       "#ffstop fork", "#ffstop deadend", "#ffstop filter", "#ffstop limit/other", "#forkstep 1 child", "#forkstep fork+deadend", "#forkstep bailed", "forkstep", "#anchor lists", "#anchor level tests", \
                       "#edges published", "#anchors by owner", "#anchors by others", "t.owner waits", "t.run by others", \
                       "#edge anchors<1ms", "#edge anchors<4ms", "#edge anchors<16ms", "#edge anchors<64ms", "#edge anchors>=64ms", "ticks anchors 16-64ms", "ticks anchors>=64ms", "(per read) anchors", "(per read) longest anchor", "(per read) bridge attempts", "(per read) longest bridge attempt"}

struct Wv {
  // kernel constants
  DevParams P; TableView T; SearchLimits C;
  // read
  const uint8_t* read; uint32_t L, n; const uint2* cov; double lambda;
  // scratch
  TrailSet G[2];                    // HBM backing store of the two Trail sets (slots >= HOT)
  int ia;                           // which set is the current one ("competingPaths"); ia^1 = newCompetingPaths
  uint8_t* seqPool;                 // the arena: nBuf buffers of C.seqCap bytes (per search)
  uint32_t nBuf;
  unsigned long long freeMask[NBUF / 64];   // wave-uniform free bitmap of the pool
  uint8_t *ref, *fullPool, *edgeLong, *edgeShort, *edgeTmp, *weak;
  AnchorRec *ancL, *ancR; uint32_t* ancPos;
  FullMeta* fullMeta;
  int *dpG;   // 3 x dpCap ints in HBM
  double* gScores; double* gDists; ValIdx* gVal; Rank4* gRank; uint32_t* gKept;
  uint32_t *regS, *regE, *wOff, *wLen;
  // explorer state (Explorer.hpp:151-174)
  int location, dirRight;          // HEAD/INNER/TAIL as 0/1/2
  uint32_t Ls, Le, Rs, Re;         // m_LEFT_KMpositions / m_RIGHT_KMpositions
  uint32_t weakLen;
  bool complexRegion;
  bool ffPopped;                   // the fast-forward recorded a bridge longer than the reference: its Trail is gone
  int nAncL, nAncR;
  uint32_t refLen;
  int nFull; uint32_t fullUsed;
  EdgeCand best2[2];               // [0] best of m_longPaths, [1] best of m_shortPaths
  int nEdges;
  // counters
  unsigned long long cells, steps;
  uint32_t overflow;
  const TraceBuf* tracep;          // the debug hook's buffers (a copy in the wave's scratch slot: 40 bytes less of LDS)
  bool tracing, traceSteps;        // this read is the traced one / ... and its steps are wanted
  // (later additions go here, at the end: the offsets of the fields above are what the hot code's LDS addressing sees)
  unsigned long long* wideBloom;   // WIDE_BLOOM_WORDS words in HBM
  int* rowPool;                    // ROW_ARENA_INTS ints in HBM (nullptr: every scoring aligns from scratch)
  uint32_t searchNo;               // number of the current search of this wave (stamps the kept rows)
  uint32_t rowStride, rowAvail;    // the current search's records: ints per record, records (0: none)
  uint32_t wideMask;               // words - 1 of the current search's wide filter; 0: the LDS filter is in use
  uint32_t launchStamp;            // a number no other k_search launch of this process carries (stamps the kept rows)
  const CovWord* covw;             // the current read's coverage words (cov: its hit pairs; talc_common.h)
  uint32_t* regH;                  // hit index of every region's start | kRegClean (k_structure), kept in step with regS
  uint32_t LH, RH;                 // ... of the current LEFT / RIGHT region
  const uint32_t* headCov;         // the current read's kHeadCov dense counts
  uint8_t* refBuf;                 // the scratch buffer a search's reference is assembled in (ref points there, or into the read)
  uint32_t noForkStep;             // TALC_NO_FORKSTEP=1 (k_search flags bit 1): forks go through the generic step (A/B switch)
  uint32_t childrenByLane;         // 0 with TALC_CHILDREN_SEQ=1 (flags bit 2): the generic step makes its children one at a time
  // edge tasks (the start anchors of a head / tail search handed to waves that have run out of reads; nullptr: none)
  uint8_t* boxes; uint32_t* avail; uint32_t* qwords;
  uint32_t boxBytes, boxSeqCap, nSlots, mySlot, nWork, taskMinWeak;
  uint32_t taskHeavy;              // a border of at least this many bases is published when its search starts (queue positions below heavyUpTo)
  uint32_t qi, heavyUpTo;          // the current read's position in the work queue; `taskHeavy` applies below this position
  uint32_t holding;                // this wave has had a read (the one before the read it takes now is finished)
  uint32_t stealSeq;               // calls of edge_task_steal (which window of avail[] the next one looks at)
  unsigned long long moreCells, moreSteps;   // of the anchors run for other waves between this wave's reads
  uint32_t nanSeen;                // record_edge has seen a distance that is not a number (the fold is then order-dependent)
  uint32_t taskTest;               // test hook (flags bit 3, TALC_TEST_EDGE_REDO): every anchor another wave has run is flagged for the in-order redo
};

enum { LOC_HEAD = 0, LOC_INNER = 1, LOC_TAIL = 2 };

// The per-wave context lives in LDS as ONE file-scope object: it is wave-uniform state (every lane
// reads the same word = a broadcast ds_read; every lane writes the same value), costs no VGPRs,
// and lets the heavy routines be real (non-inlined) functions without passing anything.
__shared__ Wv g_X;
#define X g_X
#ifdef TALC_PROF
__shared__ uint32_t g_prof[PF_N];   // per wave, in cycles: 2^32 cycles = 1.9 s (32 bits keep the build in the product's LDS size class:
                                     // LDS is handed out in steps of 1280 bytes, and one step more costs three of the 21 waves a CU holds)
#endif

// ---- optional in-kernel cycle accounting (diagnostic build only: -DTALC_PROF) ----
#ifdef TALC_PROF
#define PROF_DECL unsigned long long _pf_t
#define PF_EDGE_T0() (_pf_e0 = __builtin_amdgcn_s_memrealtime())
#define PF_EDGE_T1() (_pf_edge += (uint32_t)(__builtin_amdgcn_s_memrealtime() - _pf_e0))
#define PROF_BEGIN() (_pf_t = __builtin_amdgcn_s_memtime())
#define PROF_END(cat) (g_prof[cat] += (uint32_t)(__builtin_amdgcn_s_memtime() - _pf_t))
#define PROF_DECL2 unsigned long long _pf_t2
#define PROF_COUNT(cat, n) do { if (lane_id() == 0) g_prof[cat] += (uint32_t)(n); } while (0)
#define PROF_BEGIN2() (_pf_t2 = __builtin_amdgcn_s_memtime())
#define PROF_END2(cat) (g_prof[cat] += (uint32_t)(__builtin_amdgcn_s_memtime() - _pf_t2))
#else
#define PROF_DECL
#define PF_EDGE_T0() ((void)0)
#define PF_EDGE_T1() ((void)0)
#define PROF_BEGIN() ((void)0)
#define PROF_END(cat) ((void)0)
#define PROF_DECL2
#define PROF_COUNT(cat, n) ((void)0)
#define PROF_BEGIN2() ((void)0)
#define PROF_END2(cat) ((void)0)
#endif

// count / colour word of the k-mer at position i of the current read (0 when it is not in the table)
TALC_D CovRead cur_cov() { CovRead c; c.hits = (const uint2 TALC_AS1*)X.cov; c.words = (const CovWord TALC_AS1*)X.covw; return c; }
// (a real, cold call: the anchor search reads nearly everything through a region's hit index, and this body at every
//  fallback site was several hundred instructions of the search's code)
TALC_DNC unsigned long long cov_lookup(uint32_t i) { const uint2 v = cov_at(cur_cov(), i); return ((unsigned long long)v.y << 32) | v.x; }
#define COVX(i) ((uint32_t)cov_lookup(i))
#define COVY(i) ((uint32_t)(cov_lookup(i) >> 32))

// DP arrays of the slow paths (sequences longer than the register-resident routines handle):
// always the per-wave HBM arrays, so no pointer ever mixes LDS and HBM provenance.
TALC_D int* dp_array(int which, int need) {
  if ((uint32_t)need > X.C.dpCap) { X.overflow |= OVF_DP; }
  return X.dpG + (uint64_t)which * X.C.dpCap;
}

#define NW_REG_NB 12
TALC_DNC int nw_score(const uint8_t* a, int la, const uint8_t* b, int lb, int match, int mismatch, int gap,
                    bool freeBegin) {
  la = uni(la); lb = uni(lb);
  // the longer sequence spans the lanes (the score is symmetric in its arguments)
  if (la < lb) { const uint8_t* t = a; a = b; b = t; int tl = la; la = lb; lb = tl; }
  unsigned long long ncells = 0;
  if (la <= 64 * NW_REG_NB) {
    // an instance per block width (columns per lane): the sweep's inner loop is unrolled over the block, and a 12-wide
    // instance spends 12 predicated cell updates per step on a 130-base sequence that has 3 per lane
    const int B = (la + 63) >> 6;
    int r;
    if (B <= 1) r = wave_nw_reg<1>(a, la, b, lb, match, mismatch, gap, freeBegin, ncells);
    else if (B <= 2) r = wave_nw_reg<2>(a, la, b, lb, match, mismatch, gap, freeBegin, ncells);
    else if (B <= 3) r = wave_nw_reg<3>(a, la, b, lb, match, mismatch, gap, freeBegin, ncells);
    else if (B <= 4) r = wave_nw_reg<4>(a, la, b, lb, match, mismatch, gap, freeBegin, ncells);
    else if (B <= 6) r = wave_nw_reg<6>(a, la, b, lb, match, mismatch, gap, freeBegin, ncells);
    else if (B <= 8) r = wave_nw_reg<8>(a, la, b, lb, match, mismatch, gap, freeBegin, ncells);
    else r = wave_nw_reg<NW_REG_NB>(a, la, b, lb, match, mismatch, gap, freeBegin, ncells);
    X.cells += ncells;
    return r;
  }
  if ((uint32_t)(la + 1) > X.C.dpCap) { X.overflow |= OVF_DP; return 0; }
  int* row = dp_array(0, la + 1);
  const int r = wave_nw(a, la, b, lb, match, mismatch, gap, freeBegin, row, ncells);
  X.cells += ncells;
  return r;
}

#define NW2_REG_NB 8
// the lane-skewed sweeps behind edit_and_lcs (what is left when neither the wavefront nor the bit-vector routines apply:
// a cold call); returns editScore | lcsLen << 32 for the parts not yet known
TALC_DNC unsigned long long edit_and_lcs_sweep(const uint8_t* a, int la_, const uint8_t* b, int lb_, bool haveEdit_, bool haveLcs_) {
  const int la = uni(la_), lb = uni(lb_);
  const bool haveEdit = uni((int)haveEdit_) != 0, haveLcs = uni((int)haveLcs_) != 0;
  int editScore = 0, lcsLen = 0;
  if (!haveEdit && !haveLcs && la <= 64 * NW2_REG_NB) {
    unsigned long long ncells = 0;
    wave_edit_lcs_reg<NW2_REG_NB>(a, la, b, lb, editScore, lcsLen, ncells);
    X.cells += ncells;
  } else {
    if (!haveEdit) editScore = nw_score(a, la, b, lb, 0, -1, -1, false);
    if (!haveLcs) lcsLen = nw_score(a, la, b, lb, 1, 0, 0, false);
  }
  return ((unsigned long long)(uint32_t)uni(lcsLen) << 32) | (uint32_t)uni(editScore);
}

// edit score (globalAlignment 0/-1/-1) and LCS length (localAlignment 1/0/0) of the same pair
// needEdit = false: only the LCS is wanted (the caller does not use the edit score: a single bridge candidate is not
// compared with anything, Trajectory.cpp:282-303; an edge whose extension stopped takes the extension's own score,
// Trail.cpp:408-434) — editScore is then left at 0.
// acceptLcs > 0 (with needEdit = false): the caller only asks whether the LCS reaches acceptLcs (a threshold test,
// Explorer.cpp:973) — lcsLen is then either the exact LCS or, as soon as the wavefront proves LCS >= acceptLcs,
// acceptLcs itself.
TALC_DN void edit_and_lcs(const uint8_t* a_, int la, const uint8_t* b_, int lb, int& editScore, int& lcsLen, bool needEdit_,
                          int acceptLcs_) {
  la = uni(la); lb = uni(lb);
  const bool needEdit = uni((int)needEdit_) != 0;
  const int acceptLcs = uni(acceptLcs_);
  const int acceptA = (!needEdit && acceptLcs > 0) ? 2 * acceptLcs : INT_MAX;
  const uint8_t* a = uni_ptr(a_); const uint8_t* b = uni_ptr(b_);
  if (la < lb) { const uint8_t* t = a; a = b; b = t; int tl = la; la = lb; lb = tl; }
  // both measures as wavefronts over the two sequences staged in LDS: the levels needed are the edit distance, and
  // (without the substitution move) la + lb - 2 LCS
  bool haveEdit = !needEdit, haveLcs = false;
  if (!needEdit) editScore = 0;
  constexpr int STAGE = 3 * LDS_DP_CAP * 4;
  const int qpad = (la + 16) & ~7;
  if (lb > 0 && qpad + lb + 16 <= STAGE) {
    uint8_t TALC_AS3* stage = (uint8_t TALC_AS3*)g_dp;
    const int l = lane_id();
    gcu8 ga = (gcu8)a; gcu8 gb = (gcu8)b;
    stage_copy(stage, ga, la);
    stage_copy(stage + qpad, gb, lb);
    if (l == 0) { stage[la] = 0xF0; stage[qpad + lb] = 0xF1; }
    WSYNC();
    unsigned long long ncells = 0;
    const int lo = la - lb;   // either distance is at least the length difference
    int d = -1;
    if (needEdit) {
      int ed = -1;
      if (lo <= 31 && la <= 220) ed = wave_wfa_global<1, true>(stage, qpad, la, lb, ncells);
      if (ed < 0 && lo <= 63 && la <= 440) ed = wave_wfa_global<2, true>(stage, qpad, la, lb, ncells);
      if (ed < 0 && lo <= 127) ed = wave_wfa_global<4, true>(stage, qpad, la, lb, ncells);
      if (ed >= 0) {
        editScore = -ed; haveEdit = true;
        const int hi = 2 * ed;   // each substitution is at most one insertion plus one deletion
        if (hi <= 31) d = wave_wfa_global<1, false>(stage, qpad, la, lb, ncells);
        else if (hi <= 63) d = wave_wfa_global<2, false>(stage, qpad, la, lb, ncells);
        else if (hi <= 127) d = wave_wfa_global<4, false>(stage, qpad, la, lb, ncells);
      }
    } else {   // no bound from the edit distance: narrowest wavefront first (-1 = it needs more levels than that width holds)
      if (lo <= 31 && la <= 220) d = wave_wfa_global<1, false>(stage, qpad, la, lb, ncells, acceptA);
      if (d == -1 && lo <= 63 && la <= 440) d = wave_wfa_global<2, false>(stage, qpad, la, lb, ncells, acceptA);
      // (beyond that the bit-parallel LCS below: its cost is per row, not per level x diagonal)
    }
    if (d >= 0) { lcsLen = (la + lb - d) >> 1; haveLcs = true; }
    else if (d == -2) { lcsLen = acceptLcs; haveLcs = true; }   // proven: LCS >= acceptLcs
    X.cells += ncells;
    WSYNC();
  }
  if (haveEdit && haveLcs) return;
  // long or dissimilar sequences: the bit-vector LCS and edit distance (exact; columns beyond 4096 in blocks whose
  // hand-over bits — one or two per row — go through the first HBM DP array)
  unsigned long long* const work = ((uint64_t)(max(la, lb) + 63) / 64 * 2 * 8 <= (uint64_t)X.C.dpCap * 4) ? (unsigned long long*)X.dpG : nullptr;
  if (!haveLcs) {
    unsigned long long ncells = 0;
    const int z = wave_lcs_bitpar(a, la, b, lb, ncells, work);
    if (z >= 0) { lcsLen = z; haveLcs = true; X.cells += ncells; }
  }
  if (!haveEdit) {
    unsigned long long ncells = 0;
    const int ed = wave_edit_bitpar(a, la, b, lb, ncells, work);
    if (ed >= 0) { editScore = -ed; haveEdit = true; X.cells += ncells; }
  }
  if (haveEdit && haveLcs) return;
  const unsigned long long packed = edit_and_lcs_sweep(a, la, b, lb, haveEdit, haveLcs);   // (never reached with the HBM arrays in place)
  if (!haveEdit) editScore = (int)(uint32_t)packed;
  if (!haveLcs) lcsLen = (int)(uint32_t)(packed >> 32);
}

// The question a LONE bridge candidate is asked (Explorer.cpp:973: is its identity with the reference at least
// MIN_INNER): does the LCS of the two sequences reach acceptLcs?  Returns acceptLcs or the exact LCS, as edit_and_lcs
// with needEdit = false does — through the two narrow wavefront instances only, which settle nearly every gap of a
// unique-sequence graph; anything else returns -1 and the caller asks edit_and_lcs.  (A function of its own: edit_and_lcs carries six wavefront
// instances and the bit-vector routines, 7000 instructions, and this is the call the bridge search makes per gap.)
TALC_DN int lcs_reaches(const uint8_t* a_, int la, const uint8_t* b_, int lb, int acceptLcs_) {
  la = uni(la); lb = uni(lb);
  const int acceptLcs = uni(acceptLcs_);
  const uint8_t* a = uni_ptr(a_); const uint8_t* b = uni_ptr(b_);
  if (la < lb) { const uint8_t* t = a; a = b; b = t; int tl = la; la = lb; lb = tl; }
  constexpr int STAGE = 3 * LDS_DP_CAP * 4;
  const int qpad = (la + 16) & ~7;
  const int lo = la - lb;
  if (acceptLcs > 0 && lb > 0 && qpad + lb + 16 <= STAGE && lo <= 63 && la <= 440) {
    uint8_t TALC_AS3* stage = (uint8_t TALC_AS3*)g_dp;
    stage_copy(stage, (gcu8)a, la);
    stage_copy(stage + qpad, (gcu8)b, lb);
    if (lane_id() == 0) { stage[la] = 0xF0; stage[qpad + lb] = 0xF1; }
    WSYNC();
    unsigned long long ncells = 0;
    int d = -1;
    if (lo <= 31 && la <= 220) d = wave_wfa_global<1, false>(stage, qpad, la, lb, ncells, 2 * acceptLcs);
    if (d == -1) d = wave_wfa_global<2, false>(stage, qpad, la, lb, ncells, 2 * acceptLcs);
    X.cells += ncells;
    WSYNC();
    if (d >= 0) return (la + lb - d) >> 1;
    if (d == -2) return acceptLcs;
  }
  return -1;   // (not settled here: the caller asks edit_and_lcs — this function makes no call and saves nothing)
}

// ------------------------------------------------------------------ trace helpers
// (the debug hook's code is real, cold calls: inlined at its dozen sites it was a tenth of k_search's instructions and
//  carried forty of its spill slots — for a path no production launch takes)
TALC_DNC void trace_rec(int kind, int a, int b, int c, int d, double x, const uint8_t* s, uint32_t slen, bool rev) {
  if (!X.tracing) return;
  WSYNC();
  if (lane_id() == 0) {
    const TraceBuf T = *X.tracep;
    uint32_t k = atomicAdd(T.nrec, 1u);
    uint32_t off = 0;
    if (slen) { off = atomicAdd(T.npool, slen); }
    if (k < T.cap) {
      if (slen && off + slen <= T.poolCap) for (uint32_t i = 0; i < slen; ++i) T.pool[off + i] = rev ? s[slen - 1 - i] : s[i];
      else if (slen) slen = 0;
      T.recs[k] = TraceRec{kind, a, b, c, d, x, off, slen};
    }
  }
  WSYNC();
}

// ------------------------------------------------------------------ sequence buffer pool
// Trails are flat byte strings; a child that is the LAST successor of its parent takes over the
// parent's buffer and just appends its base (the common single-path case copies nothing), the
// other children get a fresh buffer and copy.  The free bitmap is wave-uniform register state.
TALC_D void pool_reset() {   // buffers 0 .. nBuf-1 are free
  const int nb = (int)X.nBuf;
#pragma unroll
  for (int w = 0; w < NBUF / 64; ++w) X.freeMask[w] = (nb >= 64 * (w + 1)) ? ~0ull : (nb > 64 * w ? ((1ull << (nb - 64 * w)) - 1) : 0ull);
}
// the Trail buffers of the search that starts now: stride for paths of up to K + pathMax bases (+ the base a step
// appends before it checks its limits), as many of them as the arena holds
TALC_D bool pool_shape(uint32_t pathMax) {
  const uint32_t stride = (X.P.K + pathMax + 16u + 15u) & ~15u;
  if (stride > X.C.seqArena) { X.overflow |= OVF_SEQ; return false; }
  X.C.seqCap = stride;
  X.nBuf = min((uint32_t)NBUF, X.C.seqArena / stride);
  return true;
}
TALC_D int pool_alloc() {
  int id = -1;
#pragma unroll
  for (int w = 0; w < NBUF / 64; ++w) {
    if (id < 0 && X.freeMask[w]) {
      const int b = (int)__ffsll((long long)X.freeMask[w]) - 1;
      X.freeMask[w] &= ~(1ull << b);
      id = w * 64 + b;
    }
  }
  if (id < 0) { X.overflow |= OVF_TRAILS; id = 0; }
  return id;
}
TALC_D void pool_free(uint32_t id) {
#pragma unroll
  for (int w = 0; w < NBUF / 64; ++w) if ((int)(id >> 6) == w) X.freeMask[w] |= (1ull << (id & 63));
}
// the records of the search that starts now (its reference is complete): stride, count, a new search number
TALC_D void rows_shape() {
  const uint32_t n = (uint32_t)uni((int)X.refLen);
  uint32_t stride = 0, avail = 0;
  const uint32_t sn = (uint32_t)uni((int)X.searchNo) + 1u;
  if (X.rowPool != nullptr && n <= (uint32_t)ROW_MAX_REF && sn < 0x7FFF0u) {   // (a wave's half-millionth search of a launch goes without)
    stride = (n + 2u + 3u) & ~3u;
    avail = min((uint32_t)NBUF, (uint32_t)ROW_ARENA_INTS / stride);
  }
  X.rowStride = stride; X.rowAvail = avail;
  X.searchNo = min(sn, 0x7FFF0u);
}
TALC_D int* row_of(uint32_t buf) { return X.rowPool + (uint64_t)buf * X.rowStride; }
// rows of the alignment matrix the record of buffer `buf` covers in the current search (0: none)
TALC_D uint32_t row_covered(uint32_t buf) {
  const int* rec = row_of(buf);
  const uint32_t lo = (uint32_t)uni(rec[0]), hi = (uint32_t)uni(rec[(uint32_t)uni((int)X.refLen) + 1u]);
  return (hi == (uint32_t)uni((int)X.launchStamp) && (lo >> ROW_COV_BITS) == (uint32_t)uni((int)X.searchNo)) ? (lo & ((1u << ROW_COV_BITS) - 1u)) : 0u;
}
TALC_D void row_set_covered(uint32_t buf, uint32_t covered) {
  if (lane_id() == 0) {
    int* rec = row_of(buf);
    rec[0] = (int)((X.searchNo << ROW_COV_BITS) | (covered & ((1u << ROW_COV_BITS) - 1u)));
    rec[X.refLen + 1u] = (int)X.launchStamp;
  }
}
// the record of the Trail in buffer `src` goes with a copy of that Trail into buffer `dst`: the row AND its stamps, so the
// copy covers exactly what the original covers (nothing, if the original's stamps are not this search's).  The caller has
// synchronised since the record was last written (branch_copy, garden: both do before they copy the Trail's bases), and
// the copy neither waits for that copy's stores nor looks at the stamps first: each was a memory round trip per child.
TALC_DNC void row_copy(uint32_t dst_, uint32_t src_) {
  const uint32_t dst = (uint32_t)uni((int)dst_), src = (uint32_t)uni((int)src_);
  const uint32_t avail = (uint32_t)uni((int)X.rowAvail);
  if (dst >= avail) return;
  if (src < avail) wave_copy((uint8_t*)row_of(dst), (const uint8_t*)row_of(src), (uint32_t)uni((int)X.rowStride) * 4u);
  else row_set_covered(dst, 0u);
}
// slots >= HOT live in HBM; these are real calls so that the optimiser never merges an LDS and an
// HBM access into one access through a pointer of mixed provenance (which would become flat_*)
TALC_DNC TrailRec tr_get_slow(int set, int t) {
  const TrailSet& S = X.G[set];
  TrailRec r;
  r.kmer = S.kmer[t]; r.nmask = S.nmask[t]; r.dist = S.dist[t]; r.cnt = S.cnt[t]; r.score = S.score[t]; r.fail = S.fail[t];
  r.lanc = S.lanc[t]; r.ranc = S.ranc[t]; r.buf = S.buf[t];
  return r;
}
TALC_DNC void tr_put_slow(int set, int t, TrailRec r) {
  TrailSet& S = X.G[set];
  S.kmer[t] = r.kmer; S.nmask[t] = r.nmask; S.dist[t] = r.dist; S.cnt[t] = r.cnt; S.score[t] = r.score; S.fail[t] = r.fail;
  S.lanc[t] = r.lanc; S.ranc[t] = r.ranc; S.buf[t] = r.buf;
}
TALC_D TrailRec tr_get(int set, int t) {
  if (t < HOT) return g_hot[set * HOT + t];
  return tr_get_slow(set, t);
}
TALC_D void tr_put(int set, int t, const TrailRec& r) {   // call from ONE lane
  if (t < HOT) { g_hot[set * HOT + t] = r; return; }
  tr_put_slow(set, t, r);
}
TALC_D void tr_sync(int t) { if (t < HOT) LSYNC(); else WSYNC(); }
TALC_D uint32_t tr_buf(int set, int t) { return (t < HOT) ? g_hot[set * HOT + t].buf : tr_get_slow(set, t).buf; }
TALC_D uint8_t* trail_seq(int set, int t) { return X.seqPool + (uint64_t)tr_buf(set, t) * X.C.seqCap; }

// Bloom filter over the k-mers of the current search (one 64-bit word per lane = 4096 bits; a k-mer
// sets two bits of one word): a superset of every live Trail's k-mers, so "definitely absent" skips
// the exact window search of ThinkIveAlreadyGotThere (Trail.cpp:289-302); "maybe" falls through to it.
// The hash is the table hash of the k-mer's successor key (its K-1 bases on the growing side), which the
// fast-forward loop computes anyway for the next probe.
TALC_D uint64_t bloom_hash(uint64_t kmer, uint64_t nmask) {
  const uint32_t K = X.P.K;
  const uint64_t key = X.dirRight ? (kmer & ((1ULL << (2 * (K - 1))) - 1)) : (kmer >> 2);
  return table_hash(key) ^ (nmask * 0x9E3779B97F4A7C15ULL);
}
TALC_D int bloom_word(uint64_t h) { return (int)(h >> 57); }
TALC_D unsigned long long bloom_mask(uint64_t h) { return (1ull << ((h >> 51) & 63)) | (1ull << ((h >> 45) & 63)); }
static_assert(BLOOM_WORDS == 128, "bloom_word takes 7 hash bits");
// the wide filter's word and bits from the same hash (upper 32 bits `hv`): 14 + 6 + 6 bits below the LDS filter's 7
TALC_D uint32_t wide_word(uint32_t hv, uint32_t mask) { return (hv >> 18) & mask; }
TALC_D unsigned long long wide_bits(uint32_t hv) { return (1ull << ((hv >> 12) & 63u)) | (1ull << ((hv >> 6) & 63u)); }
TALC_D unsigned long long wide_load(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
TALC_D void wide_or(unsigned long long* p, unsigned long long m) {
  __hip_atomic_fetch_or(p, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// (a search with the wide filter keeps the LDS one as well and asks it first: the wide one costs a memory round trip per
//  question, and the LDS one — however full a long walk leaves it — still answers most of them; both are supersets of
//  what was entered, so "absent" from either is exact)
TALC_D bool bloom_query_insert(uint64_t kmer, uint64_t nmask) {
  const uint64_t h = bloom_hash(kmer, nmask);
  const uint32_t wm = (uint32_t)uni((int)X.wideMask);
  const int w = bloom_word(h);
  const unsigned long long m = bloom_mask(h), v = g_bloom[w];
  bool maybe = (v & m) == m;
  if (lane_id() == 0) g_bloom[w] = v | m;
  if (wm != 0u) {
    const uint32_t hv = (uint32_t)(h >> 32);
    unsigned long long* ww = X.wideBloom + wide_word(hv, wm);
    const unsigned long long mw = wide_bits(hv);
    if (maybe) maybe = (wide_load(ww) & mw) == mw;
    if (lane_id() == 0) wide_or(ww, mw);
  }
  LSYNC();
  return maybe;
}
// The generic step asks first and enters afterwards: the k-mers of a step's children go into the filter together when
// the step is over (bloom_flush; lane i holds the upper hash half of the step's i-th child), so that a child is not
// answered "maybe" because a sibling or a cousin reached the same k-mer in this very step — a k-mer of its own path was
// entered at an earlier step, by one of its ancestors.  (Two Trails that went round the two sides of a substitution
// bubble walk on in step, k-mer for k-mer, until gardening drops one: with the entries made at once each of their steps
// paid for an exact window search — a fifth of a launch over a branching graph.)
TALC_D bool bloom_query(uint64_t kmer, uint64_t nmask, uint32_t& hv) {
  const uint64_t h = bloom_hash(kmer, nmask);
  hv = (uint32_t)(h >> 32);
  const unsigned long long m = bloom_mask(h);
  if ((g_bloom[bloom_word(h)] & m) != m) return false;
  const uint32_t wm = (uint32_t)uni((int)X.wideMask);
  if (wm != 0u) {
    const unsigned long long mw = wide_bits(hv);
    return (wide_load(X.wideBloom + wide_word(hv, wm)) & mw) == mw;
  }
  return true;
}
TALC_D void bloom_flush(uint32_t pend, int n) {
  if (n == 0) return;
  const uint32_t wm = (uint32_t)uni((int)X.wideMask);
  if (lane_id() < n) {
    atomicOr(&g_bloom[pend >> 25], (1ull << ((pend >> 19) & 63u)) | (1ull << ((pend >> 13) & 63u)));
    if (wm != 0u) wide_or(X.wideBloom + wide_word(pend, wm), wide_bits(pend));
  }
  LSYNC();
}

// ------------------------------------------------------------------ anchors (Explorer.cpp:413-543)
// side 0: anchorLEFTHandSide (walks the LEFT region leftwards from its end, degree towards RIGHT)
// side 1: anchorRIGHTHandSide (walks the RIGHT region rightwards from its start, degree towards LEFT)
TALC_DNC void sort_anchors_long(AnchorRec* anc, int n, double cc) { gnu_sort(anc, n, LessAnchor{cc}); }

// LEAF: the instance for the common anchor list — a region of at most 64 k-mers whose degrees k_coverage left with the
// counts, a dozen anchors at most — makes no call at all, so it has nothing to save or restore (2.4 M calls per config-2
// launch); whatever it cannot settle it reports by returning false BEFORE any flag is raised, and the general instance
// redoes the list from scratch.
template <bool LEAF>
TALC_D bool build_anchors_body(int side) {
  PROF_DECL;
  PROF_BEGIN();
  const DevParams& P = X.P;
  const uint32_t K = P.K, MINC = P.MIN_COUNT;
  const int l = lane_id();
  AnchorRec* anc = side == 0 ? X.ancL : X.ancR;
  uint32_t* anchorPos = X.ancPos;
  const uint32_t cap = X.C.anchCap;
  const uint32_t rs = side == 0 ? X.Ls : X.Rs, re = side == 0 ? X.Le : X.Re;
  const uint32_t nbKmers = re - rs + 1;
  const uint32_t pivot = side == 0 ? re : rs;
  const uint32_t limit = side == 0 ? rs : re;
  // walk away from the pivot while each count is "expected" after the last retained one; a count that is not, but is
  // still IN, starts a new level and is recorded (Explorer.cpp:427-447 / 493-513).  64 positions per pass: every lane
  // tests its position against the current level; the first lane that stops either becomes the new level (the lanes
  // behind it are tested again) or ends the walk.
  // The coverage the walks below look at, fetched ONCE, one position per lane: the region's pairs when it has at most 64
  // k-mers (nearly always), and the counts of the read's first kHeadCov positions (the count the reference records for
  // anchor number a is m_coverage[a], Explorer.cpp:454,520: a position of the read, not of the region; k_structure left
  // them in one line per read).  Everything after this is lane traffic; a longer region (or a later anchor number)
  // reads memory position by position.
  const CovRead cr = cur_cov();
  const bool inRegs = nbKmers <= 64u;
  if (LEAF && !inRegs) return false;
  uint2 regv = make_uint2(0u, 0u), headv = make_uint2(0u, 0u);
  // (a clean region — k_structure — holds the pair of position p at its start's index + (p - start): one load, no word)
  const uint32_t regHidx = (uint32_t)uni((int)(side == 0 ? X.LH : X.RH));
  const bool clean = (regHidx & kRegClean) != 0u;
  const uint32_t hbase = regHidx & ~kRegClean;
  if (inRegs && (uint32_t)l < nbKmers) {
    if (clean) { const v2u32 e = *(const v2u32 TALC_AS1*)(cr.hits + hbase + (uint32_t)l); regv = make_uint2(e.x, e.y); }
    else regv = cov_at(cr, rs + (uint32_t)l);
  }
  if (l < kHeadCov) headv.x = ((const uint32_t TALC_AS1*)X.headCov)[l];
  // ... parked in the wave's DP stage (LDS; no alignment routine runs inside this function): values that lived in vector
  // registers over this function's calls would each cost a stack save and a restore per call of it
  uint32_t TALC_AS3* const park = (uint32_t TALC_AS3*)g_dp;
  park[l] = regv.x; park[64 + l] = regv.y; park[128 + l] = headv.x;
  // ... and the region's bases (at most 64 + K - 1 of them), two per lane: an anchor's k-mer is then read from LDS instead
  // of by a round trip to the read per anchor
  uint8_t TALC_AS3* const pbase = (uint8_t TALC_AS3*)(park + 192);
  if (inRegs) {
    const uint32_t nb = nbKmers + K - 1u;
    gcu8 rd = (gcu8)X.read + rs;
    if ((uint32_t)l < nb) pbase[l] = rd[l];
    if ((uint32_t)l + 64u < nb) pbase[l + 64] = rd[l + 64];
  }
  LSYNC();
  // packed k-mer (+ N mask) of the read's position pos (uniform, inside the region): as wave_kmer_at
  auto kmer_at = [&](uint32_t pos, uint64_t& kmer, uint64_t& nmask) {
    if (!LEAF && !inRegs) { wave_kmer_at(X.read + pos, (int)K, kmer, nmask); return; }
    const uint32_t c = ((uint32_t)l < K) ? (uint32_t)pbase[pos - rs + (uint32_t)l] : 0u;
    nmask = ballot64(((uint32_t)l < K) && (c > 3u));
    uint64_t v = ((uint32_t)l < K) ? ((uint64_t)(c & 3u) << (2 * (K - 1 - (uint32_t)l))) : 0ull;
    if (c > 3u) v = 0;
    kmer = ((uint64_t)wave_or_u32((uint32_t)(v >> 32)) << 32) | wave_or_u32((uint32_t)v);
  };
  auto run_x = [&](uint32_t pos) -> uint32_t { return park[(pos - rs) & 63u]; };
  auto run_y = [&](uint32_t pos) -> uint32_t { return park[64u + ((pos - rs) & 63u)]; };
#define RUNX(pos) ((LEAF || inRegs) ? run_x(pos) : (clean ? (*(const v2u32 TALC_AS1*)(cr.hits + hbase + ((pos) - rs))).x : COVX(pos)))
#define RUNY(pos) ((LEAF || inRegs) ? run_y(pos) : (clean ? (*(const v2u32 TALC_AS1*)(cr.hits + hbase + ((pos) - rs))).y : COVY(pos)))
#define HEADX(a) (((a) < (uint32_t)kHeadCov) ? (uint32_t)uni((int)park[128u + (a)]) : (LEAF ? 0u : COVX(a)))
  uint32_t current_count = (uint32_t)uni((int)RUNX(pivot));
#ifdef TALC_PROF
  if (l == 0) g_prof[PF_ANCCALLS] += 1;
#endif
  uint32_t nPos = 0;
  if (l == 0) anchorPos[0] = pivot;
  nPos = 1;
  {
    const uint32_t remaining = (side == 0) ? (pivot - limit) : (limit - pivot);   // positions beyond the pivot
    bool alive = true;
    for (uint32_t visited = 0; alive && visited < remaining; visited += 64) {
      const uint32_t idx = visited + (uint32_t)l;
      const bool valid = idx < remaining;
      const uint32_t pos = (side == 0) ? (pivot - 1 - idx) : (pivot + 1 + idx);
      const uint32_t ncAll = RUNX(valid ? pos : pivot);
      const uint32_t nc = valid ? ncAll : 0u;
      const bool inRange = valid & (nc >= MINC) & ((double)nc < P.MAX_IN_COUNT);
      int from = 0;
      while (from < 64) {
#ifdef TALC_PROF
        if (l == 0) g_prof[PF_ANCITER] += 1;
#endif
        const bool go = inRange && is_expected_by_last_node(P.ALPHA, nc, current_count);
        const unsigned long long stops = ballot64(valid && !go) & (~0ull << from);
        if (stops == 0ull) break;
        const int f = (int)__builtin_ctzll(stops);
        const bool fIn = ((ballot64(inRange) >> f) & 1ull) != 0ull;
        if ((current_count >= MINC) & fIn) {
          const uint32_t pf = (uint32_t)lane_get((int)pos, f);
          if (nPos < cap) { if (l == 0) anchorPos[nPos] = pf; ++nPos; } else { if (LEAF) return false; X.overflow |= OVF_ANCHORS; }
          current_count = (uint32_t)lane_get((int)nc, f);
          from = f + 1;
        } else { alive = false; break; }
      }
    }
  }
  WSYNC();
  uint32_t nAnc = 0;
  uint32_t firstAnchorPos = 0;
  // out-degree of the k-mer at `pos` (per lane) towards the side's direction: k_coverage left it next to the count for
  // every k-mer of the table; any other position (count 0: never a recorded one) is probed here
  const int degDir = (side == 0) ? 1 : 0;
  const int degShift = (side == 0) ? kCovDegRShift : kCovDegLShift;
  bool needGeneral = false;   // (LEAF: a degree k_coverage did not leave)
  auto degree_at = [&](bool want, uint32_t pos) -> int {
    const uint32_t cyAll = RUNY(want ? pos : pivot);
    const uint32_t cy = want ? cyAll : kCovDegKnown;
    int degree = (int)((cy >> degShift) & 7u);
    const bool unknown = want && !(cy & kCovDegKnown);
    if (ballot64(unknown) != 0ull) {
      if (LEAF) { needGeneral = true; return 0; }
      if (unknown) { uint64_t km, nm; lane_kmer_at(X.read + pos, (int)K, km, nm); degree = dev_out_degree(X.T, MINC, km, nm, degDir); }
    }
    return want ? degree : 0;
  };
  for (uint32_t base = 0; base < nPos; base += 64) {   // one recorded position per lane
    const uint32_t a = base + (uint32_t)l;
    const bool valid = a < nPos;
    const uint32_t pos = valid ? anchorPos[a] : 0u;
    const int degree = degree_at(valid && a != 0, pos);   // (the pivot is taken whatever its degree)
    if (LEAF && needGeneral) return false;
    unsigned long long take = ballot64(valid && ((a == 0) || (degree > 1)));
    while (take != 0ull) {
      const int f = (int)__builtin_ctzll(take);
      take &= take - 1ull;
      // Explorer.cpp:454,520: the recorded count is m_coverage[anc] (loop index), not [anchorPos[anc]]
      if (nAnc < cap) {
        const uint32_t pf = (uint32_t)lane_get((int)pos, f);
        uint64_t km, nm;
        kmer_at(pf, km, nm);
        if (LEAF && base + (uint32_t)f >= (uint32_t)kHeadCov) return false;
        const uint32_t recorded = HEADX(base + (uint32_t)f);
        if (l == 0) anc[nAnc] = AnchorRec{km, nm, pf, recorded};
        if (nAnc == 0) firstAnchorPos = pf;
        ++nAnc;
      } else { if (LEAF) return false; X.overflow |= OVF_ANCHORS; }
    }
  }
  const uint32_t want = min(P.MIN_START_ANCHORS, nbKmers);
  if (nAnc < want) {
    const uint32_t j = pivot;
    bool goFurther = true;
    if (side == 0) {
      // :461-475: walk from the pivot towards the region start and take every k-mer with out-degree > 1 until
      // `want` anchors exist.  (:465 `for(unsigned i(0); i<size; --i)` looks at element 0 only: once the position
      // of the FIRST anchor is met the walk records nothing more.)  One position per lane, 64 probes in flight.
      uint32_t stopAt = limit;   // lowest position still examined
      if (nAnc > 0 && firstAnchorPos < pivot && firstAnchorPos >= limit) stopAt = firstAnchorPos + 1;
      const uint32_t remaining = (pivot > stopAt) ? (pivot - stopAt) : 0u;
      bool full = false;
      for (uint32_t visited = 0; !full && (nAnc < want) && visited < remaining; visited += 64) {
        const uint32_t idx = visited + (uint32_t)l;
        const bool valid = idx < remaining;
        const uint32_t pos = pivot - 1 - (valid ? idx : 0u);
        const int degree = degree_at(valid, pos);
        if (LEAF && needGeneral) return false;
        unsigned long long branching = ballot64(valid && degree > 1);
        while (branching != 0ull && nAnc < want) {
          const int f = (int)__builtin_ctzll(branching);
          branching &= branching - 1ull;
          if (nAnc < cap) {
            const uint32_t pf = (uint32_t)lane_get((int)pos, f);
            uint64_t km, nm;
            kmer_at(pf, km, nm);
            const uint32_t cpf = (uint32_t)uni((int)RUNX(pf));
            if (l == 0) anc[nAnc] = AnchorRec{km, nm, pf, cpf};
            ++nAnc;
          } else { if (LEAF) return false; X.overflow |= OVF_ANCHORS; full = true; break; }
        }
      }
    } else {
      // :529-540: j is DEcremented (sic); with unsigned wrap-around the loop can only ever add the
      // k-mer at pivot+1 (first iteration); afterwards goFurther is false for good (j+1 == pivot).
      if (((j + 1) <= limit) & (nAnc < want)) {
        if (nAnc > 0) goFurther &= (firstAnchorPos != (j + 1));
        if (goFurther) {
          uint64_t km, nm;
          kmer_at(j + 1, km, nm);
          // (its degree towards LEFT is with its count — j + 1 lies in the region, a k-mer of the table —: no probe)
          const uint32_t cy1 = (uint32_t)uni((int)RUNY(j + 1));
          int degree;
          if (cy1 & kCovDegKnown) degree = (int)((cy1 >> kCovDegLShift) & 7u);
          else { if (LEAF) return false; degree = dev_out_degree(X.T, MINC, km, nm, 0); }
          if (degree > 1) {
            const uint32_t cj1 = (uint32_t)uni((int)RUNX(j + 1));
            if (nAnc < cap) { if (l == 0) anc[nAnc] = AnchorRec{km, nm, j + 1, cj1}; ++nAnc; }
            else { if (LEAF) return false; X.overflow |= OVF_ANCHORS; }
          }
        }
      }
    }
  }
  WSYNC();
  // std::sort of a list of at most 16 elements IS its final insertion sort (the introsort loop does nothing below 17):
  // nearly every anchor list is that short, and the full routine — a real call — keeps its stack arrays and registers
  // out of this function
  if (LEAF && nAnc > 16) return false;
  if (l == 0 && nAnc > 1) {
    if (LEAF || nAnc <= 16) gs_insertion_sort(anc, 0, (int)nAnc, LessAnchor{X.lambda / P.ERR});
    else sort_anchors_long(anc, (int)nAnc, X.lambda / P.ERR);
  }
  WSYNC();
  if (side == 0) X.nAncL = (int)nAnc; else X.nAncR = (int)nAnc;
#undef RUNX
#undef RUNY
#undef HEADX
  PROF_END(PF_ANCHORS);
  return true;
}
TALC_DN bool build_anchors_leaf(int side) { return build_anchors_body<true>(side); }
TALC_DN void build_anchors_general(int side) { (void)build_anchors_body<false>(side); }
TALC_D void build_anchors(int side) {
  if (!uni((int)build_anchors_leaf(side))) build_anchors_general(side);
}

// the anti-diagonal form of the x-drop extension (anti-diagonals in LDS, or in HBM when too long): the fallback for a band
// wider than the wavefront routines take (a real, cold call: it keeps two instances of the sweep out of the extension's code);
// returns extCols | extRows << 32
TALC_DNC unsigned long long xdrop_sweep(const uint8_t* q, int qlen_, const uint8_t* d, int dlen_, int xdrop_) {
  const int qlen = uni(qlen_), dlen = uni(dlen_), xdrop = uni(xdrop_);
  q = uni_ptr(q); d = uni_ptr(d);
  const int need = qlen + 3;
  int extCols = 0, extRows = 0;
  unsigned long long ncells = 0;
  if (need <= LDS_DP_CAP) {
    XDropBufT<int TALC_AS3*> buf;
    buf.d1 = (int TALC_AS3*)g_dp; buf.d2 = buf.d1 + LDS_DP_CAP; buf.d3 = buf.d2 + LDS_DP_CAP;
    wave_xdrop<int TALC_AS3*, true>(q, qlen, d, dlen, 0, -1, -1, xdrop, buf, extCols, extRows, ncells);
  } else {
    XDropBuf buf;
    buf.d1 = dp_array(0, need); buf.d2 = dp_array(1, need); buf.d3 = dp_array(2, need);
    if (!((uint32_t)need > X.C.dpCap))
      wave_xdrop<int*, false>(q, qlen, d, dlen, 0, -1, -1, xdrop, buf, extCols, extRows, ncells);
  }
  X.cells += ncells;
  return ((unsigned long long)(uint32_t)uni(extRows) << 32) | (uint32_t)uni(extCols);
}

// ------------------------------------------------------------------ getSeedAndExtension (Trail.cpp:341-437)
struct SeedExt { int lenRefExt, lenHistExt, posOnRef, score; bool stop; int extRef, extCand; bool fallback; };

// growth-order restatement: the seed sits at the anchor end; extension starts at offset S
// (K-1 walking RIGHT: Seed(0,0,K-1,K-1); K walking LEFT: Seed(len-K, len-K, len-1, len-1)).
// WIDE: the instance for extensions whose x needs more than 255 diagonals (x beyond 127: the tail of a long edge, whose
// x grows by 2 per scoring, Explorer.cpp:713) — a function of its own, entered by a tail call, so that the registers of
// its eight-diagonals-per-lane phase (and the callee-saved ones it has to save) are paid by the few calls that get that
// far, not by every extension.
// MODE 0: the band fits one diagonal per lane (x up to 31: two thirds of all calls, and the instance the hot path's
// registers and instruction cache see); 1: phases of one, two and four diagonals per lane (up to 255 diagonals);
// 2: WIDE, with an eight-wide and a sixteen-wide phase behind them (up to 1023 diagonals: x to 511)
// LEAF: the instance the path search calls for its own Trails — it makes no call (what would need one, the anti-diagonal
// sweep or a score by alignment, is reported as `fallback` and the caller asks the general function), so it saves and
// restores nothing; and it takes the two anchors for equal, which they are by construction there (a Trail starts with
// the anchor its reference starts with): the general form compares them, one more round trip to memory per scoring.
template <int MODE, bool LEAF = false>
TALC_D SeedExt seed_and_extension_body(const uint8_t* ref, int refLen, const uint8_t* cand, int candLen, int xdrop,
                                       bool withScore, uint32_t keepKey = 0) {
  PROF_DECL;
  refLen = uni(refLen); candLen = uni(candLen); xdrop = uni(xdrop); ref = uni_ptr(ref); cand = uni_ptr(cand);
  const int K = (int)X.P.K;
  const int S = X.dirRight ? K - 1 : K;
  SeedExt r;
  r.stop = false; r.score = 0; r.fallback = false;
  // seq1 (database, H, rows) = the longer (reference unless strictly shorter), seq2 (query, V, cols)
  const bool state = !(refLen < candLen);
  const uint8_t* seq1 = state ? ref : cand; const int len1 = state ? refLen : candLen;
  const uint8_t* seq2 = state ? cand : ref; const int len2 = state ? candLen : refLen;
  int extCols = 0, extRows = 0, extScore = 0, rc = 0;
  unsigned long long ncells = 0;
  const int qlen = len2 - S, dlen = len1 - S;
  if (qlen > 0 && dlen > 0) {
    PROF_BEGIN();
#ifdef TALC_PROF
    if (lane_id() == 0) g_prof[PF_XCALLS] += 1;
#endif
    constexpr int STAGE = 3 * LDS_DP_CAP * 4;
    uint8_t TALC_AS3* stage = (uint8_t TALC_AS3*)g_dp;
    // furthest-reaching wavefronts, 1 / 2 / 4 diagonals per lane (x up to 31 / 63 / 127)
    const int ndiagonals = min(max(xdrop, 0), qlen) + min(max(xdrop, 0), dlen) + 1;
    if (MODE == 2) {
      // in phases (WfaPhase): levels 0..31 one diagonal per lane, 32..63 two, 64..127 four, the rest eight.  (A 500-base
      // edge ends near x = 210; without the eight-wide phase its last forty scorings per anchor fell back to the
      // anti-diagonal sweep — most of the 40 ms of the heaviest reads of a batch, which is what the launch waits for.)
      int* mem = X.dpG + 2ull * X.C.dpCap + 256;   // (past the flags of the multi-x run)
      WfaPhase ph{-1, 31, mem, mem + 512};
      rc = wave_xdrop_wfa<1>(seq2 + S, qlen, seq1 + S, dlen, xdrop, stage, STAGE, extCols, extRows, extScore, ncells, &ph);
      if (rc == 2) {
        ph.fromLevel = 31; ph.toLevel = 63;
        rc = wave_xdrop_wfa<2>(seq2 + S, qlen, seq1 + S, dlen, xdrop, stage, STAGE, extCols, extRows, extScore, ncells, &ph);
        if (rc == 2) {
          ph.fromLevel = 63; ph.toLevel = 127;
          rc = wave_xdrop_wfa<4>(seq2 + S, qlen, seq1 + S, dlen, xdrop, stage, STAGE, extCols, extRows, extScore, ncells, &ph);
          if (rc == 2) {
            ph.fromLevel = 127; ph.toLevel = (ndiagonals <= 511) ? -1 : 255;
            rc = wave_xdrop_wfa<8>(seq2 + S, qlen, seq1 + S, dlen, xdrop, stage, STAGE, extCols, extRows, extScore, ncells, &ph);
            if (rc == 2) {   // x beyond 255 (a Trail of a thousand steps and more: a start anchor deep inside a long solid region)
              ph.fromLevel = 255; ph.toLevel = -1;
              rc = wave_xdrop_wfa<16>(seq2 + S, qlen, seq1 + S, dlen, xdrop, stage, STAGE, extCols, extRows, extScore, ncells, &ph);
            }
          }
        }
      }
    }
    else if (MODE == 0 && ndiagonals <= 63) rc = wave_xdrop_wfa<1>(seq2 + S, qlen, seq1 + S, dlen, xdrop, stage, STAGE, extCols, extRows, extScore, ncells, nullptr, state ? keepKey : 0u);
    else if (MODE == 1 && ndiagonals <= 255) {   // (the hand-over state lives in the HBM DP arrays: make_caps keeps them at 2048 ints or more)
      // in phases (WfaPhase): levels 0..31 one diagonal per lane, 32..63 two, the rest four
      int* mem = X.dpG + 2ull * X.C.dpCap + 256;   // (past the flags of the multi-x run)
      WfaPhase ph{-1, 31, mem, mem + 512};
      rc = wave_xdrop_wfa<1>(seq2 + S, qlen, seq1 + S, dlen, xdrop, stage, STAGE, extCols, extRows, extScore, ncells, &ph, state ? keepKey : 0u);
      if (rc == 2) {
        ph.fromLevel = 31; ph.toLevel = (ndiagonals <= 127) ? -1 : 63;
        rc = wave_xdrop_wfa<2>(seq2 + S, qlen, seq1 + S, dlen, xdrop, stage, STAGE, extCols, extRows, extScore, ncells, &ph);
        if (rc == 2) {
          ph.fromLevel = 63; ph.toLevel = -1;
          rc = wave_xdrop_wfa<4>(seq2 + S, qlen, seq1 + S, dlen, xdrop, stage, STAGE, extCols, extRows, extScore, ncells, &ph);
        }
      }
    }
    else rc = -1;
    if (LEAF && rc < 0) { r.fallback = true; return r; }
    if (rc < 0) {   // band wider than the wavefront routines take: the anti-diagonal sweep
      rc = -1;
      const unsigned long long packed = xdrop_sweep(seq2 + S, qlen, seq1 + S, dlen, xdrop);
      extCols = (int)(uint32_t)packed; extRows = (int)(uint32_t)(packed >> 32);
    }
    PROF_END(PF_XDROP);
    X.cells += ncells;
  }
  const int ext1 = extRows, ext2 = extCols;            // on seq1 / seq2
  r.extRef = state ? ext1 : ext2;
  r.extCand = state ? ext2 : ext1;
  r.lenRefExt = S + r.extRef;
  r.lenHistExt = S + r.extCand;
  r.posOnRef = X.dirRight ? (S + r.extRef) : (refLen - K - r.extRef);
  if (max(r.lenRefExt, r.lenHistExt) >= K) {
    // Trail.cpp:408-434 scores the two extensions (anchor included) by a global alignment (0,-1,-1) = minus their edit
    // distance.  Both start with the same anchor, so that is the edit distance of the extended segments, and the
    // x-drop value of the cell it stopped on is exactly that (a defined cell holds the best path from the origin;
    // a path that leaves the x-drop region costs more than x, the cell's own cost is at most x).
    if (withScore) {
      bool sameAnchor = LEAF && rc == 1;
      if (LEAF && !sameAnchor) { r.fallback = true; return r; }
      if (!LEAF && rc == 1) {   // (the reference does not require equal anchors; every caller on the correction path has them)
        const int l = lane_id();
        sameAnchor = ballot64(l < S && ((gcu8)ref)[l < S ? l : 0] != ((gcu8)cand)[l < S ? l : 0]) == 0ull;
      }
      if (sameAnchor) r.score = extScore;
      else { PROF_BEGIN(); r.score = nw_score(ref, r.lenRefExt, cand, r.lenHistExt, 0, -1, -1, false); PROF_END(PF_EXTNW); }
    }
  } else {
    r.score = (-1) * xdrop;
    r.stop = true;
  }
  return r;
}

TALC_DNC SeedExt seed_and_extension_wide(const uint8_t* ref, int refLen, const uint8_t* cand, int candLen, int xdrop, bool withScore) {
  return seed_and_extension_body<2>(ref, refLen, cand, candLen, xdrop, withScore);
}
TALC_DN SeedExt seed_and_extension_mid(const uint8_t* ref, int refLen, const uint8_t* cand, int candLen, int xdrop, bool withScore) {
  return seed_and_extension_body<1>(ref, refLen, cand, candLen, xdrop, withScore);
}
// the leaf instances: bands of at most 63 diagonals, and (entered by a tail call) of 64 to 255; anything else: fallback
TALC_DN SeedExt seed_and_extension_mid_leaf(const uint8_t* ref, int refLen, const uint8_t* cand, int candLen, int xdrop, uint32_t keepKey) {
  return seed_and_extension_body<1, true>(ref, refLen, cand, candLen, xdrop, true, keepKey);
}
TALC_DN SeedExt seed_and_extension_leaf(const uint8_t* ref, int refLen, const uint8_t* cand, int candLen, int xdrop, uint32_t keepKey) {
  {
    const int rl = uni(refLen), cl = uni(candLen), x = max(uni(xdrop), 0);
    const int S = uni(X.dirRight) ? (int)X.P.K - 1 : (int)X.P.K;
    const int qlen = min(rl, cl) - S, dlen = max(rl, cl) - S;
    if (qlen > 0 && dlen > 0) {
      const int nd = min(x, qlen) + min(x, dlen) + 1;
      if (nd > 63 && nd <= 255) [[clang::musttail]] return seed_and_extension_mid_leaf(ref, refLen, cand, candLen, xdrop, keepKey);
    }
  }
  return seed_and_extension_body<0, true>(ref, refLen, cand, candLen, xdrop, true, keepKey);
}
TALC_DN SeedExt seed_and_extension(const uint8_t* ref, int refLen, const uint8_t* cand, int candLen, int xdrop, bool withScore) {
  {   // more than 255 diagonals (and a stage that takes the segments): the wide instance
    const int rl = uni(refLen), cl = uni(candLen), x = max(uni(xdrop), 0);
    const int S = uni(X.dirRight) ? (int)X.P.K - 1 : (int)X.P.K;
    const int qlen = min(rl, cl) - S, dlen = max(rl, cl) - S;
    if (qlen > 0 && dlen > 0) {
      const int nd = min(x, qlen) + min(x, dlen) + 1;
      if (nd > 255 && nd <= 1023) [[clang::musttail]] return seed_and_extension_wide(ref, refLen, cand, candLen, xdrop, withScore);
      if (nd > 63 && nd <= 255) [[clang::musttail]] return seed_and_extension_mid(ref, refLen, cand, candLen, xdrop, withScore);
    }
  }
  return seed_and_extension_body<0>(ref, refLen, cand, candLen, xdrop, withScore);
}

// getSeedAndExtension without the score (the form findStopPosition uses) from a given extension (extCols on the
// query = the shorter sequence, extRows on the other)
TALC_D SeedExt seedext_plain(int refLen, int candLen, int extCols, int extRows, int xdrop) {
  const int K = uni((int)X.P.K);
  const int S = uni(X.dirRight) ? K - 1 : K;
  const bool state = !(refLen < candLen);
  const int qlen = (state ? candLen : refLen) - S, dlen = (state ? refLen : candLen) - S;
  if (!(qlen > 0 && dlen > 0)) { extCols = 0; extRows = 0; }
  SeedExt r;
  r.stop = false; r.score = 0;
  r.extRef = state ? extRows : extCols;
  r.extCand = state ? extCols : extRows;
  r.lenRefExt = S + r.extRef;
  r.lenHistExt = S + r.extCand;
  r.posOnRef = X.dirRight ? (S + r.extRef) : (refLen - K - r.extRef);
  if (!(max(r.lenRefExt, r.lenHistExt) >= K)) { r.score = (-1) * xdrop; r.stop = true; }
  return r;
}

// the extensions of getSeedAndExtension for every x-drop in [0, xHi] from one wavefront run (wave_xdrop_wfa_multi):
// resCols[x], resRows[x] as wave_xdrop_wfa(x) would report them.  false = not available (ask x by x).
TALC_DN bool seed_and_extension_multi(const uint8_t* ref, int refLen, const uint8_t* cand, int candLen, int xHi, int* resCols,
                                      int* resRows, int* resScore) {
  PROF_DECL;
  refLen = uni(refLen); candLen = uni(candLen); xHi = uni(xHi); ref = uni_ptr(ref); cand = uni_ptr(cand);
  resCols = (int*)uni_ptr(resCols); resRows = (int*)uni_ptr(resRows); resScore = (int*)uni_ptr(resScore);
  const int K = (int)X.P.K;
  const int S = X.dirRight ? K - 1 : K;
  const bool state = !(refLen < candLen);
  const uint8_t* seq1 = state ? ref : cand; const int len1 = state ? refLen : candLen;
  const uint8_t* seq2 = state ? cand : ref; const int len2 = state ? candLen : refLen;
  const int qlen = len2 - S, dlen = len1 - S;
  if (!(qlen > 0 && dlen > 0) || xHi < 0) return false;
  PROF_BEGIN();
  constexpr int STAGE = 3 * LDS_DP_CAP * 4;
  uint8_t TALC_AS3* stage = (uint8_t TALC_AS3*)g_dp;
  unsigned long long ncells = 0;
  const int ndiagonals = min(xHi, qlen) + min(xHi, dlen) + 1;
  int rc = -1;
  if (ndiagonals <= 63) rc = wave_xdrop_wfa_multi<1>(seq2 + S, qlen, seq1 + S, dlen, xHi, stage, STAGE, resCols, resRows, resScore, ncells);
  else if (ndiagonals <= 127) rc = wave_xdrop_wfa_multi<2>(seq2 + S, qlen, seq1 + S, dlen, xHi, stage, STAGE, resCols, resRows, resScore, ncells);
  PROF_END(PF_XDROP);
  X.cells += ncells;
  WSYNC();   // lane 0's results are read by every lane
  return rc >= 0;
}

// Trail::seedAndExtend (Trail.cpp:193-216) on slot t of set S; returns `ok`
// (Everything here is wave-uniform: the arguments and the extension's results are moved to scalar registers and the
// Trail's record is read again after the extension instead of being held across it, so that this function keeps no
// vector register alive over the call — a vector register held over a call has to be a callee-saved one, and every
// callee-saved register a function touches costs a scratch store and a load per call of it.)
TALC_D bool trail_seed_and_extend(int set_, int t_, int len_, int xdrop_) {
  const int set = uni(set_), t = uni(t_), len = uni(len_), xdrop = uni(xdrop_);
  WSYNC();   // the Trail's last bases were appended by lane 0: make them visible to the DP lanes
  const uint32_t buf = (uint32_t)uni((int)tr_buf(set, t));
  // (the kept wavefront, talc_wave.h: the pair is this search's reference and the Trail in buffer `buf`)
  SeedExt e = seed_and_extension_leaf(X.ref, (int)X.refLen, X.seqPool + (uint64_t)buf * X.C.seqCap, len, xdrop, buf + 1u);
  if (uni((int)e.fallback) != 0) e = seed_and_extension(X.ref, (int)X.refLen, X.seqPool + (uint64_t)buf * X.C.seqCap, len, xdrop, true);
  const int lenHistExt = uni(e.lenHistExt), score = uni(e.score), posOnRef = uni(e.posOnRef);
  const bool stop = uni((int)e.stop) != 0;
  TrailRec r = tr_get(set, t);
  const bool ok1 = (lenHistExt == len);
  r.fail = ok1 ? 0u : r.fail + 1u;
  r.score = score;
  if (X.dirRight) r.ranc = posOnRef; else r.lanc = posOnRef;
  const bool ok = (uni((int)r.fail) <= (uint32_t)X.P.MAX_BORDER_FAILURES) & !stop;
  if (lane_id() == 0) tr_put(set, t, r);
  tr_sync(t);
  return ok;
}

// ------------------------------------------------------------------ recordEdge (Explorer.cpp:1103-1118)
// Trajectory(trail) + trim + reshape + cutAnchors, then the fold of findBestBORDER
// (Trajectory.cpp:306-334) into the best long / best short candidate.
TALC_DN void record_edge(int set_, int t_, int len0_) {
  PROF_DECL;
  // (wave-uniform throughout: every value that outlives a call sits in a scalar register)
  const int set = uni(set_), t = uni(t_), len0 = uni(len0_);
  const int K = (int)X.P.K;
  WSYNC();
  uint32_t trBuf, trFail, lanc, ranc; int lastScore; unsigned long long distBits;
  {
    const TrailRec tr = tr_get(set, t);
    trBuf = (uint32_t)uni((int)tr.buf); trFail = (uint32_t)uni((int)tr.fail); lastScore = uni(tr.score);
    lanc = (uint32_t)uni(tr.lanc); ranc = (uint32_t)uni(tr.ranc);
    distBits = uni64((unsigned long long)__double_as_longlong(tr.dist / ((double)len0 + 0.01)));   // Trajectory.cpp:45
  }
  const uint8_t* path = X.seqPool + (uint64_t)trBuf * X.C.seqCap;
  // trim (Trajectory.cpp:89-112)
  int len = len0;
  const uint32_t nbBases = trFail * X.P.CHECK_INTERVAL;
  if ((uint32_t)len >= nbBases + (uint32_t)K) len = len - (int)nbBases;
  const int refLen = uni((int)X.refLen);
  const bool shorter = (len <= refLen);
  // reshape (Trajectory.cpp:114-155) with findStopPosition (:482-503)
  int xdrop1 = (int)lastScore * (-1);
  SeedExt cur, nxt;
  // findStopPosition(A, B): `reference` = A, `shorterPath` = B
  const uint8_t* A = uni_ptr(shorter ? X.ref : path); const int lenA = shorter ? refLen : len;
  const uint8_t* Bq = uni_ptr(shorter ? path : X.ref); const int lenB = shorter ? len : refLen;
  // the loop below asks for x, x-1, x-2, ... : all of them come out of one wavefront run when the band fits
  int* const resCols = uni_ptr(X.dpG);
  int* const resRows = resCols + X.C.dpCap;
  int* const resScore = resCols + 2ull * X.C.dpCap;   // ([0, 128): below the phased x-drop's hand-over area)
  bool multi = false;
  if (xdrop1 >= 1 && (uint32_t)(xdrop1 + 1) <= X.C.dpCap && xdrop1 < 128)
    multi = seed_and_extension_multi(A, lenA, Bq, lenB, xdrop1, resCols, resRows, resScore);
  auto scalar = [](SeedExt e) -> SeedExt {
    e.lenRefExt = uni(e.lenRefExt); e.lenHistExt = uni(e.lenHistExt); e.posOnRef = uni(e.posOnRef); e.score = uni(e.score);
    e.stop = uni((int)e.stop) != 0; e.extRef = uni(e.extRef); e.extCand = uni(e.extCand);
    return e;
  };
  auto ext_at = [&](int x) -> SeedExt {
    if (multi && x >= 0) return seedext_plain(lenA, lenB, uni(resCols[x]), uni(resRows[x]), x);
    return scalar(seed_and_extension(A, lenA, Bq, lenB, x, false));
  };
  nxt = ext_at(xdrop1);
  bool goFurther = true;
  do {
    --xdrop1;
    cur = nxt;
    nxt = ext_at(xdrop1);
    if (nxt.lenHistExt < cur.lenHistExt) goFurther = false;
  } while (goFurther & (xdrop1 > 0));
  // `cur` is the extension for x = xdrop1 + 1.  Its score (Trail.cpp:408-434: minus the edit distance of the two
  // extensions) is the cost of the cell the run reported, when both start with the same anchor (seed_and_extension)
  bool haveScore = false;
  int multiScore = 0;
  if (multi && !cur.stop) {
    const int sc = uni(resScore[xdrop1 + 1]);
    const int S = X.dirRight ? K - 1 : K;
    const int l = lane_id();
    if (sc <= 0 && ballot64(l < S && ((gcu8)A)[l < S ? l : 0] != ((gcu8)Bq)[l < S ? l : 0]) == 0ull) { haveScore = true; multiScore = sc; }
  }
  // score of the retained extension (Trail.cpp:408-434)
  PROF_BEGIN();
  int scoreI;        // (the score is an integer in every branch: kept as one, compared as a double below)
  int lcs;
  {
    // score of the retained extension (Trail.cpp:408-434) and computePercentID (Trajectory.cpp:505-528):
    // -edit distance and LCS / max length of the same two extensions
    int es;
    edit_and_lcs(A, cur.lenRefExt, Bq, cur.lenHistExt, es, lcs, !cur.stop && !haveScore, 0);
    es = uni(es); lcs = uni(lcs);
    scoreI = cur.stop ? cur.score : (haveScore ? multiScore : es);
  }
  const double score = (double)scoreI;
  const double idscore = (double)lcs / (double)max(cur.lenRefExt, cur.lenHistExt);
  const double dist = __longlong_as_double((long long)distBits);
  if (dist != dist) X.nanSeen = 1u;
  // new sequence in growth order
  uint8_t* tmp = uni_ptr(X.edgeTmp);
  uint32_t newLen;
  if (!shorter) {
    // prefix(path, pos) walking RIGHT / suffix(path, pos) walking LEFT == growth-order prefix of length S+ext(path)
    newLen = (uint32_t)cur.lenRefExt;   // `reference` of findStopPosition is the path here
    if (newLen > X.C.edgeCap) { X.overflow |= OVF_SEQ; return; }
    copy_bytes(tmp, path, newLen, false);
  } else {
    // path followed by the rest of the reference beyond the stop position
    const uint32_t from = (uint32_t)cur.lenRefExt;
    const uint32_t rest = (uint32_t)refLen > from ? (uint32_t)refLen - from : 0;
    newLen = (uint32_t)len + rest;
    if (newLen > X.C.edgeCap) { X.overflow |= OVF_SEQ; return; }
    copy_bytes(tmp, path, (uint32_t)len, false);
    copy_bytes(tmp + len, X.ref + from, rest, false);
  }
  WSYNC();
  // cutAnchors HEAD/TAIL (Trajectory.cpp:168-175): drop the anchor (growth-order front); never fails
  uint32_t cutLen = 0, cutFrom = 0;
  if (newLen > (uint32_t)K) { cutLen = newLen - (uint32_t)K; cutFrom = (uint32_t)K; }
  X.nEdges++;
  // (no references / pointers into the LDS context are formed conditionally: copy in, copy out)
  const int bi = shorter ? 1 : 0;
  EdgeCand best = X.best2[bi];
  uint8_t* bestSeq = shorter ? X.edgeShort : X.edgeLong;
  bool take = false;
  if (!best.have) take = true;
  else if (score > best.score) take = true;
  else if (score == best.score && dist > best.dist) take = true;
  if (uni((int)take)) {
    best.have = true; best.score = score; best.dist = dist; best.idscore = idscore; best.len = cutLen;
    best.lanc = lanc; best.ranc = ranc;
    X.best2[bi] = best;
    copy_bytes(bestSeq, tmp + cutFrom, cutLen, false);
    WSYNC();
  }
  PROF_END(PF_EDGEMISC);
}

// ------------------------------------------------------------------ one expansion step
// Shared front half of oneMoreStep / oneMoreStepInTheDark: probe the table for the successors of
// every Trail of the current set (one lane per Trail, 64 at a time) and tag them (tagNextNodes).
struct StepTags { int tags; uint32_t nc[4]; };   // (a child's distance term is computed when the child is made: four
                                                 //  doubles per lane less to carry through the step's calls)

// isExpectedbyMyModel's thresholds for tagNextNodes of a tip with count `count` (talc_common.h: DevParams.thr); only
// meaningful when both the count and its lambda_noise are below thrN (inTable)
TALC_D ModelThresholds model_thresholds(uint32_t count, bool& inTable) {
  ModelThresholds m;
  const uint32_t TALC_AS1* thr = (const uint32_t TALC_AS1*)X.P.thr;
  const uint32_t lnRaw = lambda_noise_of(count, X.P.ERR);
  inTable = (count < X.P.thrN) & (lnRaw < X.P.thrN);   // (lambda_noise exceeds the count when --SR_ERROR_RATE is above 1)
  const uint32_t c = min(count, X.P.thrN - 1u), ln = min(lnRaw, X.P.thrN - 1u);
  m.minExpected = thr[2u * c];
  m.belowUnexpected = thr[2u * ln + 1u];
  return m;
}

TALC_D StepTags probe_and_tag(int t, bool valid, bool complex) {
  StepTags r;
  uint32_t cnt[4] = {0, 0, 0, 0}, jc[4] = {0, 0, 0, 0};
  int tg[4] = {TAG_NONE, TAG_NONE, TAG_NONE, TAG_NONE};
  double ds[4] = {0, 0, 0, 0};
  if (valid) {
    uint64_t km, nm; uint32_t lc;
    if (t < HOT) { const TrailRec& h = g_hot[X.ia * HOT + t]; km = h.kmer; nm = h.nmask; lc = h.cnt; }
    else { const TrailRec h = tr_get_slow(X.ia, t); km = h.kmer; nm = h.nmask; lc = h.cnt; }
    const uint32_t K = X.P.K;
    const uint64_t succN = X.dirRight ? (nm >> 1) : (nm & ((1ULL << (K - 1)) - 1));
    // the count model's two thresholds for this Trail's count are requested before the bucket, and arrive with it
    bool inTable;
    const ModelThresholds mt = model_thresholds(lc, inTable);
    if (!succN) dev_next_counts(X.T, km, X.dirRight, cnt, jc);
    if (inTable) tag_next_nodes_with(mt, X.P.ERR, X.P.MIN_COUNT, cnt, jc, lc, complex, tg, (double*)nullptr);
    else tag_next_nodes(X.P.ALPHA, X.P.ERR, X.P.MIN_COUNT, cnt, jc, lc, complex, tg, ds);
  }
  r.tags = (tg[0] & 0xff) | ((tg[1] & 0xff) << 8) | ((tg[2] & 0xff) << 16) | ((tg[3] & 0xff) << 24);
#pragma unroll
  for (int b = 0; b < 4; ++b) r.nc[b] = cnt[b];
  return r;
}
TALC_D double shfl_f64(double v, int src) {   // src is wave-uniform
  long long x = __double_as_longlong(v);
  int lo = lane_get((int)(x & 0xffffffffll), src), hi = lane_get((int)(x >> 32), src);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// create child `c` of the new set from Trail `t` of the current set with base `b`; sequences have
// length len -> len+1.  inherit: the child takes over the parent's sequence buffer (it is the
// parent's last successor).  Returns the child's tip k-mer (wave-uniform).
// a Trail that is not its parent's last successor: a buffer of its own with a copy of the parent's `len` bases (and, in a
// bridge search, of the parent's kept alignment row).  A real call: branching is the rare case of the step.
// A Trail's first len bases from buffer src to buffer dst, and (withRow: bridge searches) its kept alignment row with them
// (row_copy's rule: row and stamps as they are).  Both are requested before either is stored: one memory round trip per
// copied Trail instead of two.  The caller has synchronised since either was written.
TALC_D void copy_trail(uint32_t dst, uint32_t src, int len, bool withRow) {
  const uint32_t avail = (uint32_t)uni((int)X.rowAvail);
  if (withRow && dst < avail && src < avail) {
    gcu8 s1 = (gcu8)uni_ptr(X.seqPool + (uint64_t)src * X.C.seqCap);
    gu8 d1 = (gu8)uni_ptr(X.seqPool + (uint64_t)dst * X.C.seqCap);
    const v4u32 TALC_AS1* s2 = (const v4u32 TALC_AS1*)uni_ptr(row_of(src));
    v4u32 TALC_AS1* d2 = (v4u32 TALC_AS1*)uni_ptr(row_of(dst));
    const uint32_t l = (uint32_t)lane_id();
    const uint32_t n1 = (uint32_t)len, nv1 = n1 >> 4, nv2 = (uint32_t)uni((int)X.rowStride) >> 2;   // 16-byte vectors
    const v4u32 TALC_AS1* s1v = (const v4u32 TALC_AS1*)s1;
    v4u32 TALC_AS1* d1v = (v4u32 TALC_AS1*)d1;
    const v4u32 zero = {0u, 0u, 0u, 0u};
    const v4u32 a0 = (l < nv1) ? s1v[l] : zero;
    const uint32_t tb = (nv1 << 4) + l;
    const uint8_t at = (tb < n1) ? s1[tb] : (uint8_t)0;
    const v4u32 b0 = (l < nv2) ? s2[l] : zero, b1 = (l + 64u < nv2) ? s2[l + 64u] : zero;
    if (l < nv1) d1v[l] = a0;
    if (tb < n1) d1[tb] = at;
    if (l < nv2) d2[l] = b0;
    if (l + 64u < nv2) d2[l + 64u] = b1;
    for (uint32_t i = l + 64u; i < nv1; i += 64u) d1v[i] = s1v[i];      // (Trails beyond 1 kb, references beyond 510 bases)
    for (uint32_t i = l + 128u; i < nv2; i += 64u) d2[i] = s2[i];
    return;
  }
  wave_copy(X.seqPool + (uint64_t)dst * X.C.seqCap, X.seqPool + (uint64_t)src * X.C.seqCap, (uint32_t)len);
  if (withRow) row_copy(dst, src);
}
TALC_DNC uint32_t branch_copy(uint32_t parentBuf_, int len_, bool bridge_) {
  const uint32_t parentBuf = (uint32_t)uni((int)parentBuf_);
  const int len = uni(len_);
  const uint32_t cbuf = (uint32_t)pool_alloc();
  if (lane_id() == 0) g_keep.owner = 0u;   // (a buffer changes hands: a kept wavefront may be about its former contents)
  WSYNC();   // bases appended by lane 0 in earlier steps must be visible to the copying lanes
  copy_trail(cbuf, parentBuf, len, uni((int)bridge_) != 0);
  return cbuf;
}

// BRIDGE: the step of a bridge search (a copied Trail takes its kept alignment row along; an instance of its own, so
// that the edge step carries neither the call nor the registers held across it)
template <bool BRIDGE>
TALC_D void make_child(int t, int c, int b, int len, uint32_t count, bool inherit, uint64_t& km2, uint64_t& nm2) {
  const uint32_t K = X.P.K;
  const uint64_t kmask = (1ULL << (2 * K)) - 1;
  const TrailRec p = tr_get(X.ia, t);
  uint32_t cbuf = p.buf;
  if (!inherit) cbuf = (uint32_t)uni((int)branch_copy(p.buf, len, BRIDGE));
  if (X.dirRight) { km2 = ((p.kmer << 2) | (uint64_t)b) & kmask; nm2 = p.nmask >> 1; }
  else { km2 = ((uint64_t)b << (2 * (K - 1))) | (p.kmer >> 2); nm2 = (p.nmask << 1) & ((1ULL << K) - 1); }
  if (lane_id() == 0) {
    ((gu8)X.seqPool)[(uint64_t)cbuf * X.C.seqCap + len] = (uint8_t)b;
    TrailRec ch;
    // (tagNextNodes' distance term of this successor, talc_pure.h: |count of the tip - count of the successor| / sqrt(count of the tip))
    ch.kmer = km2; ch.nmask = nm2; ch.cnt = count; ch.score = p.score; ch.fail = p.fail;
    ch.dist = p.dist + fabs((double)p.cnt - (double)count) / sqrt((double)p.cnt);
    ch.lanc = p.lanc; ch.ranc = p.ranc; ch.buf = cbuf;
    tr_put(X.ia ^ 1, c, ch);
  }
  tr_sync(c);
}

// Trail::ThinkIveAlreadyGotThere (Trail.cpp:289-302) for child c (length len+1, tip km2/nm2)
// against its parent t
// ... the exact window search behind the filter: a real, cold call (the filter answers "nowhere yet" for 99 % of the
// generic steps of a unique-sequence graph, and the search's body does not belong in the step's loop)
TALC_DNC bool is_cycle_exact(int t_, int c_, int len_) {
  const int t = uni(t_), c = uni(c_), len = uni(len_);
  const int K = (int)X.P.K;
  WSYNC();
  const uint8_t* parent = trail_seq(X.ia, t);
  const uint8_t* child = trail_seq(X.ia ^ 1, c);
  const uint8_t* pat = child + (len + 1 - K);
  if (X.dirRight) {
    const int p = wave_find_window(parent, len, pat, K, false);
    return p > 0;
  }
  // walking LEFT the text is reversed: the reference's first occurrence is our last one, and its
  // position 0 is the parent's tip (growth index len-K)
  const int q = wave_find_window(parent, len, pat, K, true);
  return q >= 0 && q != len - K;
}
TALC_D bool is_cycle(int t, int c, int len, uint64_t km2, uint64_t nm2, uint32_t& hv) {
  const int K = (int)X.P.K;
  const bool maybe = bloom_query(km2, nm2, hv);
#ifdef TALC_PROF
  if (lane_id() == 0) { g_prof[PF_CYQ] += 1; if (maybe && len > K) g_prof[PF_CYX] += 1; }
  if (maybe && len > K && X.wideMask == 0u) {   // how full the LDS filter is when it says "maybe" (percent, summed)
    int pc = __popcll(g_bloom[lane_id()]) + __popcll(g_bloom[lane_id() + 64]);
    for (int o = 32; o > 0; o >>= 1) pc += __shfl_xor(pc, o);
    if (lane_id() == 0) g_prof[PF_CYFILL] += (uint32_t)(pc * 100 / 8192);
  }
#endif
  if (!(len > K)) return false;
  if (!maybe) return false;   // the k-mer occurs nowhere in this search so far
  const bool found = uni((int)is_cycle_exact(t, c, len)) != 0;
#ifdef TALC_PROF
  if (lane_id() == 0 && found) g_prof[PF_CYHIT] += 1;
#endif
  return found;
}
TALC_D void swap_sets() { X.ia ^= 1; }

// doABitOfGardening on the new set (n trails); survivors are copied into the other set, which
// becomes the current one; returns their number
TALC_DNC int garden(int n, int len, bool& isComplex) {
  __shared__ int s_nk;
  __shared__ int s_cx;
  const int l = lane_id();
  const int ib = X.ia ^ 1;   // the Trails to rank
  if (l == 0) g_keep.owner = 0u;   // (the survivors move to fresh buffers)
  WSYNC();
  for (int i = l; i < n; i += 64) { const TrailRec r = tr_get(ib, i); X.gScores[i] = (double)r.score; X.gDists[i] = r.dist; }
  WSYNC();
  if (l == 0) {
    bool cx = false;
    s_nk = gardening(X.P.MAXB, n, X.gScores, X.gDists, X.gVal, X.gRank, X.gRank + (TCAP + 64), X.gKept, &cx);
    s_cx = cx ? 1 : 0;
  }
  WSYNC();
  int nk = s_nk;
  isComplex = s_cx != 0;
  if (nk > TCAP) { X.overflow |= OVF_TRAILS; nk = TCAP; }
  for (int i = 0; i < nk; ++i) {
    const uint32_t src = X.gKept[i];
    TrailRec r = tr_get(ib, (int)src);
    const uint32_t nb = (uint32_t)pool_alloc();
    copy_trail(nb, (uint32_t)uni((int)r.buf), len, uni((int)X.location) == LOC_INNER);
    r.buf = nb;
    if (l == 0) tr_put(X.ia, i, r);
  }
  WSYNC();
  for (int i = 0; i < n; ++i) pool_free(tr_buf(ib, i));
  return nk;
}

TALC_D int last_successor(int tags) {
  int lastI = -1;
#pragma unroll
  for (int i = 0; i < 4; ++i) { const int tg = (int)(int8_t)((tags >> (8 * i)) & 0xff); if (tg != TAG_NONE && tg != TAG_UNEXPECTED) lastI = i; }
  return lastI;
}

// scoreBridges (Explorer.cpp:689-706): Trail::Overlapscore (Trail.cpp:145-173) of every Trail of the new set against the
// reference truncated to K+step+WINDOW (growth-order prefix), each from the row the Trail (or the Trail it was copied
// from) left at its last scoring.  One instance per width of the reference (NB columns per lane); functions of their own:
// they only run in complex regions, and the step's common path should not carry their registers.
// What a scoring waits for is memory, not arithmetic (six new rows are ~ 500 instructions): the reference's bases are
// fetched once for all Trails, the Trails' buffers and the rows their records cover are looked up one Trail per lane
// (one round trip for 64 Trails), no Trail waits for the stores of the one before, and only the score goes back into the
// Trail's record.
template <int NB>
TALC_DNC void score_bridges_rows(int ib_, int nNew_, int m_, int tlen_, uint32_t rowAvail_) {
  const int ib = uni(ib_), nNew = uni(nNew_), m = uni(m_), tlen = uni(tlen_);
  const uint32_t rowAvail = (uint32_t)uni((int)rowAvail_);
  const int n = (int)uni((int)X.refLen);
  const int l = lane_id();
  unsigned hp[(NB + 3) / 4];
  nw_rows_cols<NB>(X.ref, n, hp);
  const uint32_t TALC_AS1* bufG = (const uint32_t TALC_AS1*)uni_ptr(X.G[ib].buf);
  int TALC_AS1* scoreG = (int TALC_AS1*)uni_ptr(X.G[ib].score);
  const uint32_t stamp = (uint32_t)uni((int)X.launchStamp), sno = (uint32_t)uni((int)X.searchNo);
  unsigned long long ncells = 0;
  for (int base = 0; base < nNew; base += 64) {
    const int tj = base + l;
    uint32_t myBuf = 0xFFFFFFFFu, myCov = 0u;
    if (tj < nNew) {
      if (tj < HOT) myBuf = g_hot[ib * HOT + tj].buf;
      else myBuf = bufG[tj];
      if (myBuf < rowAvail) {   // row_covered, one Trail per lane
        const int TALC_AS1* rec = (const int TALC_AS1*)row_of(myBuf);
        const uint32_t lo = (uint32_t)rec[0], hi = (uint32_t)rec[n + 1];
        myCov = (hi == stamp && (lo >> ROW_COV_BITS) == sno) ? (lo & ((1u << ROW_COV_BITS) - 1u)) : 0u;
      }
    }
    const int cnt = min(64, nNew - base);
    for (int jj = 0; jj < cnt; ++jj) {
      const uint32_t rb = (uint32_t)__builtin_amdgcn_readlane((int)myBuf, jj);
      const uint8_t* cand = X.seqPool + (uint64_t)rb * X.C.seqCap;
      int sc;
      if (rb < rowAvail) {
        int i0 = __builtin_amdgcn_readlane((int)myCov, jj);
        if (i0 > m) i0 = 0;   // (cannot happen: a Trail only grows)
        sc = nw_rows_run<NB>(hp, n, cand, i0, m, 4, -3, -2, row_of(rb), tlen, ncells);
        row_set_covered(rb, (uint32_t)m);
      } else {
        sc = nw_score(X.ref, tlen, cand, m, 4, -3, -2, true);
      }
      const int j = base + jj;
      if (l == 0) { if (j < HOT) g_hot[ib * HOT + j].score = sc; else scoreG[j] = sc; }
    }
  }
  X.cells += ncells;
  WSYNC();
}
static_assert(ROW_MAX_REF <= 64 * 128 - 1 && ROW_MAX_REF < (1 << ROW_COV_BITS), "the widest instance takes 128 columns per lane");

TALC_DNC void score_bridges(int ib_, int nNew_, int len_, uint32_t stepCounter_) {
  const DevParams& P = X.P;
  const int ib = uni(ib_), nNew = uni(nNew_), len = uni(len_);
  const uint32_t stepCounter = (uint32_t)uni((int)stepCounter_);
  const uint32_t bound = P.K + stepCounter + P.WINDOW;
  const int tlen = (int)min(bound, X.refLen);
  WSYNC();   // rows and stamps may have arrived by a copy, the Trails' last bases by lane 0
  const uint32_t rowAvail = ((uint32_t)(len + 1) < (1u << ROW_COV_BITS)) ? (uint32_t)uni((int)X.rowAvail) : 0u;
  const int B = ((int)uni((int)X.refLen) + 63) >> 6;
  if (rowAvail == 0u || B <= 2) score_bridges_rows<2>(ib, nNew, len + 1, tlen, rowAvail);   // (no rows: every Trail from scratch, nw_score)
  else if (B <= 4) score_bridges_rows<4>(ib, nNew, len + 1, tlen, rowAvail);
  else if (B <= 6) score_bridges_rows<6>(ib, nNew, len + 1, tlen, rowAvail);
  else if (B <= 8) score_bridges_rows<8>(ib, nNew, len + 1, tlen, rowAvail);
  else if (B <= 12) score_bridges_rows<12>(ib, nNew, len + 1, tlen, rowAvail);
  else if (B <= 16) score_bridges_rows<16>(ib, nNew, len + 1, tlen, rowAvail);
  else if (B <= 24) score_bridges_rows<24>(ib, nNew, len + 1, tlen, rowAvail);
  else if (B <= 32) score_bridges_rows<32>(ib, nNew, len + 1, tlen, rowAvail);
  // (references of 2048-8191 bases: the row no longer fits the registers and these instances spill — still a thousand
  //  times less work than aligning every Trail of a 2 kb gap from scratch at every scoring, which is what made ONE walk of
  //  stress set 105 take 40 s)
  else if (B <= 48) score_bridges_rows<48>(ib, nNew, len + 1, tlen, rowAvail);
  else if (B <= 64) score_bridges_rows<64>(ib, nNew, len + 1, tlen, rowAvail);
  else if (B <= 96) score_bridges_rows<96>(ib, nNew, len + 1, tlen, rowAvail);
  else score_bridges_rows<128>(ib, nNew, len + 1, tlen, rowAvail);
}

// A child of the generic step whose tip is an aim (checkAims, Trail.cpp:273-285): recordBridge (Explorer.cpp:1097-1101).
// Returns false when the recorded path is longer than the reference and the Trail ends there (:579-582).  A real, cold
// call: nearly every step onto an aim is taken by the fast-forward, and this body sat in the middle of the step's loop.
TALC_DNC bool record_bridge_at_aim(int nNew_, int hit_, int len_, uint64_t km2, uint64_t nm2) {
  const int nNew = uni(nNew_), hit = uni(hit_), len = uni(len_);
  const int l = lane_id();
  const int ib = X.ia ^ 1;
  const AnchorRec* aims = X.dirRight ? X.ancR : X.ancL;
  const int apos = (hit < AIMS_LDS) ? (int)g_aimPos[hit] : (int)aims[hit].pos;
  TrailRec ch = tr_get(ib, nNew);
  if (X.dirRight) ch.ranc = apos; else ch.lanc = apos;
  if (l == 0) tr_put(ib, nNew, ch);
  WSYNC();   // also publishes the base lane 0 has just appended
  const uint32_t clen = (uint32_t)len + 1;
  if (X.nFull >= (int)X.C.fullCap) X.overflow |= OVF_FULLPATHS;
  else if (X.fullUsed + clen > X.C.fullPool) X.overflow |= OVF_FULLPOOL;
  else {
    copy_bytes(X.fullPool + X.fullUsed, X.seqPool + (uint64_t)ch.buf * X.C.seqCap, clen, false);
    if (l == 0) X.fullMeta[X.nFull] = FullMeta{X.fullUsed, clen, ch.lanc, ch.ranc, ch.dist / ((double)clen + 0.01)};
    X.fullUsed += (clen + 15u) & ~15u;
    X.nFull++;
    WSYNC();
  }
  bloom_query_insert(uni64(km2), uni64(nm2));   // keep the filter a superset of every live Trail's k-mers
  if (clen > X.refLen) { pool_free((uint32_t)uni((int)ch.buf)); return false; }
  return true;
}

// ---- the children of one group of up to 64 Trails, one child per LANE (bridge searches)
// oneMoreStep (Explorer.cpp:546-612) makes the children Trail by Trail, successor by successor, and everything it does
// per child — the record, the base, checkAims, ThinkIveAlreadyGotThere's filter — is independent of every other child
// except for where the child lands in the new set (Trail-major, A < C < G < T) and for the few that need the whole wave
// (a copy for a child that does not inherit its parent's buffer; recordBridge for one that stands on an aim; the exact
// window search for one whose k-mer the filter has seen).  So: the parents' lanes count their followed successors, a
// prefix sum gives every child its place, the children's lanes take over — parent's record, k-mer, distance term
// (make_child's expression), buffer, base, record into the new set at its PROVISIONAL place, aims, filter — and only the
// few special ones are then taken one at a time, in the children's order, by the very functions the sequential form
// uses (record_bridge_at_aim, is_cycle_exact); children that end there (:579-587) are closed up at the end.  A branching
// graph's steps carry 5 children on average: their per-child instructions (~ 2500 wave-cycles each) become per-step
// ones.  Returns false (nothing changed) when the group does not fit the form — more than 64 children, no room for
// them in the new set — and the caller takes the sequential form.
TALC_DN int bridge_children_by_lane(int tags_, uint32_t nc0, uint32_t nc1, uint32_t nc2, uint32_t nc3, int base_, int cnt_, int len_, int nNew_) {
  // (a real call: the arguments are the lanes' tags and counts, everything else is scalar; returns the new set's size, or
  //  -1 when the group does not fit the form)
  StepTags mine; mine.tags = tags_; mine.nc[0] = nc0; mine.nc[1] = nc1; mine.nc[2] = nc2; mine.nc[3] = nc3;
  const int base = uni(base_), cnt = uni(cnt_), len = uni(len_);
  int nNew = uni(nNew_);
  const int l = lane_id();
  const int ib = X.ia ^ 1;
  const int tl = base + l;
  const bool valid = l < cnt;
  int fm = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) { const int tag = (int)(int8_t)((mine.tags >> (8 * i)) & 0xff); if (valid && tag != TAG_NONE && tag != TAG_UNEXPECTED) fm |= 1 << i; }
  const int nF = __builtin_popcount((unsigned)fm);
  int incl = nF;
  TALC_WAVE_REDUCE(incl, talc_add_i32, 0);   // (the reduction's steps are an inclusive scan over the lanes)
  const int excl = incl - nF;
  const int nC = __builtin_amdgcn_readlane(incl, 63);
  if (nC > 64 || nNew + nC > TCAP || (uint32_t)(len + 1) > X.C.seqCap) return -1;
  {   // a Trail without a followed successor just ends
    unsigned long long m = ballot64(valid && nF == 0);
    while (m != 0ull) { const int t = base + (int)__builtin_ctzll(m); m &= m - 1ull; pool_free(tr_buf(X.ia, t)); }
  }
  if (nC == 0) return nNew;
  uint32_t TALC_AS3* const desc = (uint32_t TALC_AS3*)g_dp;            // [64] parent | base << 9 | inherits << 11
  uint32_t TALC_AS3* const dcnt = desc + 64;                            // [64] the successor's count
  uint32_t TALC_AS3* const fresh = desc + 128;                          // [64] buffers handed out in this step
  if (fm != 0) {
    const int lastI = 31 - __builtin_clz((unsigned)fm);                 // the last followed successor inherits the buffer
    int r = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if ((fm >> i) & 1) {
        desc[excl + r] = (uint32_t)tl | ((uint32_t)i << 9) | ((i == lastI) ? (1u << 11) : 0u);
        dcnt[excl + r] = mine.nc[i];
        ++r;
      }
    }
  }
  LSYNC();
  const bool has = l < nC;
  const uint32_t d = has ? desc[l] : 0u;
  const int t = (int)(d & 511u), bI = (int)((d >> 9) & 3u);
  const bool inherit = ((d >> 11) & 1u) != 0u;
  const uint32_t count = has ? dcnt[l] : 1u;
  const uint32_t K = X.P.K;
  const uint64_t kmask = (1ULL << (2 * K)) - 1;
  TrailRec ch = TrailRec();
  uint32_t pbuf = 0;
  if (has) {
    const TrailRec p = tr_get(X.ia, t);
    pbuf = p.buf;
    if (X.dirRight) { ch.kmer = ((p.kmer << 2) | (uint64_t)bI) & kmask; ch.nmask = p.nmask >> 1; }
    else { ch.kmer = ((uint64_t)bI << (2 * (K - 1))) | (p.kmer >> 2); ch.nmask = (p.nmask << 1) & ((1ULL << K) - 1); }
    ch.cnt = count; ch.score = p.score; ch.fail = p.fail;
    ch.dist = p.dist + fabs((double)p.cnt - (double)count) / sqrt((double)p.cnt);   // (tagNextNodes' distance term, as make_child)
    ch.lanc = p.lanc; ch.ranc = p.ranc; ch.buf = p.buf;
  }
  // buffers for the children that do not inherit one
  const unsigned long long freshM = ballot64(has && !inherit);
  const int nFresh = (int)__popcll(freshM);
  if (nFresh != 0) {
    for (int k = 0; k < nFresh; ++k) { const uint32_t id = (uint32_t)pool_alloc(); if (l == 0) fresh[k] = id; }
    if (l == 0) g_keep.owner = 0u;   // (buffers change hands: a kept wavefront may be about former contents)
    WSYNC();                         // (also: bases appended by single lanes in earlier steps become visible to the copying lanes)
    const unsigned long long below = (l == 0) ? 0ull : (~0ull >> (64 - l));
    if (has && !inherit) ch.buf = fresh[(int)__popcll(freshM & below)];
    unsigned long long m = freshM;
    while (m != 0ull) {
      const int f = (int)__builtin_ctzll(m); m &= m - 1ull;
      copy_trail((uint32_t)lane_get((int)ch.buf, f), (uint32_t)lane_get((int)pbuf, f), len, true);
    }
  }
  const int slot = nNew + l;                                            // provisional place in the new set
  if (has) {
    ((gu8)X.seqPool)[(uint64_t)ch.buf * X.C.seqCap + (uint32_t)len] = (uint8_t)bI;
    tr_put(ib, slot, ch);
  }
  WSYNC();
  // checkAims (Trail.cpp:273-285): the first aim whose k-mer is the child's tip
  int hit = -1;
  {
    const AnchorRec* aims = X.dirRight ? X.ancR : X.ancL;
    const int nAims = X.dirRight ? X.nAncR : X.nAncL;
    const int nl = min(nAims, AIMS_LDS);
    for (int a = 0; a < nl; ++a) if (has && hit < 0 && g_aimK[a] == ch.kmer && g_aimN[a] == ch.nmask) hit = a;
    for (int a = AIMS_LDS; a < nAims; ++a) { const AnchorRec ar = aims[a]; if (has && hit < 0 && ar.kmer == ch.kmer && ar.nmask == ch.nmask) hit = a; }
  }
  // ThinkIveAlreadyGotThere's filter (Trail.cpp:289-302)
  uint32_t hv = 0;
  bool maybe = false;
  if (has && hit < 0) maybe = bloom_query(ch.kmer, ch.nmask, hv) && (len > (int)K);
#ifdef TALC_PROF
  { const int q = (int)__popcll(ballot64(has && hit < 0)), x = (int)__popcll(ballot64(maybe)); if (l == 0) { g_prof[PF_CYQ] += (uint32_t)q; g_prof[PF_CYX] += (uint32_t)x; } }
#endif
  // the few that need the whole wave, in the children's order
  unsigned long long sp = ballot64(has && (hit >= 0 || maybe)), popM = 0ull;
  while (sp != 0ull) {
    const int f = (int)__builtin_ctzll(sp); sp &= sp - 1ull;
    const int hitF = lane_get(hit, f);
    if (hitF >= 0) {
      const uint64_t kmF = ((uint64_t)(uint32_t)lane_get((int)(uint32_t)(ch.kmer >> 32), f) << 32) | (uint32_t)lane_get((int)(uint32_t)ch.kmer, f);
      const uint64_t nmF = ((uint64_t)(uint32_t)lane_get((int)(uint32_t)(ch.nmask >> 32), f) << 32) | (uint32_t)lane_get((int)(uint32_t)ch.nmask, f);
      if (!uni((int)record_bridge_at_aim(nNew + f, hitF, len, kmF, nmF))) popM |= 1ull << f;   // :579-582 pop_back
    } else {
      const bool cyc = uni((int)is_cycle_exact(lane_get(t, f), nNew + f, len)) != 0;
#ifdef TALC_PROF
      if (l == 0 && cyc) g_prof[PF_CYHIT] += 1;
#endif
      if (cyc) { pool_free((uint32_t)lane_get((int)ch.buf, f)); popM |= 1ull << f; }          // :586-587 pop_back
    }
  }
  const bool surv = has && (((popM >> l) & 1ull) == 0ull);
  // the survivors' k-mers enter the filter now (a child on an aim was entered by recordBridge's path already)
  if (surv && hit < 0) {
    atomicOr(&g_bloom[hv >> 25], (1ull << ((hv >> 19) & 63u)) | (1ull << ((hv >> 13) & 63u)));
    const uint32_t wm = X.wideMask;
    if (wm != 0u) wide_or(X.wideBloom + wide_word(hv, wm), wide_bits(hv));
  }
  const unsigned long long survM = ballot64(surv);
  if (popM != 0ull) {   // close the gaps (a record may have changed where it stands: recordBridge sets the anchor reached)
    TrailRec r = TrailRec();
    if (surv) r = tr_get(ib, slot);
    WSYNC();
    const unsigned long long below = (l == 0) ? 0ull : (~0ull >> (64 - l));
    if (surv) tr_put(ib, nNew + (int)__popcll(survM & below), r);
  }
  nNew += (int)__popcll(survM);
  WSYNC();
  return nNew;
}

// (inlined at its one call site: as a real call — tried again in round 3, after step_edge had become one — the 4.3 M
//  generic bridge steps of a config-2 launch pay for a prologue each: 40.7 -> 42.5 ms)
TALC_D int step_bridge(int nCur, int len, uint32_t& stepCounter) {
  PROF_DECL;
  const DevParams& P = X.P;
  const int l = lane_id();
  const AnchorRec* aims = X.dirRight ? X.ancR : X.ancL;
  const int nAims = X.dirRight ? X.nAncR : X.nAncL;
  const int ib = X.ia ^ 1;
  int nNew = 0;
  uint32_t pend = 0; int nPend = 0;   // this step's children, to be entered into the filter (bloom_flush)
  const bool complexIn = ((uint32_t)nCur > P.MAXB);
  const bool byLane = uni((int)X.childrenByLane) != 0;
  for (int base = 0; base < nCur; base += 64) {
    const int tl = base + l;
    PROF_BEGIN();
    const StepTags mine = probe_and_tag(tl, tl < nCur, complexIn);
    PROF_END(PF_PROBE);
    const int cnt = min(64, nCur - base);
    X.steps += (unsigned long long)cnt;
    if (byLane && nCur > 1) {   // (one Trail: the sequential form is as quick, and a call saves and restores 17 registers)
      PROF_BEGIN();
      const int nn = uni(bridge_children_by_lane(mine.tags, mine.nc[0], mine.nc[1], mine.nc[2], mine.nc[3], base, cnt, len, nNew));
      PROF_END(PF_CHILD);
      if (nn >= 0) { nNew = nn; continue; }
    }
    for (int tt = 0; tt < cnt; ++tt) {
      const int t = base + tt;
      const int tags = lane_get(mine.tags, tt);
      // this Trail's 4 successor counts / distances, broadcast once (static register indices)
      const uint32_t pnc0 = (uint32_t)lane_get((int)mine.nc[0], tt), pnc1 = (uint32_t)lane_get((int)mine.nc[1], tt),
                     pnc2 = (uint32_t)lane_get((int)mine.nc[2], tt), pnc3 = (uint32_t)lane_get((int)mine.nc[3], tt);
      // the last successor of this Trail inherits its sequence buffer
      const int lastI = last_successor(tags);
      if (lastI < 0) pool_free(tr_buf(X.ia, t));   // no successor: the Trail just ends
#pragma nounroll
      for (int i = 0; i < 4; ++i) {   // (one copy of the child's code, not four)
        const int tag = (int)(int8_t)((tags >> (8 * i)) & 0xff);
        if (tag == TAG_NONE || tag == TAG_UNEXPECTED) continue;
        const uint32_t nc = (i == 0) ? pnc0 : (i == 1) ? pnc1 : (i == 2) ? pnc2 : pnc3;
        if (nNew >= TCAP) { X.overflow |= OVF_TRAILS; continue; }
        if ((uint32_t)(len + 1) > X.C.seqCap) { X.overflow |= OVF_SEQ; continue; }
        PROF_BEGIN();
        uint64_t km2, nm2;
        make_child<true>(t, nNew, i, len, nc, i == lastI, km2, nm2);
        PROF_END(PF_CHILD);
        // checkAims (Trail.cpp:273-285): first aim whose k-mer equals the child's tip
        int hit = -1;
        PROF_BEGIN();
        {
          const int nl = min(nAims, AIMS_LDS);
          const bool eq0 = (l < nl) && (g_aimK[l] == km2) && (g_aimN[l] == nm2);
          const unsigned long long m0 = ballot64(eq0);
          if (m0) hit = (int)__ffsll((long long)m0) - 1;
          for (int ab = AIMS_LDS; ab < nAims && hit < 0; ab += 64) {
            const int a = ab + l;
            const bool eq = (a < nAims) && (aims[a].kmer == km2) && (aims[a].nmask == nm2);
            const unsigned long long m = ballot64(eq);
            if (m) hit = ab + (int)__ffsll((long long)m) - 1;
          }
        }
        PROF_END(PF_AIMS);
        if (hit >= 0) {
          if (!uni((int)record_bridge_at_aim(nNew, hit, len, km2, nm2))) continue;   // :579-582 pop_back
          ++nNew;
        } else {
          PROF_BEGIN();
          uint32_t hv;
          const bool cyc = uni((int)is_cycle(t, nNew, len, km2, nm2, hv)) != 0;
          PROF_END(PF_CYCLE);
          if (cyc) { pool_free(tr_buf(ib, nNew)); continue; }   // :586-587 pop_back
          if (l == nPend) pend = hv;
          if (++nPend == 64) { bloom_flush(pend, 64); nPend = 0; }
          ++nNew;
        }
      }
    }
  }
  bloom_flush(pend, nPend);
  const bool complex = ((uint32_t)nNew > P.MAXB);
  ++stepCounter;
#ifdef TALC_PROF
  if (l == 0) g_prof[(nCur == 1 && nNew == 1) ? PF_SB11 : (nCur == 1 && nNew == 2) ? PF_SB12 : (nCur == 2 && nNew == 1) ? PF_SB21 : (nCur == 1 && nNew == 0) ? PF_SB10 : PF_SBOTHER] += 1;
#endif
  int nOut;
  if (complex & (stepCounter % P.CHECK_INTERVAL == 0)) {
    PROF_BEGIN();
    score_bridges(ib, nNew, len, stepCounter);
    PROF_END(PF_SCOREBR);
    bool cx = false;
    PROF_BEGIN();
    nOut = uni(garden(nNew, len + 1, cx));
    PROF_END(PF_GARDEN);
    X.complexRegion |= cx;
  } else {
    swap_sets();
    nOut = nNew;
  }
  return nOut;
}

// Explorer::scoreEdges (Explorer.cpp:709-740) on the new set (n trails of length len); survivors
// are compacted in place; returns their number
// (ib = the Trail set to score: the new set of a generic step, or the current set when the fast-forward took the step)
TALC_DNC int score_edges_multi(int ib_, int n_, int len_, int& xdrop_);
TALC_D int score_edges(int ib_, int n_, int len_, int& xdrop_) {
  const int n = uni(n_), len = uni(len_);
  if (n == 0) return 0;
  const int ib = uni(ib_);
  const int xdrop = uni(xdrop_) + 2;
  if (n == 1) {
    // the common case, one live Trail: the same outcome without the survivor flags' trip through memory and its barriers
    const bool ok = uni((int)trail_seed_and_extend(ib, 0, len, xdrop)) != 0;
    xdrop_ = ok ? (int)((double)uni(tr_get(ib, 0).score) * (-1)) : 0;
    if (!ok) { record_edge(ib, 0, len); pool_free((uint32_t)uni((int)tr_buf(ib, 0))); return 0; }
    return 1;
  }
  int xd = xdrop - 2;
  const int kept = uni(score_edges_multi(ib, n, len, xd));
  xdrop_ = uni(xd);
  return kept;
}
// ... with several live Trails (a real, cold call: the edge loop of a unique-sequence graph never gets here)
TALC_DNC int score_edges_multi(int ib_, int n_, int len_, int& xdrop_) {
  const int n = uni(n_), len = uni(len_), ib = uni(ib_);
  const int l = lane_id();
  const int xdrop = uni(xdrop_) + 2;
  int new_xdrop = 0;
  int nSel = 0;
  // trash paths are only needed when nobody survives: remember them by flag in gKept
  for (int t = 0; t < n; ++t) {
    const bool ok = uni((int)trail_seed_and_extend(ib, t, len, xdrop)) != 0;
    if (l == 0) X.gKept[t] = ok ? 1u : 0u;
    if (ok) {
      const int current_xdrop = (int)((double)uni(tr_get(ib, t).score) * (-1));
      if ((new_xdrop > current_xdrop) || (new_xdrop == 0)) new_xdrop = current_xdrop;
      ++nSel;
    }
  }
  WSYNC();
  xdrop_ = new_xdrop;
  if (nSel == 0) {
    for (int t = 0; t < n; ++t) { record_edge(ib, t, len); pool_free((uint32_t)uni((int)tr_buf(ib, t))); }
    return 0;
  }
  // compact survivors to the front, keeping their order (metadata only: the sequence buffers stay
  // where they are); the buffers of the trashed Trails go back to the pool
  int w = 0;
  for (int t = 0; t < n; ++t) {
    if (uni((int)X.gKept[t])) {
      if (w != t) {
        const TrailRec r = tr_get(ib, t);
        WSYNC();
        if (l == 0) tr_put(ib, w, r);
        WSYNC();
      }
      ++w;
    } else {
      pool_free((uint32_t)uni((int)tr_buf(ib, t)));
    }
  }
  WSYNC();
  return w;
}

// Explorer::oneMoreStepInTheDark (Explorer.cpp:615-687)
// (a real call since round 3: the generic step is 4 % of an edge's steps, and its body in the middle of search_edge's
//  loop cost the single-Trail iterations around it more than the call costs the generic ones: 41.4 -> 40.7 ms)
TALC_DN int step_edge(int nCur, int len, uint32_t& stepCounter, uint32_t PATH_MAXLENGTH, int& xdrop) {
  PROF_DECL;
  const DevParams& P = X.P;
  const int l = lane_id();
  const int ib = X.ia ^ 1;
  int nNew = 0;
  uint32_t pend = 0; int nPend = 0;   // (see bloom_query)
  const bool complexIn = (nCur > 7);   // :634 hard-coded
  for (int base = 0; base < nCur; base += 64) {
    const int tl = base + l;
    PROF_BEGIN();
    const StepTags mine = probe_and_tag(tl, tl < nCur, complexIn);
    PROF_END(PF_PROBE);
    const int cnt = min(64, nCur - base);
    X.steps += (unsigned long long)cnt;
    for (int tt = 0; tt < cnt; ++tt) {
      const int t = base + tt;
      const int tags = lane_get(mine.tags, tt);
      // this Trail's 4 successor counts / distances, broadcast once (static register indices)
      const uint32_t pnc0 = (uint32_t)lane_get((int)mine.nc[0], tt), pnc1 = (uint32_t)lane_get((int)mine.nc[1], tt),
                     pnc2 = (uint32_t)lane_get((int)mine.nc[2], tt), pnc3 = (uint32_t)lane_get((int)mine.nc[3], tt);
      int counter = 0;
      const int lastI = last_successor(tags);
#pragma nounroll
      for (int i = 0; i < 4; ++i) {   // (one copy of the child's code, not four)
        const int tag = (int)(int8_t)((tags >> (8 * i)) & 0xff);
        if (tag == TAG_NONE || tag == TAG_UNEXPECTED) continue;
        ++counter;
        const uint32_t nc = (i == 0) ? pnc0 : (i == 1) ? pnc1 : (i == 2) ? pnc2 : pnc3;
        if (nNew >= TCAP) { X.overflow |= OVF_TRAILS; continue; }
        if ((uint32_t)(len + 1) > X.C.seqCap) { X.overflow |= OVF_SEQ; continue; }
        PROF_BEGIN();
        uint64_t km2, nm2;
        make_child<false>(t, nNew, i, len, nc, i == lastI, km2, nm2);
        PROF_END(PF_CHILD);
        PROF_BEGIN();
        uint32_t hv;
        const bool cycle = uni((int)is_cycle(t, nNew, len, km2, nm2, hv)) != 0;
        PROF_END(PF_CYCLE);
        if (cycle || (stepCounter + 1 > PATH_MAXLENGTH)) {
          trail_seed_and_extend(ib, nNew, len + 1, xdrop);
          record_edge(ib, nNew, len + 1);
          pool_free(tr_buf(ib, nNew));
          continue;   // pop_back
        }
        if (l == nPend) pend = hv;
        if (++nPend == 64) { bloom_flush(pend, 64); nPend = 0; }
        ++nNew;
      }
      if (counter == 0) {   // dead end (:657-662)
        trail_seed_and_extend(X.ia, t, len, xdrop);
        record_edge(X.ia, t, len);
        pool_free(tr_buf(X.ia, t));
      }
    }
  }
  bloom_flush(pend, nPend);
  ++stepCounter;
#ifdef TALC_PROF
  if (l == 0) g_prof[PF_SEGEN] += 1;
#endif
  int nOut;
  if ((stepCounter % P.CHECK_INTERVAL == 0) || ((uint32_t)nNew >= P.MAX_BORDER_PATHS)) {
    nNew = uni(score_edges(X.ia ^ 1, nNew, len + 1, xdrop));
    if (nNew > 5) {
      bool cx = false;
      PROF_BEGIN();
      nOut = uni(garden(nNew, len + 1, cx));
      PROF_END(PF_GARDEN);
      X.complexRegion |= cx;
    } else { swap_sets(); nOut = nNew; }
  } else { swap_sets(); nOut = nNew; }
  return nOut;
}

// ------------------------------------------------------------------ single-Trail fast-forward
// The overwhelmingly common state of a search is ONE live Trail whose tip has exactly ONE
// successor in the table (96 % of all Trail steps).  For that state oneMoreStep /
// oneMoreStepInTheDark reduce to: the successor is EXPECTED (counter == 1, Explorer.cpp:1251), the
// child inherits everything, no aim is hit, no cycle, no scoring is due.  fast_forward performs
// exactly those steps and stops BEFORE any step that is not of that kind (several successors, dead
// end, aim reached, possible cycle, a scoreEdges step, limits): the generic step then redoes that
// step from the unchanged state.
//
// A wave is one instruction stream and the SIMD's issue rate is what the walk is bound by, so the
// loop is written for instruction count: everything is wave-uniform and lives on the scalar unit —
// the bucket arrives by a scalar load (the table is read through the constant address space), its
// address is one multiply-hash of the key, the same hash indexes the cycle filter, the filter's
// words sit in one register pair (word i in lane i, read and written by v_readlane / v_writelane) —
// and the per-step work that does not decide anything is batched: the committed bases and counts go
// to lane (step mod 64) of two registers, bases are stored and the distance terms
// |c - n| / sqrt(c) evaluated 64 steps at a time, lane-parallel, then added in path order (the same
// double operations in the same order as the step-by-step form).
template <bool dirRight>
TALC_D int fast_forward_dir(int len_, uint32_t& stepCounter_, uint32_t PATH_MAXLENGTH_, bool edge_) {
  const DevParams& P = X.P;
  const int l = lane_id();
  // (arguments of a non-inlined function arrive in vector registers: make every one of them scalar)
  const bool edge = uni((int)edge_) != 0;
  if (uni((int)(X.traceSteps)) != 0) return 0;
  const TrailRec r0 = tr_get(X.ia, 0);
  if (uni64(r0.nmask) != 0ull) return 0;
  const uint32_t K = (uint32_t)uni((int)P.K), MINC = (uint32_t)uni((int)P.MIN_COUNT), CHECK = (uint32_t)uni((int)P.CHECK_INTERVAL);
  const uint32_t seqCap = (uint32_t)uni((int)X.C.seqCap), PMAX = (uint32_t)uni((int)PATH_MAXLENGTH_);
  const uint64_t cap = uni64(X.T.capacity);
  const Bucket TALC_AS4* tab = (const Bucket TALC_AS4*)uni_ptr(dirRight ? X.T.right : X.T.left);
  const uint64_t kmask = (1ULL << (2 * K)) - 1, m1 = (1ULL << (2 * (K - 1))) - 1;
  uint64_t kmer = uni64(r0.kmer);
  uint32_t cnt = (uint32_t)uni((int)r0.cnt);
  const int len0 = uni(len_);
  const uint32_t sc0 = (uint32_t)uni((int)stepCounter_);
  // number of steps this call may commit: path length limit, buffer, and (edges) the next scoreEdges step
  int maxSteps = 0;
  if (sc0 < PMAX && (uint32_t)len0 < seqCap) {
    maxSteps = (int)min(PMAX - sc0, seqCap - (uint32_t)len0);
    if (edge) maxSteps = min(maxSteps, (int)(CHECK - (sc0 % CHECK)));   // up to and including the next scoring step
  }
  gu8 seq = (gu8)uni_ptr(X.seqPool + (uint64_t)r0.buf * X.C.seqCap);
  const unsigned long long bw0 = g_bloom[l], bw1 = g_bloom[l + 64];   // words l and l + 64 of the filter in this lane
  int bwLo = (int)(uint32_t)bw0, bwHi = (int)(uint32_t)(bw0 >> 32), bxLo = (int)(uint32_t)bw1, bxHi = (int)(uint32_t)(bw1 >> 32);
  int recN = 0, recB = 0;
  uint32_t cFlush = cnt;   // count of the tip before the first unflushed step
  double dist = r0.dist;
  int done = 0, flushed = 0;
  uint64_t key = dirRight ? (kmer & m1) : (kmer >> 2);

  auto flush = [&]() {
    const int n = done - flushed;
    if (n <= 0) return;
    if (l < n) seq[len0 + flushed + l] = (uint8_t)recB;
    // distance terms |c - n| / sqrt(c) (Explorer.cpp:1247): c of step j is n of step j-1
    int cprev = lane_shr1(recN);
    if (l == 0) cprev = (int)cFlush;
    const double term = fabs((double)(uint32_t)cprev - (double)(uint32_t)recN) / sqrt((double)(uint32_t)cprev);
    const unsigned long long tb = (unsigned long long)__double_as_longlong(term);
    const int tLo = (int)(uint32_t)tb, tHi = (int)(uint32_t)(tb >> 32);
    for (int j = 0; j < n; ++j) {
      const unsigned long long v = ((unsigned long long)(uint32_t)lane_get(tHi, j) << 32) | (uint32_t)lane_get(tLo, j);
      dist = dist + __longlong_as_double((long long)v);
    }
    cFlush = (uint32_t)lane_get(recN, n - 1);
    flushed = done;
  };

  // The load of a step's bucket is issued as soon as its address is known — right after the previous step has chosen
  // its base, ahead of that step's aim / cycle checks and bookkeeping — so that the dependent memory latency runs
  // under that work instead of after it.
  uint64_t slot = table_slot(table_hash(key), cap);
  v8u32 b = *(const v8u32 TALC_AS4*)(tab + slot);
  while (done < maxSteps) {
    // ---- the tip's bucket (linear probing from its home slot; the home slot's load is already in flight)
    bool found = false;
    while (true) {
      const uint64_t bk = ((uint64_t)b[1] << 32) | b[0];
      if ((bk & kKeyMask) == key) { found = true; break; }
      if (bk == kEmptyKey) break;
      if (++slot == cap) slot = 0;
      b = *(const v8u32 TALC_AS4*)(tab + slot);
    }
    if (!found) break;
    // (keeps all eight registers of the load occupied until it has landed: the compiler would otherwise put a
    //  temporary into the unused colour words and wait for the load right after issuing it)
    asm volatile("" :: "s"(b[6]), "s"(b[7]));
    // ---- exactly one successor with count >= MIN_COUNT?  (bit i of m: count i >= MIN_COUNT; the compare's SCC is
    // shifted in with s_addc: two scalar instructions per count)
    const uint32_t c0 = b[2], c1 = b[3], c2 = b[4], c3 = b[5];
    int m;
    asm("s_cmp_ge_u32 %4, %5\n\ts_cselect_b32 %0, 1, 0\n\ts_cmp_ge_u32 %3, %5\n\ts_addc_u32 %0, %0, %0\n\t"
        "s_cmp_ge_u32 %2, %5\n\ts_addc_u32 %0, %0, %0\n\ts_cmp_ge_u32 %1, %5\n\ts_addc_u32 %0, %0, %0"
        : "=&s"(m) : "s"(c0), "s"(c1), "s"(c2), "s"(c3), "s"(MINC) : "scc");
    if (m == 0 || (m & (m - 1)) != 0) break;
    const int which = __builtin_ctz((unsigned)m);
    uint32_t nc;   // the one count >= MIN_COUNT is the largest of the four (taken before the next load reuses b's registers)
    asm("s_max_u32 %0, %1, %2\n\ts_max_u32 %0, %0, %3\n\ts_max_u32 %0, %0, %4" : "=&s"(nc) : "s"(c0), "s"(c1), "s"(c2), "s"(c3) : "scc");
    uint64_t km2;
    if (dirRight) km2 = ((kmer << 2) | (uint64_t)which) & kmask;
    else km2 = ((uint64_t)which << (2 * (K - 1))) | (kmer >> 2);
    // ---- the next tip's bucket: issue its load now (the new tip's filter hash is the table hash of its successor
    // key = the hash of this probe); everything that does not feed the address comes after
    key = dirRight ? (km2 & m1) : (km2 >> 2);
    const uint64_t h2 = table_hash(key);
    slot = table_slot(h2, cap);
    b = *(const v8u32 TALC_AS4*)(tab + slot);
    // ---- aim check (bridges) and cycle prefilter in one query: the search's filter holds the aims as well as every
    // k-mer walked so far (init_first_trail), so "absent" means neither an aim nor a cycle; a possible hit of either
    // kind is left to the generic step, which redoes this step from the unchanged state
    const int bwi = bloom_word(h2);
    const int bwl = bwi & 63;
    const bool upper = (bwi & 64) != 0;
    const unsigned long long bm = bloom_mask(h2);
    const unsigned long long bv = upper ? (((unsigned long long)(uint32_t)lane_get(bxHi, bwl) << 32) | (uint32_t)lane_get(bxLo, bwl))
                                        : (((unsigned long long)(uint32_t)lane_get(bwHi, bwl) << 32) | (uint32_t)lane_get(bwLo, bwl));
    if ((bv & bm) == bm) break;
    // ---- commit the step
    if (upper) { bxLo = lane_set(bxLo, (int)(uint32_t)(bv | bm), bwl); bxHi = lane_set(bxHi, (int)(uint32_t)((bv | bm) >> 32), bwl); }
    else { bwLo = lane_set(bwLo, (int)(uint32_t)(bv | bm), bwl); bwHi = lane_set(bwHi, (int)(uint32_t)((bv | bm) >> 32), bwl); }
    const int rs = done - flushed;
    recN = lane_set(recN, (int)nc, rs);
    recB = lane_set(recB, which, rs);
    kmer = km2;
    cnt = nc;
    ++done;
    if (done - flushed == 64) flush();
  }

  flush();
  g_bloom[l] = ((unsigned long long)(uint32_t)bwHi << 32) | (uint32_t)bwLo;
  g_bloom[l + 64] = ((unsigned long long)(uint32_t)bxHi << 32) | (uint32_t)bxLo;
  stepCounter_ = sc0 + (uint32_t)done;
#ifdef TALC_PROF
  if (l == 0) { g_prof[PF_NCALLS] += 1; g_prof[PF_NSTEPS] += (uint32_t)done; }
#endif
  if (done) {
    X.steps += (unsigned long long)done;
    if (l == 0) {
      TrailRec r = r0;
      r.kmer = kmer; r.cnt = cnt; r.dist = dist;
      tr_put(X.ia, 0, r);
    }
  }
  LSYNC();
  return done;
}

// ---- walk-table form of the fast-forward: one 32-byte record (WalkEntry, talc_common.h) describes the next
// TALC_WALK_LEVELS steps of a Trail that keeps following its only solid successor, so a record's steps are taken
// together, one level per lane: lane j < 12 loads level j (lanes 12, 13 the key), the lanes test "exactly one successor"
// and build their k-mers from the prefix of the levels' bases, hash them, query the search's filter (LDS) — a
// ballot gives the number of steps that can be committed, and those lanes insert their k-mers, store their bases
// and record their counts.  One dependent memory access and ~130 instructions per record instead of per step.
// A k-mer that repeats WITHIN a record (a cycle of period <= 11) would not be seen by a query that precedes the
// record's inserts: lanes compare their hashes with the lower lanes' (equal hash = possible cycle = stop there).
template <int P>
TALC_D uint32_t dpp_row_shr(uint32_t v, uint32_t old) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, 0x110 + P, 0xF, 0xF, false);
}

// WIDEF: the search's cycle filter is the wide one in HBM (an instance of its own: the common form's loop stays as it is)
template <bool dirRight, bool WIDEF>
TALC_D int fast_forward_walk(int len_, uint32_t& stepCounter_, uint32_t PATH_MAXLENGTH_, bool edge_) {
  PROF_DECL; PROF_DECL2;
  PROF_BEGIN2();
  const DevParams& P = X.P;
  const int l = lane_id();
  const bool edge = uni((int)edge_) != 0;
  if (uni((int)(X.traceSteps)) != 0) return 0;
  const TrailRec r0 = tr_get(X.ia, 0);
  if (uni64(r0.nmask) != 0ull) return 0;
  const uint32_t K = (uint32_t)uni((int)P.K), CHECK = (uint32_t)uni((int)P.CHECK_INTERVAL);
  const uint32_t seqCap = (uint32_t)uni((int)X.C.seqCap), PMAX = (uint32_t)uni((int)PATH_MAXLENGTH_);
  const uint64_t cap = uni64(X.T.capacity);
  const uint32_t TALC_AS1* wtab = (const uint32_t TALC_AS1*)uni_ptr(dirRight ? X.T.walkRight : X.T.walkLeft);
  const uint32_t wideMask = WIDEF ? (uint32_t)uni((int)X.wideMask) : 0u;
  unsigned long long* const wideBloom = WIDEF ? (unsigned long long*)uni_ptr(X.wideBloom) : nullptr;
  const uint64_t kmask = (1ULL << (2 * K)) - 1, m1 = (1ULL << (2 * (K - 1))) - 1;
  uint64_t kmer = uni64(r0.kmer);
  uint32_t cnt = (uint32_t)uni((int)r0.cnt);
  const int len0 = uni(len_);
  const uint32_t sc0 = (uint32_t)uni((int)stepCounter_);
  int maxSteps = 0;
  if (sc0 < PMAX && (uint32_t)len0 < seqCap) {
    maxSteps = (int)min(PMAX - sc0, seqCap - (uint32_t)len0);
    if (edge) maxSteps = min(maxSteps, (int)(CHECK - (sc0 % CHECK)));   // up to and including the next scoring step
  }
  gu8 seq = (gu8)uni_ptr(X.seqPool + (uint64_t)r0.buf * X.C.seqCap);
  const int nAims = edge ? 0 : uni(dirRight ? X.nAncR : X.nAncL);
  int lanc = uni((int)r0.lanc), ranc = uni((int)r0.ranc);
  bool popped = false;
  uint32_t* recN = (uint32_t*)g_dp;   // counts of the committed, not yet flushed steps (the DP stage is idle here)
  uint32_t cFlush = cnt;
  double dist = r0.dist;
  int done = 0, flushed = 0;

  auto flush = [&]() {
    const int n = done - flushed;
    if (n <= 0) return;
    PROF_DECL; PROF_BEGIN();
    // distance terms |c - n| / sqrt(c) (Explorer.cpp:1247): c of step j is n of step j-1
    const uint32_t cn = (l < n) ? recN[l] : 1u;
    const uint32_t cprev = (l == 0) ? cFlush : ((l < n) ? recN[l - 1] : 1u);
    const double term = fabs((double)cprev - (double)cn) / sqrt((double)cprev);
    // added in path order (the same double additions as step by step): the terms go through LDS, whose reads the
    // compiler can issue several at a time ahead of the dependent additions
    double* recT = (double*)(g_dp + 64);
    recT[l] = term;
    LSYNC();
#pragma unroll 8
    for (int j = 0; j < n; ++j) dist = dist + recT[j];
    cFlush = (uint32_t)lane_get((int)cn, n - 1);
    flushed = done;
    LSYNC();
    PROF_END(PF_FFFLUSH);
  };

  // lane j < 12 reads the dword that holds level j (two 16-bit levels per dword), lanes 12 / 13 the two halves of the
  // key (the lanes above them repeat lane 12)
  const uint32_t laneOff = (l < TALC_WALK_LEVELS) ? (uint32_t)(2 + (l >> 1)) : (l == TALC_WALK_LEVELS + 1 ? 1u : 0u);
  const uint32_t laneShift = (l < TALC_WALK_LEVELS && (l & 1)) ? 16u : 0u;
  const int lj = min(l, TALC_WALK_LEVELS - 1);                // shift amounts stay in range on the idle lanes
  const uint32_t laneSingle = (l < TALC_WALK_LEVELS) ? kWalkSingle : 0u;
  uint64_t key = dirRight ? (kmer & m1) : (kmer >> 2);
  uint32_t hh = (uint32_t)(table_hash(key) >> 32);
  PROF_END2(PF_FFENTRY);
  // The record a Trail reaches when it takes all of the current record's steps is requested as soon as the lanes have
  // hashed their tips — before the filter query, the aim check and the commit — so that its latency runs under that work.
  uint32_t ePre = 0;
  bool havePre = false;
  while (done < maxSteps) {
    if (done - flushed > 64 - TALC_WALK_LEVELS) flush();
    PROF_BEGIN();
    uint64_t slot = ((uint64_t)hh * (uint64_t)(uint32_t)cap) >> 32;
    uint32_t e = havePre ? ePre : wtab[slot * 8 + laneOff];
    havePre = false;
    bool found = true;
    while (true) {   // linear probing, as in the bucket table (same slots)
      const uint64_t bk = ((uint64_t)(uint32_t)lane_get((int)e, TALC_WALK_LEVELS + 1) << 32) | (uint32_t)lane_get((int)e, TALC_WALK_LEVELS);
      if (bk == key) break;
      if (bk == kEmptyKey) { found = false; break; }
      if (++slot == cap) slot = 0;
      e = wtab[slot * 8 + laneOff];
    }
    PROF_END(PF_FFLOAD);
#ifdef TALC_PROF
    if (l == 0) g_prof[PF_NRECS] += 1;
#endif
    if (!found) break;
    PROF_BEGIN();
    const uint32_t lev = (e >> laneShift) & 0xFFFFu, top = lev & kWalkTopNone;
#ifdef TALC_PROF
    {   // why the walk will stop in this record, if it does: the first level that is not "single"
      const unsigned long long pm = ballot64((l < TALC_WALK_LEVELS) && (lev & kWalkSingle) != 0u);
      const int j0 = __builtin_ctzll(~pm);
      if (j0 < TALC_WALK_LEVELS && j0 < maxSteps - done) {
        const uint32_t t0 = (uint32_t)lane_get((int)top, j0);
        if (l == 0) g_prof[(t0 >= X.P.MIN_COUNT) ? PF_FSFORK : PF_FSDEAD] += 1;
      }
    }
#endif
    // levels that are "exactly one successor with count >= MIN_COUNT" (decided at upload for the table's MIN_COUNT, which
    // is this context's), from level 0 up to the first that is not
    const unsigned long long passMask = ballot64((lev & laneSingle) != 0u);   // (a plain compare: the ballot is its lane mask)
    int nOK = min(__builtin_ctzll(~passMask), maxSteps - done);
    if (nOK == 0) break;
    // lane j's tip after its step: the current tip shifted by j+1 bases, with the bases of levels 0..j
    const uint32_t which = (l < TALC_WALK_LEVELS) ? (lev >> kWalkBaseShift) : 0u;
    uint32_t pre = which << (dirRight ? 2 * (TALC_WALK_LEVELS - 1 - lj) : 2 * lj);
    pre |= dpp_row_shr<1>(pre, 0u);
    pre |= dpp_row_shr<2>(pre, 0u);
    pre |= dpp_row_shr<4>(pre, 0u);
    pre |= dpp_row_shr<8>(pre, 0u);
    uint64_t km;
    if (dirRight) km = ((kmer << (2 * (lj + 1))) | (uint64_t)(pre >> (2 * (TALC_WALK_LEVELS - 1 - lj)))) & kmask;
    else km = ((uint64_t)pre << (2 * (K - 1 - (uint32_t)lj))) | (kmer >> (2 * (lj + 1)));
    const uint64_t key2 = dirRight ? (km & m1) : (km >> 2);
    const uint32_t hv = (uint32_t)(table_hash(key2) >> 32);   // the filter hash of the new tip = the hash of its probe
    {
      const uint64_t slotN = ((uint64_t)(uint32_t)lane_get((int)hv, TALC_WALK_LEVELS - 1) * (uint64_t)(uint32_t)cap) >> 32;
      ePre = wtab[slotN * 8 + laneOff];
    }
    // possible cycle inside the record: the same hash on a lower lane
    // (gfx950 has no DPP form of the vector compares — "dpp variant of this instruction is not supported" — so each
    //  distance stays a copy, a DPP move and a compare)
    bool dup = dpp_row_shr<1>(hv, ~hv) == hv;
    dup |= dpp_row_shr<2>(hv, ~hv) == hv;
    dup |= dpp_row_shr<3>(hv, ~hv) == hv;
    dup |= dpp_row_shr<4>(hv, ~hv) == hv;
    dup |= dpp_row_shr<5>(hv, ~hv) == hv;
    dup |= dpp_row_shr<6>(hv, ~hv) == hv;
    dup |= dpp_row_shr<7>(hv, ~hv) == hv;
    dup |= dpp_row_shr<8>(hv, ~hv) == hv;
    dup |= dpp_row_shr<9>(hv, ~hv) == hv;
    dup |= dpp_row_shr<10>(hv, ~hv) == hv;
    dup |= dpp_row_shr<11>(hv, ~hv) == hv;
    static_assert(TALC_WALK_LEVELS == 12, "the lane roles above are written for 12 levels (+ 2 key lanes) in a 16-lane row");
    // aim / cycle query against the search's filter (init_first_trail entered the aims)
    const int bwi = (int)(hv >> 25);     // bloom_word / bloom_mask on the upper half of the hash
    const unsigned long long bm = (1ull << ((hv >> 19) & 63u)) | (1ull << ((hv >> 13) & 63u));
    bool seen = (g_bloom[bwi] & bm) == bm;
    unsigned long long* const wideW = WIDEF ? wideBloom + wide_word(hv, wideMask) : nullptr;
    const unsigned long long bmW = WIDEF ? wide_bits(hv) : 0ull;
    if (WIDEF) {   // a long search's filter (HBM): only the levels the LDS filter cannot clear ask it
      const bool ask = seen && (l < TALC_WALK_LEVELS);
      if (ballot64(ask) != 0ull) { const unsigned long long bvW = ask ? wide_load(wideW) : 0ull; seen = ask && ((bvW & bmW) == bmW); }
    }
    const unsigned long long hitMask = ballot64(seen || dup) | (1ull << TALC_WALK_LEVELS);
    const int hitLevel = __builtin_ctzll(hitMask);
    // a filter hit on a level that could otherwise be taken: if its k-mer is an aim, that step is a plain step that
    // also records the bridge (oneMoreStep, Explorer.cpp:566-583) — taken here as well; anything else (a possible
    // cycle, a false positive, aims beyond the LDS copy) is left to the generic step
    int aimIdx = -1;
    if (nAims > 0 && nAims <= AIMS_LDS && hitLevel < nOK) {
      const uint64_t kmH = ((uint64_t)(uint32_t)lane_get((int)(uint32_t)(km >> 32), hitLevel) << 32) | (uint32_t)lane_get((int)(uint32_t)km, hitLevel);
      const unsigned long long am = ballot64((l < nAims) && (g_aimK[l < AIMS_LDS ? l : 0] == kmH) && (g_aimN[l < AIMS_LDS ? l : 0] == 0ull));
      if (am != 0ull) aimIdx = (int)__builtin_ctzll(am);   // checkAims takes the first aim that matches (Trail.cpp:273-285)
    }
#ifdef TALC_PROF
    if (hitLevel < nOK && aimIdx < 0 && l == 0) g_prof[PF_FSFILT] += 1;
#endif
    nOK = min(nOK, hitLevel);
    const int nTake = nOK + (aimIdx >= 0 ? 1 : 0);
    if (nTake == 0) break;
    // ---- commit nTake steps
    if (l < nTake) {
      atomicOr(&g_bloom[bwi], bm);
      if (WIDEF) wide_or(wideW, bmW);
      recN[done - flushed + l] = top;
      seq[len0 + done + l] = (uint8_t)which;
    }
    const int last = nTake - 1;
    kmer = ((uint64_t)(uint32_t)lane_get((int)(uint32_t)(km >> 32), last) << 32) | (uint32_t)lane_get((int)(uint32_t)km, last);
    key = dirRight ? (kmer & m1) : (kmer >> 2);
    hh = (uint32_t)lane_get((int)hv, last);
    havePre = (last == TALC_WALK_LEVELS - 1);   // the requested record is the one the next round starts with
    cnt = (uint32_t)lane_get((int)top, last);
    done += nTake;
    LSYNC();
    PROF_END(PF_FFREC);
    if (aimIdx >= 0) {
      // recordBridge (Explorer.cpp:1097-1101) for the Trail as it stands after this step
      flush();
      WSYNC();   // the bases the lanes have stored are read back by the copy
      const int apos = (int)g_aimPos[aimIdx];
      if (dirRight) ranc = apos; else lanc = apos;
      const uint32_t clen = (uint32_t)(len0 + done);
      if (X.nFull >= (int)X.C.fullCap) X.overflow |= OVF_FULLPATHS;
      else if (X.fullUsed + clen > X.C.fullPool) X.overflow |= OVF_FULLPOOL;
      else {
        wave_copy_bytes(X.fullPool + X.fullUsed, (const uint8_t*)seq, clen, false);   // (inline: the fast-forward stays a leaf)
        if (l == 0) X.fullMeta[X.nFull] = FullMeta{X.fullUsed, clen, lanc, ranc, dist / ((double)clen + 0.01)};
        X.fullUsed += (clen + 15u) & ~15u;
        X.nFull++;
        WSYNC();
      }
      if (X.overflow) break;
      if (clen > X.refLen) { popped = true; break; }   // :579-582 pop_back: the Trail ends here
      continue;                                        // the record was cut at the aim: next record from the new tip
    }
    if (nOK < TALC_WALK_LEVELS) break;
  }
  flush();
  stepCounter_ = sc0 + (uint32_t)done;
#ifdef TALC_PROF
  if (l == 0) { g_prof[PF_NCALLS] += 1; g_prof[PF_NSTEPS] += (uint32_t)done; }
#endif
  if (done) {
    X.steps += (unsigned long long)done;
    if (l == 0) {
      TrailRec r = r0;
      r.kmer = kmer; r.cnt = cnt; r.dist = dist; r.lanc = lanc; r.ranc = ranc;
      tr_put(X.ia, 0, r);
    }
  }
  if (popped) { pool_free((int)r0.buf); X.ffPopped = true; }
  LSYNC();
  return done;
}

// (the per-step form — tables too large for walk records, or a MIN_COUNT beyond their fields — and the long searches'
//  instances: functions of their own, so that the common one's code stays together)
TALC_DN int fast_forward_steps(int len, uint32_t& stepCounter, uint32_t PATH_MAXLENGTH, bool edge) {
  return uni((int)X.dirRight) ? fast_forward_dir<true>(len, stepCounter, PATH_MAXLENGTH, edge)
                              : fast_forward_dir<false>(len, stepCounter, PATH_MAXLENGTH, edge);
}
TALC_DNC int fast_forward_wide(int len, uint32_t& stepCounter, uint32_t PATH_MAXLENGTH, bool edge) {
  return uni((int)X.dirRight) ? fast_forward_walk<true, true>(len, stepCounter, PATH_MAXLENGTH, edge)
                              : fast_forward_walk<false, true>(len, stepCounter, PATH_MAXLENGTH, edge);
}
TALC_DN int fast_forward(int len, uint32_t& stepCounter, uint32_t PATH_MAXLENGTH, bool edge) {
  // the walk tables encode "count >= MIN_COUNT" for MIN_COUNT below 2^13 - 1 only (kWalkTopNone = 0x1FFF, talc_common.h)
  if (uni((int)(X.T.walkRight != nullptr && X.P.MIN_COUNT < kWalkTopNone)) != 0)
  {
    if (uni((int)X.wideMask) != 0) [[clang::musttail]] return fast_forward_wide(len, stepCounter, PATH_MAXLENGTH, edge);
    return uni((int)X.dirRight) ? fast_forward_walk<true, false>(len, stepCounter, PATH_MAXLENGTH, edge)
                                : fast_forward_walk<false, false>(len, stepCounter, PATH_MAXLENGTH, edge);
  }
  [[clang::musttail]] return fast_forward_steps(len, stepCounter, PATH_MAXLENGTH, edge);
}

// ------------------------------------------------------------------ a fork whose second branch ends at once
// The walk above stops at every tip that has more than one successor in the table.  On a transcriptome whose dump
// carries sequencing-error k-mers (counts 2-3 beside the true k-mer's 30) nearly all of those stops are of two kinds:
//   (a) tagNextNodes (Explorer.cpp:1226-1298) follows only ONE of the successors — the step is a plain step after all;
//   (b) it follows two, and at the NEXT step one of the two children has no successor to follow while the other has
//       exactly one: oneMoreStep (Explorer.cpp:546-612) makes two Trails, drops one a step later, and the search is back
//       to one Trail — two generic steps (~15 000 wave-cycles each: a probe, two children with a buffer copy, aim and
//       cycle tests, set swaps) for what is two plain steps of the surviving Trail.
// fork_step decides both cases from ONE round of probes — lane 4 the tip's successor bucket, lanes 0-3 the buckets of
// its four possible children, in flight together — with the reference's own tagging (tag_next_nodes, the function the
// generic step uses), and commits the one or two steps exactly as the generic steps would leave the surviving Trail:
// k-mer, count, the distance terms |c - n| / sqrt(c) in path order (make_child's expression), bases, step counter, the
// Trail-step statistic (1 + 2 probes for case b).  Whatever is not exactly (a) or (b) with every involved k-mer absent
// from the search's filter (no aim, no possible cycle) returns 0 and the generic step takes the step from the unchanged
// state: more than two followed successors, a dead end, both children alive, limits about to be reached, scoring due
// (MAX_NB_BRANCHES below 2), a k-mer with N, the step trace.  Bridge searches only: an edge search scores two Trails.
// The child that ends is not entered in the filter (the generic step would enter it): the filter stays a superset of
// the live Trail's own k-mers, which is all ThinkIveAlreadyGotThere's exact search relies on.
TALC_DN int fork_step(int len_, uint32_t sc_, uint32_t PMAX_) {
  PROF_DECL;
  PROF_BEGIN();
  const DevParams& P = X.P;
  const int l = lane_id();
  const int len = uni(len_);
  const uint32_t sc = (uint32_t)uni((int)sc_), PMAX = (uint32_t)uni((int)PMAX_);
  if (uni((int)X.traceSteps) != 0 || uni((int)X.noForkStep) != 0) return 0;
  if ((uint32_t)uni((int)P.MAXB) < 2u || (uint32_t)uni((int)P.MAX_INNER_PATHS) < 2u) return 0;
  const TrailRec r0 = tr_get(X.ia, 0);
  if (uni64(r0.nmask) != 0ull) return 0;
  const uint32_t K = (uint32_t)uni((int)P.K), MINC = (uint32_t)uni((int)P.MIN_COUNT), seqCap = (uint32_t)uni((int)X.C.seqCap);
  if (!(sc < PMAX) || (uint32_t)len + 1u > seqCap) return 0;
  const bool dirRight = uni(X.dirRight) != 0;
  const uint64_t kmask = (1ULL << (2 * K)) - 1, m1 = (1ULL << (2 * (K - 1))) - 1;
  const uint64_t kmer = uni64(r0.kmer);
  const uint32_t cnt = (uint32_t)uni((int)r0.cnt);
  auto child_of = [&](uint64_t km, uint32_t b) -> uint64_t {
    return dirRight ? (((km << 2) | (uint64_t)b) & kmask) : (((uint64_t)b << (2 * (K - 1))) | (km >> 2));
  };
  auto get64 = [&](uint64_t v, int src) -> uint64_t {
    return ((uint64_t)(uint32_t)lane_get((int)(uint32_t)(v >> 32), src) << 32) | (uint32_t)lane_get((int)(uint32_t)v, src);
  };
  // ---- one round of loads: the count model's thresholds for the tip's count, lane 4 the tip's successor bucket, lanes
  // b < 4 the HOME bucket of child b's successors (a child that is not followed is never looked at again; one that is
  // nearly always sits in its home bucket)
  bool tipInTable;
  const ModelThresholds mtTip = model_thresholds(cnt, tipInTable);
  if (uni((int)tipInTable) == 0) return 0;   // (a count beyond the threshold table: the generic step has the formula)
  const uint64_t kmMine = (l == 4) ? kmer : child_of(kmer, (uint32_t)(l & 3));
  const uint64_t keyMine = dirRight ? (kmMine & m1) : (kmMine >> 2);
  const Bucket* tab = dirRight ? X.T.right : X.T.left;
  const uint64_t cap = X.T.capacity;
  const uint64_t hMine = table_hash(keyMine);
  uint64_t slotMine = table_slot(hMine, cap);
  BucketRegs br;
  br.key = kEmptyKey; br.cnt[0] = br.cnt[1] = br.cnt[2] = br.cnt[3] = 0u; br.jc01 = br.jc23 = 0u;
  if (l <= 4) br = load_bucket(tab + slotMine);
  bool okTip = false;
  if (l == 4) okTip = probe_bucket_from(tab, cap, keyMine, slotMine, br);
  uint32_t tc[4], tj[4];
  {
    const uint32_t found4 = (uint32_t)lane_get((int)okTip, 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      tc[i] = found4 ? (uint32_t)lane_get((int)br.cnt[i], 4) : 0u;
      tj[i] = found4 ? (uint32_t)lane_get((int)br.jc(i), 4) : 0u;
    }
  }
  int tg[4];
  tag_next_nodes_with(mtTip, P.ERR, MINC, tc, tj, cnt, false, tg, (double*)nullptr);   // (complex: one Trail never exceeds MAX_NB_BRANCHES >= 2)
  int fm = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) if (tg[i] != TAG_NONE && tg[i] != TAG_UNEXPECTED) fm |= 1 << i;
  fm = uni(fm);
  const int nF = __builtin_popcount((unsigned)fm);
  int taken = 0;
  if (nF == 1 || nF == 2) {
    // (the walk also stops at a tip with ONE successor when the filter has seen that successor — an aim, a possible
    //  cycle: the query below then says "maybe" and the generic step takes it, as it must)
    // aim / cycle: the followed children against the search's filter (it holds the aims, init_first_trail); the hash of a
    // k-mer's successor key is the hash its bucket was addressed with
    const bool mine = (l < 4) && (((fm >> l) & 1) != 0);
    const uint32_t hvMine = (uint32_t)(hMine >> 32);
    bool maybe = false;
    if (mine) { uint32_t hq; maybe = bloom_query(kmMine, 0ull, hq); }
    const int iA = __builtin_ctz((unsigned)fm);
    int iS = iA, bC = 0;
    uint32_t cS = 0, cC = 0, hvS = 0, hvC = 0;
    uint64_t kmS = 0, kmC = 0;
    bool ok = ballot64(mine && maybe) == 0ull;
    if (ok && nF == 2) {
      ok = (sc + 1u < PMAX) && ((uint32_t)len + 2u <= seqCap);
      if (ok) {
        // the step after the fork: a child with NO successor in the table ends there, a child with exactly ONE is followed
        // whatever the count model says (tagNextNodes: counter == 1, Explorer.cpp:1251); anything else is not this case
        bool found = false;
        if (mine) found = probe_bucket_from(tab, cap, keyMine, slotMine, br);
        uint32_t nSucc = 0, which = 0, cSucc = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) if (found && br.cnt[i] >= MINC) { ++nSucc; which = (uint32_t)i; cSucc = br.cnt[i]; }
        const int iB = __builtin_ctz((unsigned)(fm & (fm - 1)));
        const int nA = lane_get((int)nSucc, iA), nB = lane_get((int)nSucc, iB);
        ok = ((nA == 1) & (nB == 0)) | ((nA == 0) & (nB == 1));
        if (ok) {
          iS = nA ? iA : iB;
          bC = lane_get((int)which, iS);
          cC = (uint32_t)lane_get((int)cSucc, iS);
        }
      }
    }
    if (ok) {
      cS = (iS == 0) ? tc[0] : (iS == 1) ? tc[1] : (iS == 2) ? tc[2] : tc[3];
      kmS = get64(kmMine, iS);
      hvS = (uint32_t)lane_get((int)hvMine, iS);
      if (nF == 2) {
        kmC = child_of(kmS, (uint32_t)bC);
        ok = !bloom_query(kmC, 0ull, hvC);
        hvC = (uint32_t)uni((int)hvC);
      }
    }
    if (ok) {
      // ---- commit: the surviving Trail as the generic steps would leave it
      const uint32_t wm = (uint32_t)uni((int)X.wideMask);
      gu8 seq = (gu8)uni_ptr(X.seqPool + (uint64_t)r0.buf * X.C.seqCap);
      if (l == 0) {
        TrailRec r = r0;
        seq[len] = (uint8_t)iS;
        r.dist = r0.dist + fabs((double)cnt - (double)cS) / sqrt((double)cnt);
        r.kmer = kmS; r.cnt = cS;
        atomicOr(&g_bloom[hvS >> 25], (1ull << ((hvS >> 19) & 63u)) | (1ull << ((hvS >> 13) & 63u)));
        if (wm != 0u) wide_or(X.wideBloom + wide_word(hvS, wm), wide_bits(hvS));
        if (nF == 2) {
          seq[len + 1] = (uint8_t)bC;
          r.dist = r.dist + fabs((double)cS - (double)cC) / sqrt((double)cS);
          r.kmer = kmC; r.cnt = cC;
          atomicOr(&g_bloom[hvC >> 25], (1ull << ((hvC >> 19) & 63u)) | (1ull << ((hvC >> 13) & 63u)));
          if (wm != 0u) wide_or(X.wideBloom + wide_word(hvC, wm), wide_bits(hvC));
        }
        tr_put(X.ia, 0, r);
      }
      taken = nF;
      X.steps += (nF == 2) ? 3ull : 1ull;   // Trails probed: one at the fork, two at the step after it
      LSYNC();
    }
  }
#ifdef TALC_PROF
  if (l == 0) g_prof[taken == 1 ? PF_FK1 : taken == 2 ? PF_FK2 : PF_FKBAIL] += 1;
#endif
  PROF_END(PF_FORK);
  return taken;
}

// first Trail of a search: the start anchor (Trail.cpp:57-65)
// A bridge's search also enters its aims (the target anchors, Explorer.cpp:920) in the search's filter: the
// fast-forward loop then needs no aim comparison of its own (a step onto an aim is a filter hit).
TALC_D void init_first_trail(const AnchorRec& a, const AnchorRec* inList, bool withAims, uint32_t pathMax) {
  const int K = (int)X.P.K;
  // the first Trail's count (Trail.cpp:57-65: the coverage at the anchor): two dependent loads, word then pair, requested
  // here and consumed at the end, behind everything else this function does
  uint32_t firstCount;
  {
    const uint32_t h = (uint32_t)uni((int)(X.dirRight ? X.LH : X.RH)), rs0 = (uint32_t)uni((int)(X.dirRight ? X.Ls : X.Rs));
    if ((h & kRegClean) != 0u && a.pos >= rs0) firstCount = (*(const v2u32 TALC_AS1*)((const uint2 TALC_AS1*)X.cov + (h & ~kRegClean) + (a.pos - rs0))).x;
    else firstCount = COVX(a.pos);
  }
  (void)inList;
  pool_reset();
  const uint32_t b0 = (uint32_t)pool_alloc();
  {   // the anchor's K <= 31 bases in growth order: one base per lane
    const int l0 = lane_id();
    if (l0 < K) ((gu8)(X.seqPool + (uint64_t)b0 * X.C.seqCap))[l0] = ((gcu8)X.read)[a.pos + (uint32_t)(X.dirRight ? l0 : K - 1 - l0)];
  }
  // a bridge search's first buffer starts without a kept alignment row.  Every other buffer a search hands out gets its
  // record from row_copy (branch_copy, garden), so no record is ever read that this search has not written: the stamps in
  // a record's end words (row_covered) are a second line of defence, not the guarantee.
  if (withAims && (uint32_t)uni((int)X.rowAvail) != 0u) row_set_covered(b0, 0u);
  // the search's cycle filter: the wave's 8192 bits of LDS, or — for a Trail that may grow to thousands of k-mers, and
  // when the walk-table form of the fast-forward (the one that can read it) is in use — the wide one in HBM
  uint32_t wm = 0;
  if (pathMax > (uint32_t)WIDE_BLOOM_MIN_PATH && X.wideBloom != nullptr && X.T.walkRight != nullptr && X.P.MIN_COUNT < kWalkTopNone) {
    uint32_t words = 1024;
    while (words < (uint32_t)WIDE_BLOOM_WORDS && words < pathMax / 2) words <<= 1;
    wm = words - 1;
    for (uint32_t i = (uint32_t)lane_id(); i < words; i += 64) X.wideBloom[i] = 0ull;
    WSYNC();
  }
  X.wideMask = (uint32_t)uni((int)wm);
  if (lane_id() == 0) g_keep.owner = 0u;   // (no wavefront of an earlier search is this one's)
  g_bloom[lane_id()] = 0ull; g_bloom[lane_id() + 64] = 0ull;
  LSYNC();
  if (withAims) {
    const AnchorRec* aims = X.dirRight ? X.ancR : X.ancL;
    const int nAims = X.dirRight ? X.nAncR : X.nAncL;
    for (int ai = lane_id(); ai < nAims; ai += 64) {
      // (the first AIMS_LDS aims are in LDS already, search_bridge put them there)
      const uint64_t tk = ai < AIMS_LDS ? g_aimK[ai] : aims[ai].kmer, tn = ai < AIMS_LDS ? g_aimN[ai] : aims[ai].nmask;
      if (tn != 0ull) continue;   // a fast-forwarded tip never holds an N
      const uint64_t h = bloom_hash(tk, 0ull);
      if (wm != 0u) wide_or(X.wideBloom + wide_word((uint32_t)(h >> 32), wm), wide_bits((uint32_t)(h >> 32)));
      atomicOr(&g_bloom[bloom_word(h)], bloom_mask(h));
    }
    LSYNC();
  }
  bloom_query_insert(a.kmer, a.nmask);
  if (lane_id() == 0) {
    TrailRec r;
    r.kmer = a.kmer; r.nmask = a.nmask; r.cnt = firstCount; r.score = 0; r.fail = 0; r.dist = 0.0;
    r.lanc = X.dirRight ? (int)a.pos : -1;
    r.ranc = X.dirRight ? -1 : (int)a.pos;
    r.buf = b0;
    tr_put(X.ia, 0, r);
  }
  WSYNC();
}

// append read[from, to) to the reference buffer in growth order
TALC_D void ref_append(uint32_t from, uint32_t to) {
  if (to <= from) return;
  const uint32_t n = to - from;
  if (X.refLen + n > X.C.refCap) { X.overflow |= OVF_SEQ; return; }
  copy_bytes(X.refBuf + X.refLen, X.read + from, n, !X.dirRight);
  X.refLen += n;
}

struct GapResult { bool found; uint32_t Le, Rs; uint32_t wOff, wLen; };

// an anchor record in scalar registers (it is the same for every lane, and it lives across the whole search)
TALC_D AnchorRec uni_anchor(const AnchorRec* p) {
  const AnchorRec a = *p;
  AnchorRec r;
  r.kmer = uni64(a.kmer); r.nmask = uni64(a.nmask); r.pos = (uint32_t)uni((int)a.pos); r.count = (uint32_t)uni((int)a.count);
  return r;
}

// findBestBridge (Trajectory.cpp:282-303) over several scored candidates (gScores / fullMeta): the first best score,
// then the largest distance among the later candidates that tie with it.  (Cold: config 2 never records two bridges.)
TALC_DNC int find_best_bridge(int nCand_) {
  const int nCand = uni(nCand_);
  int index = 0;
  for (int i = 1; i < nCand; ++i) if (X.gScores[i] > X.gScores[index]) index = i;
  const int first = index;
  for (int i = first + 1; i < nCand; ++i)
    if (X.gScores[i] == X.gScores[first] && X.fullMeta[i].dist > X.fullMeta[index].dist) index = i;
  return index;
}

// Explorer::searchBridge (Explorer.cpp:868-989) after initializeINNER(…, direction)
TALC_D bool search_bridge(uint32_t& weakOutOff, uint32_t& weakOutLen, uint32_t& weakUsed) {
  PROF_DECL; PROF_DECL2;
  const DevParams& P = X.P;
  const uint32_t K = (uint32_t)uni((int)P.K);
  const int l = lane_id();
  const AnchorRec* anchors = uni_ptr(X.dirRight ? X.ancL : X.ancR);
  const int nAnch = uni(X.dirRight ? X.nAncL : X.nAncR);
  const int limit = min(nAnch, uni((int)P.MAX_START_ANCHORS));
  bool found = false;
  {  // the target anchors ("aims") go to LDS for the per-child aim check
    const AnchorRec* aims = X.dirRight ? X.ancR : X.ancL;
    const int nAims = X.dirRight ? X.nAncR : X.nAncL;
    if (l < min(nAims, AIMS_LDS)) { g_aimK[l] = aims[l].kmer; g_aimN[l] = aims[l].nmask; g_aimPos[l] = aims[l].pos; }
    LSYNC();
  }
  for (int s = 0; s < limit && !found; ++s) {
#ifdef TALC_PROF
    const unsigned long long _pf_b0 = __builtin_amdgcn_s_memrealtime();
#endif
    const AnchorRec a = uni_anchor(anchors + s);
    const uint32_t whichStart = a.pos;
    const uint32_t Rs = (uint32_t)uni((int)X.Rs), Re = (uint32_t)uni((int)X.Re), Ls = (uint32_t)uni((int)X.Ls), Le = (uint32_t)uni((int)X.Le);
    X.nFull = 0; X.fullUsed = 0;
    uint32_t stepCounter = 0;
    // gap between the start anchor and the target region (:916-918)
    uint32_t gapLen = 0;
    X.refLen = 0;
    X.ref = X.refBuf;
    PROF_BEGIN();
    if (X.dirRight && whichStart + K <= Rs) {
      // walking RIGHT the reference — anchor, gap, target region (:916-936) — is a stretch of the read as it stands:
      // no copy, the search reads it in place (growth order = text order)
      gapLen = Rs - (whichStart + K);
      X.ref = const_cast<uint8_t*>(X.read) + whichStart;
      X.refLen = Re + K - whichStart;
    } else if (X.dirRight) {
      ref_append(whichStart, whichStart + K);
      if (whichStart + K < Rs) { gapLen = Rs - (whichStart + K); ref_append(whichStart + K, Rs); }
      ref_append(Rs, Re + K);
    } else {
      ref_append(whichStart, whichStart + K);
      if (Le + K < whichStart) { gapLen = whichStart - (Le + K); ref_append(Le + K, whichStart); }
      ref_append(Ls, Le + K);
      // growth order walking LEFT = reversed text: anchor, then gap, then target, each reversed,
      // which is exactly the reverse of target+gap+anchor (:934-936)
    }
    WSYNC();
    PROF_END(PF_REFB);
    const uint32_t PATH_MAXLENGTH = (uint32_t)(int)(1.2 * (double)gapLen + (double)(3 * K));
    if (!pool_shape(PATH_MAXLENGTH)) return false;
    rows_shape();
    PROF_BEGIN2(); init_first_trail(a, anchors + s, true, PATH_MAXLENGTH); PROF_END2(PF_INITTR);
    int nCur = 1;
    int len = (int)K;
    const uint32_t maxInner = (uint32_t)uni((int)P.MAX_INNER_PATHS);
    while ((int)(nCur > 0) & (int)((uint32_t)nCur <= maxInner) & (int)((uint32_t)uni((int)stepCounter) < PATH_MAXLENGTH) & (int)(uni((int)X.overflow) == 0)) {
      if (nCur == 1) {
        PROF_BEGIN2(); len += uni(fast_forward(len, stepCounter, PATH_MAXLENGTH, false)); PROF_END2(PF_FFWD);
        if (X.ffPopped) { X.ffPopped = false; nCur = 0; break; }   // its last step recorded a bridge and ended the Trail
        if (!(stepCounter < PATH_MAXLENGTH)) break;
        // where the walk stopped: a fork whose second branch ends at once is two more plain steps (fork_step)
        const int fs = uni(fork_step(len, stepCounter, PATH_MAXLENGTH));
        if (fs > 0) { len += fs; stepCounter += (uint32_t)fs; continue; }
      }
      PROF_BEGIN2();
      nCur = uni(step_bridge(nCur, len, stepCounter));
      PROF_END2(PF_STEPB);
      ++len;
      if (uni((int)(X.traceSteps))) trace_rec(TR_STEP, (int)stepCounter, nCur, X.nFull, 0, 0.0, nullptr, 0, false);
    }
#ifdef TALC_PROF
    if (l == 0) {   // (the walk of this start anchor, without the evaluation of what it recorded)
      const uint32_t dt = (uint32_t)(__builtin_amdgcn_s_memrealtime() - _pf_b0);
      g_prof[PF_RBR] += 1; if (dt > g_prof[PF_RBRMAX]) g_prof[PF_RBRMAX] = dt;
    }
#endif
    if (X.overflow) return false;
    if (X.nFull > 0) {
      // :945-960 score every recorded bridge, cut its anchors; quirk :955-960 keeps the FIRST nOK
      int nOK = 0;
      const uint32_t limit2 = X.Re;
      // per full path: score (edit distance), idscore, ok, new right anchor, cut (off,len)
      // stored in gScores (score), gDists (idscore) and gKept (ok | cutLen<<1), ranc updated in meta
      PROF_BEGIN();
      FullMeta lone = FullMeta();
      uint32_t loneFlags = 0;
      unsigned long long loneIdBits = 0;
      for (int t = 0; t < X.nFull; ++t) {
        const FullMeta fm = X.fullMeta[t];
        const uint8_t* ps = X.fullPool + fm.off;
        double score, idv;
        // computeEditDistance / computeIDScore (Trajectory.cpp:386-428, 337-384): both non-empty here
        int es, lcs;
        // one candidate: its score is never compared, and its identity only with MIN_INNER (Explorer.cpp:973): the
        // least LCS that passes, T, is enough — lcs >= T  <=>  (double)lcs / maxLen >= MIN_INNER (monotone in lcs)
        int accept = 0;
        bool needed = true;
        if (X.nFull == 1 && P.MIN_INNER > 0.0) {
          const double mxl = (double)max(X.refLen, fm.len);
          int T = (int)ceil(P.MIN_INNER * mxl);
          while (T > 1 && (double)(T - 1) / mxl >= P.MIN_INNER) --T;
          while ((double)T / mxl < P.MIN_INNER) ++T;
          accept = T;
          // ... and only if its length passes: Explorer.cpp:973 and-s the two tests, and a lone candidate whose cut
          // length fails the first one is rejected whatever its identity is (same arithmetic as the test below)
          uint32_t cutLen1 = 0; bool ok1 = true;
          if (fm.len >= 2 * K) cutLen1 = fm.len - 2 * K;
          else if (fm.len > K) ok1 = ((uint32_t)fm.ranc + 2 * K - fm.len <= limit2);
          else ok1 = false;
          const uint32_t bestLen1 = ok1 ? cutLen1 : 0u;
          const double diff1 = (double)X.weakLen - (double)bestLen1;
          needed = ok1 && ((diff1 < X.weakLen * 0.05) || ((X.weakLen < 6) & (bestLen1 < 6)));
        }
        es = 0; lcs = 0;
        if (uni((int)needed)) {
          int quick = -1;
          if (X.nFull == 1 && accept > 0) quick = uni(lcs_reaches(X.ref, (int)X.refLen, ps, (int)fm.len, accept));
          if (quick >= 0) lcs = quick;
          else edit_and_lcs(X.ref, (int)X.refLen, ps, (int)fm.len, es, lcs, X.nFull > 1, accept);
        }
        score = (double)es;
        idv = (double)lcs / (double)max(X.refLen, fm.len);
        // cutAnchors INNER (Trajectory.cpp:176-197)
        bool ok = true;
        uint32_t cutLen = 0;
        uint32_t ranc = (uint32_t)fm.ranc;
        if (fm.len >= 2 * K) cutLen = fm.len - 2 * K;
        else if ((fm.len < 2 * K) & (fm.len > K)) {
          if (ranc + 2 * K - fm.len <= limit2) ranc = ranc + 2 * K - fm.len;
          else ok = false;
        } else ok = false;
        if (X.nFull == 1) {   // a lone candidate: its figures stay in (scalar) registers, no trip through memory
          lone = fm; lone.ranc = (int32_t)ranc;
          loneFlags = (ok ? 1u : 0u) | (cutLen << 1);
          loneIdBits = uni64((unsigned long long)__double_as_longlong(idv));
        } else if (l == 0) {
          X.gScores[t] = score; X.gDists[t] = idv; X.gKept[t] = (ok ? 1u : 0u) | (cutLen << 1);
          X.fullMeta[t].ranc = (int32_t)ranc;
        }
        nOK += ok ? 1 : 0;
      }
      if (X.nFull > 1) WSYNC();
      PROF_END(PF_EVALFULL);
      const int nCand = (nOK == X.nFull) ? X.nFull : nOK;   // :955-960
      PROF_BEGIN();
      if (nCand > 0) {
        // findBestBridge (Trajectory.cpp:282-303)
        FullMeta bm;
        uint32_t flags;
        double bestId;
        if (X.nFull == 1) {
          bm.off = (uint32_t)uni((int)lone.off); bm.len = (uint32_t)uni((int)lone.len); bm.lanc = uni(lone.lanc); bm.ranc = uni(lone.ranc); bm.dist = 0.0;
          flags = (uint32_t)uni((int)loneFlags);
          bestId = __longlong_as_double((long long)loneIdBits);
        } else {
          const int index = uni(find_best_bridge(nCand));
          bm = X.fullMeta[index];
          flags = X.gKept[index];
          bestId = X.gDists[index];
        }
        // a path whose cutAnchors failed has an empty sequence (truncSeq stays empty)
        const uint32_t bestLen = (flags & 1u) ? (flags >> 1) : 0u;
        const double diff = (double)X.weakLen - (double)bestLen;
        if (((diff < X.weakLen * 0.05) || ((X.weakLen < 6) & (bestLen < 6))) & (bestId >= P.MIN_INNER)) {
          X.Le = (uint32_t)bm.lanc;
          X.Rs = (uint32_t)bm.ranc;
          // weak sequence := path without its two anchors, stored in natural orientation
          if (weakUsed + bestLen > X.C.weakPool) { X.overflow |= OVF_WEAKPOOL; return false; }
          if (bestLen) copy_bytes(X.weak + weakUsed, X.fullPool + bm.off + K, bestLen, !X.dirRight);
          weakOutOff = weakUsed; weakOutLen = bestLen;
          weakUsed += bestLen;
          WSYNC();
          found = true;
        }
      }
      PROF_END(PF_RESULT);
    }
  }
  return found;
}

// Explorer::searchEdge (Explorer.cpp:992-1081) after initializeHEAD / initializeTAIL
// one start anchor of an edge search: the body of searchEdge's loop (Explorer.cpp:1026-1055).  What it records is folded
// into X.best2 / X.edgeLong / X.edgeShort by record_edge; X.overflow set: the search is void.
TALC_D void edge_anchor_search(const AnchorRec* anchors_, int s_) {
  PROF_DECL2;
#ifdef TALC_PROF
  const unsigned long long _pf_a0 = __builtin_amdgcn_s_memrealtime();
#endif
  const DevParams& P = X.P;
  const AnchorRec* anchors = uni_ptr(anchors_);
  const int s = uni(s_);
  const uint32_t K = (uint32_t)uni((int)P.K);
  const uint32_t readLen = (uint32_t)uni((int)X.L), maxInner = (uint32_t)uni((int)P.MAX_INNER_PATHS), CHECK = (uint32_t)uni((int)P.CHECK_INTERVAL);
  const AnchorRec a = uni_anchor(anchors + s);
  const uint32_t whichStart = a.pos;
  uint32_t stepCounter = 0;
  int xdrop = (int)((double)(int)P.CHECK_INTERVAL * P.FAILURE_RATE + 1.0);   // :1031
  // currentGap = extractWeakBorderSequence(seq, whichStart, K, location) (:1035)
  uint32_t gapLen;
  X.refLen = 0;
  X.ref = X.refBuf;
  if (X.dirRight) {   // TAIL: anchor + suffix(seq, whichStart+K): the read's own tail, read in place
    gapLen = readLen - (whichStart + K);
    X.ref = const_cast<uint8_t*>(X.read) + whichStart;
    X.refLen = readLen - whichStart;
  } else {            // HEAD: prefix(seq, whichStart) + anchor
    gapLen = whichStart;
    ref_append(whichStart, whichStart + K);
    ref_append(0, whichStart);
  }
  WSYNC();
  const uint32_t PATH_MAXLENGTH = (uint32_t)(int)(1.2 * (double)gapLen + (double)(2 * K));
  if (!pool_shape(PATH_MAXLENGTH)) return;
  PROF_BEGIN2(); init_first_trail(a, anchors + s, false, PATH_MAXLENGTH); PROF_END2(PF_INITTR);
  int nCur = 1;
  int len = (int)K;
  while ((int)(nCur > 0) & (int)((uint32_t)nCur <= maxInner) & (int)((uint32_t)uni((int)stepCounter) < PATH_MAXLENGTH) & (int)(uni((int)X.overflow) == 0)) {
    if (nCur == 1) {
      PROF_BEGIN2();
      const int ff = uni(fast_forward(len, stepCounter, PATH_MAXLENGTH, true));
      len += ff;
      PROF_END2(PF_FFWD);
      if (ff > 0 && ((uint32_t)uni((int)stepCounter) % CHECK == 0)) {
        // the fast-forward took the step after which scoring is due (Explorer.cpp:672-686): the one Trail stays
        // where it is (set ia, slot 0) — score it there; five or fewer survivors means no gardening
        PROF_BEGIN2();
        nCur = uni(score_edges(X.ia, 1, len, xdrop));
        PROF_END2(PF_STEPE);
        continue;
      }
      if (!((uint32_t)uni((int)stepCounter) < PATH_MAXLENGTH)) break;
    }
    PROF_BEGIN2();
    nCur = uni(step_edge(nCur, len, stepCounter, PATH_MAXLENGTH, xdrop));
    PROF_END2(PF_STEPE);
    ++len;
    if (uni((int)(X.traceSteps)))
      trace_rec(TR_STEP, (int)stepCounter, nCur, X.nEdges, xdrop, 0.0, nullptr, 0, false);
  }
#ifdef TALC_PROF
  if (lane_id() == 0) {
    const unsigned long long dt = __builtin_amdgcn_s_memrealtime() - _pf_a0;
    const int bk = dt < 100000ull ? 0 : dt < 400000ull ? 1 : dt < 1600000ull ? 2 : dt < 6400000ull ? 3 : 4;
    g_prof[PF_EA0 + bk] += 1;
    if (bk == 3) g_prof[PF_EASUM3] += (uint32_t)dt;
    if (bk == 4) g_prof[PF_EASUM4] += (uint32_t)dt;
    g_prof[PF_RANCH] += 1; if ((uint32_t)dt > g_prof[PF_RANCHMAX]) g_prof[PF_RANCHMAX] = (uint32_t)dt;
  }
#endif
}

// (the copy the edge tasks call: search_edge has the body inline, as its loop always had)
TALC_DN void edge_anchor_search_task(const AnchorRec* anchors, int s) { edge_anchor_search(anchors, s); }

// ------------------------------------------------------------------ edge tasks
// The start anchors of searchEdge (Explorer.cpp:1026: at most MAX_START_ANCHORS of them) are independent searches: each
// starts from its own first Trail, what it records goes through recordEdge into m_shortPaths / m_longPaths, of which
// sortOutBestBorder keeps the FIRST best by (score, distance) — a fold, so the best of every anchor, merged in anchor order
// under the same strict comparison, is the sequential result (m_complexRegion is an OR).  A read with a 500-base head
// and tail is ten such searches and, over a branching graph, a tenth of a second of ONE wave: the launch ended when the
// heaviest of them did, with two thirds of the waves idle.  So: a wave about to start an anchor publishes the anchors its
// search still has to do in its box (one per wave slot, in HBM) — from the search's start for a long border (taskHeavy;
// its anchors take tens of milliseconds each), otherwise once the work queue has run dry (taskMinWeak) —, waves between
// two reads and the waves that stay when the reads are gone claim them one at a time, run them in their own scratch and
// write the anchor's best long / best short candidate back; the owner claims from the same counter, waits for the
// claimed ones, and merges in anchor order.  (What did not work on the way to this form: docs/results_log_r04.md.)
//   avail[slot]   unclaimed anchors of the slot's box (claim = atomic decrement; the claimed anchor is limit - old value)
//   hdr.done      anchors finished (owner waits for claimed == done)
//   qwords[256]   reads finished (a wave without a read leaves when this reaches n_work: nothing can be published any more)
//   qwords[512]   unclaimed anchors over all boxes (a hint: whether scanning avail[] is worth it)
// Box traffic is agent-scope dword atomics (the L2 of another XCD is not coherent with this one's for plain accesses)
// between agent-scope fences.  A thief never waits for anything, so every wait here is bounded by one anchor's search.
#define EDGE_BOX_ANCHORS 8
struct EdgeBoxHdr {
  const uint8_t* read; const uint2* cov; const CovWord* covw; double lambda;
  uint32_t L, n, location, dirRight, Ls, Le, Rs, Re, LH, RH, weakLen, limit;
  uint32_t done, first, pad[2];
  AnchorRec anc[EDGE_BOX_ANCHORS];
};
struct EdgeBoxRes { EdgeCand best[2]; uint32_t complex, overflow, redo, pad; };
static_assert(sizeof(EdgeCand) == 48 && sizeof(EdgeBoxRes) == 112 && sizeof(AnchorRec) == 24, "edge box records are moved as dwords");
static_assert(offsetof(EdgeBoxHdr, anc) == 96 && sizeof(EdgeBoxHdr) == 96 + 24 * EDGE_BOX_ANCHORS, "edge box header layout");
static const uint32_t kBoxResOff = 320, kBoxResStride = 128, kBoxSeqOff = kBoxResOff + EDGE_BOX_ANCHORS * kBoxResStride;
__host__ __device__ inline uint64_t edge_box_bytes(uint32_t seqCap) { return (kBoxSeqOff + 2ull * EDGE_BOX_ANCHORS * seqCap + 255ull) / 256ull * 256ull; }

// the launch's edge-task arguments (host -> k_search)
struct EdgeTaskArgs {
  uint8_t* boxes; uint32_t* avail;   // nullptr: no edge tasks in this launch
  uint32_t seqCap;                   // bytes per box sequence
  uint32_t minWeak;                  // a border of at least this many bases is published once the work queue is dry ...
  uint32_t heavy, heavyRounds;       // ... and one of at least `heavy` bases from the start of its search, for the reads of the queue's first `heavyRounds` rounds
  uint32_t lingerMod;                // one wave in so many stays when the reads are gone
  uint32_t test;                     // TALC_TEST_EDGE_REDO
  const uint32_t* autoSwitch;        // not null: the tasks are used when this word is not 0 (k_order_scale: the batch's graph branches)
};
TALC_D uint32_t aload32(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
TALC_D void astore32(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
TALC_D uint32_t aadd32(uint32_t* p, uint32_t v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
TALC_D uint32_t axchg32(uint32_t* p, uint32_t v) { return __hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
TALC_D void agent_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent"); }
TALC_D uint8_t* edge_box(uint32_t slot) { return X.boxes + (uint64_t)slot * X.boxBytes; }
TALC_D uint32_t* edge_box_seq(uint8_t* box, int s, int bi) { return (uint32_t*)(box + kBoxSeqOff + (uint64_t)(2 * s + bi) * X.boxSeqCap); }

// X.best2 and its two sequences -> result s of the box (g_dp, free between searches, is the staging area)
TALC_DNC void edge_box_store(uint8_t* box_, int s_, uint32_t complexFlag_) {
  uint8_t* box = uni_ptr(box_); const int s = uni(s_);
  const int l = lane_id();
  LSYNC();
  EdgeBoxRes* st = (EdgeBoxRes*)g_dp;
  const uint32_t cap = (uint32_t)uni((int)X.boxSeqCap);
  const bool tooLong = (X.best2[0].have && X.best2[0].len > cap) || (X.best2[1].have && X.best2[1].len > cap);
  if (l == 0) {
    st->best[0] = X.best2[0]; st->best[1] = X.best2[1];
    st->complex = uni((int)complexFlag_); st->overflow = X.overflow; st->redo = (X.nanSeen != 0u || tooLong) ? 1u : 0u; st->pad = 0;
  }
  LSYNC();
  uint32_t* dst = (uint32_t*)(box + kBoxResOff + (uint32_t)s * kBoxResStride);
  if (l < (int)(sizeof(EdgeBoxRes) / 4)) astore32(dst + l, (uint32_t)g_dp[l]);
  if (!tooLong) {
    for (int bi = 0; bi < 2; ++bi) {
      if (!X.best2[bi].have) continue;
      const uint32_t nw = (X.best2[bi].len + 3u) >> 2;
      const uint32_t* src = (const uint32_t*)uni_ptr(bi == 0 ? X.edgeLong : X.edgeShort);
      uint32_t* q = uni_ptr(edge_box_seq(box, s, bi));
      for (uint32_t i = (uint32_t)l; i < nw; i += 64) astore32(q + i, src[i]);
    }
  }
  LSYNC();
}
// result s of the box -> g_dp (read it there as an EdgeBoxRes)
TALC_D void edge_box_load(uint8_t* box, int s) {
  const int l = lane_id();
  LSYNC();
  const uint32_t* src = (const uint32_t*)(box + kBoxResOff + (uint32_t)s * kBoxResStride);
  if (l < (int)(sizeof(EdgeBoxRes) / 4)) g_dp[l] = (int)aload32(src + l);
  LSYNC();
}
TALC_D void edge_box_take(uint8_t* box, int s, int bi, uint32_t len) {   // the sequence of a candidate back into the wave's own buffer
  uint32_t* dst = (uint32_t*)uni_ptr(bi == 0 ? X.edgeLong : X.edgeShort);
  const uint32_t* q = uni_ptr(edge_box_seq(box, s, bi));
  const uint32_t nw = (len + 3u) >> 2;
  for (uint32_t i = (uint32_t)lane_id(); i < nw; i += 64) dst[i] = aload32(q + i);
}
// results first .. limit-1 merged in order into X.best2 and the wave's two sequence buffers
TALC_D void edge_box_fold(uint8_t* box, int first, int limit) {
  EdgeCand best0, best1; best0.have = false; best1.have = false;
  int win0 = -1, win1 = -1;
  for (int s = first; s < limit; ++s) {
    edge_box_load(box, s);
    const EdgeBoxRes* r = (const EdgeBoxRes*)g_dp;
    const EdgeCand c0 = r->best[0], c1 = r->best[1];
    if (c0.have && (!best0.have || c0.score > best0.score || (c0.score == best0.score && c0.dist > best0.dist))) { best0 = c0; win0 = s; }
    if (c1.have && (!best1.have || c1.score > best1.score || (c1.score == best1.score && c1.dist > best1.dist))) { best1 = c1; win1 = s; }
  }
  LSYNC();
  X.best2[0] = best0; X.best2[1] = best1;
  if (uni(win0) >= 0) edge_box_take(box, uni(win0), 0, best0.len);
  if (uni(win1) >= 0) edge_box_take(box, uni(win1), 1, best1.len);
  WSYNC();
}

TALC_D bool queue_is_dry() {
  uint32_t v = 0;
  if (lane_id() == 0) v = aload32(X.qwords);
  return (uint32_t)uni((int)v) >= X.nWork;
}

// the read and the edge search a box describes -> X (g_dp is the staging area); returns the search's number of anchors
TALC_D int edge_ctx_load(uint8_t* box) {
  const int l = lane_id();
  const uint32_t* box32 = (const uint32_t*)box;
  LSYNC();
  if (l < 24) g_dp[l] = (int)aload32(box32 + l);
  LSYNC();
  const EdgeBoxHdr* h = (const EdgeBoxHdr*)g_dp;
  const int limit = uni((int)h->limit);
  X.read = h->read; X.cov = h->cov; X.covw = h->covw; X.lambda = h->lambda; X.L = h->L; X.n = h->n;
  X.location = (int)h->location; X.dirRight = (int)h->dirRight;
  X.Ls = h->Ls; X.Le = h->Le; X.Rs = h->Rs; X.Re = h->Re; X.LH = h->LH; X.RH = h->RH; X.weakLen = h->weakLen;
  X.ffPopped = false; X.tracing = false; X.traceSteps = false;
  X.nAncL = X.dirRight ? limit : 0; X.nAncR = X.dirRight ? 0 : limit;
  LSYNC();
  return limit;
}

TALC_DNC bool edge_task_steal(int tries);

// Anchors s0 .. limit-1 of the current edge search as tasks; X.best2 holds the fold of anchors 0 .. s0-1.
// 1: X.best2 / edgeLong / edgeShort hold the fold of all anchors; 0: the fold of 0 .. s0-1 is back in place and the caller
// goes on in order (or X.overflow is set)
TALC_DNC int edge_anchors_as_tasks(const AnchorRec* anchors_, int s0_, int limit_) {
  const AnchorRec* anchors = uni_ptr(anchors_);
  const int s0 = uni(s0_), limit = uni(limit_);
  const int l = lane_id();
  const uint32_t me = (uint32_t)uni((int)X.mySlot);
  uint8_t* box = uni_ptr(edge_box(me));
  uint32_t* box32 = (uint32_t*)box;
  uint32_t* done = box32 + offsetof(EdgeBoxHdr, done) / 4;
  const int first = s0 > 0 ? s0 - 1 : s0;
  if (s0 > 0) {   // the fold so far takes the place of the last anchor it covers (if a box sequence holds it)
    const uint32_t cap = (uint32_t)uni((int)X.boxSeqCap);
    if ((X.best2[0].have && X.best2[0].len > cap) || (X.best2[1].have && X.best2[1].len > cap)) return 0;
    edge_box_store(box, s0 - 1, 0u);
  }
  LSYNC();
  {
    EdgeBoxHdr* h = (EdgeBoxHdr*)g_dp;
    if (l == 0) {
      h->read = X.read; h->cov = X.cov; h->covw = X.covw; h->lambda = X.lambda;
      h->L = X.L; h->n = X.n; h->location = (uint32_t)X.location; h->dirRight = (uint32_t)X.dirRight;
      h->Ls = X.Ls; h->Le = X.Le; h->Rs = X.Rs; h->Re = X.Re; h->LH = X.LH; h->RH = X.RH; h->weakLen = X.weakLen; h->limit = (uint32_t)limit;
      h->done = 0; h->first = (uint32_t)s0; h->pad[0] = 0; h->pad[1] = 0;
    }
    LSYNC();
    if (l < 24) astore32(box32 + l, (uint32_t)g_dp[l]);
    for (int i = l; i < 6 * limit; i += 64) astore32(box32 + 24 + i, ((const uint32_t*)anchors)[i]);
    LSYNC();
  }
  agent_fence();
  const uint32_t nTasks = (uint32_t)(limit - s0);
  if (l == 0) { aadd32(X.qwords + kQueueOpen, nTasks); (void)axchg32(X.avail + me, nTasks); }
  PROF_COUNT(PF_TPUB, 1);
  uint32_t nMine = 0;
  while (true) {
    int old = 0;
    if (l == 0) old = (int)aadd32(X.avail + me, 0xFFFFFFFFu);
    old = uni(old);
    if (old <= 0) break;
    if (l == 0) (void)aadd32(X.qwords + kQueueOpen, 0xFFFFFFFFu);
    ++nMine;
    PROF_COUNT(PF_TOWN, 1);
    const int s = limit - old;
    X.best2[0].have = false; X.best2[1].have = false; X.nanSeen = 0u;
    edge_anchor_search_task(anchors, s);
    edge_box_store(box, s, 0u);
    if (uni((int)X.overflow)) break;
  }
  int rest = 0;
  if (l == 0) { rest = (int)axchg32(X.avail + me, 0u); if (rest < 0) rest = 0; if (rest > 0) (void)aadd32(X.qwords + kQueueOpen, (uint32_t)(-rest)); }
  rest = uni(rest);
  const uint32_t expected = nTasks - (uint32_t)rest - nMine;   // (the anchors other waves have claimed)
  PROF_COUNT(PF_TSTOLEN, expected);
  {   // wait for the anchors other waves have claimed.  (Not by running other searches' anchors meanwhile: the wave of a heavy
      // read that waits a millisecond for its last anchor and takes a 40 ms one from elsewhere has moved the launch's end.)
    PROF_DECL; PROF_BEGIN();
    if (l == 0) { while (aload32(done) < expected) __builtin_amdgcn_s_sleep(32); }
    WSYNC();
    PROF_END(PF_TWAIT);
  }
  agent_fence();
  if (uni((int)X.overflow)) return 0;
  uint32_t ovf = 0, redo = 0, cx = 0;
  for (int s = s0; s < limit; ++s) {
    edge_box_load(box, s);
    const EdgeBoxRes* r = (const EdgeBoxRes*)g_dp;
    ovf |= r->overflow; redo |= r->redo; cx |= r->complex;
  }
  LSYNC();
  if (uni((int)ovf)) { X.overflow |= (uint32_t)uni((int)ovf); return 0; }
  if (uni((int)redo)) {   // (a distance that is not a number, a candidate longer than a box sequence: in order, here)
    edge_box_fold(box, first, s0);
    return 0;
  }
  if (uni((int)cx)) X.complexRegion = true;
  edge_box_fold(box, first, limit);
  return 1;
}

// a claimed anchor of another wave's edge search, run here: `old` is what the claim's decrement returned
TALC_DNC void edge_task_run(uint32_t victim_, int old_) {
  PROF_DECL2;
  PROF_BEGIN2();
  const int l = lane_id();
  const int old = uni(old_);
  agent_fence();
  uint8_t* box = uni_ptr(edge_box((uint32_t)uni((int)victim_)));
  uint32_t* box32 = (uint32_t*)box;
  const int limit = edge_ctx_load(box), s = limit - old;
  X.headCov = nullptr;
  X.cells = 0; X.steps = 0; X.overflow = 0; X.complexRegion = false;
  X.nanSeen = 0u; X.nEdges = 0;
  AnchorRec* mine = uni_ptr(X.dirRight ? X.ancL : X.ancR);
  if (s >= 0 && s < limit && limit <= EDGE_BOX_ANCHORS) {
    if (l < 6) ((uint32_t*)(mine + s))[l] = aload32(box32 + 24 + 6 * s + l);
    WSYNC();
    X.best2[0].have = false; X.best2[1].have = false;
    edge_anchor_search_task(mine, s);
    if (X.taskTest) X.nanSeen = 1u;
    edge_box_store(box, s, X.complexRegion ? 1u : 0u);
  }
  agent_fence();
  if (l == 0) (void)aadd32(box32 + offsetof(EdgeBoxHdr, done) / 4, 1u);
  PROF_END2(PF_TRUN);
}

// a wave without a read looks for a published anchor; true: it has run one
TALC_DNC bool edge_task_steal(int tries_) {
  const int l = lane_id();
  const int tries = uni(tries_);
  const uint32_t nS = (uint32_t)uni((int)X.nSlots), me = (uint32_t)uni((int)X.mySlot);
  {
    int open = 0;
    if (l == 0) open = (int)aload32(X.qwords + kQueueOpen);
    if (uni(open) <= 0) return false;
  }
  // a window of 64 counters per try, a different one every time: with thousands of waves polling, every box is looked
  // at within microseconds — and a wave between two reads spends one load on it (scanning all of avail[] from every idle
  // wave was 7 TB/s of L2 traffic at the end of a launch)
  const uint32_t nWin = (nS + 63u) >> 6;
  for (int k = 0; k < tries; ++k) {
    X.stealSeq += 1u;
    const uint32_t win = (me * 2654435761u + X.stealSeq * 40503u) % nWin;
    const uint32_t i = win * 64u + (uint32_t)l;
    const int v = (i < nS && i != me) ? (int)aload32(X.avail + i) : 0;
    unsigned long long m = ballot64(v > 0);
    while (m != 0ull) {
      const int src = (int)__builtin_ctzll(m);
      m &= m - 1ull;
      const uint32_t victim = (uint32_t)__shfl((int)i, src);
      int old = 0;
      if (l == 0) old = (int)aadd32(X.avail + victim, 0xFFFFFFFFu);
      old = uni(old);
      if (old > 0) {
        if (l == 0) (void)aadd32(X.qwords + kQueueOpen, 0xFFFFFFFFu);
        edge_task_run(victim, old);
        return true;
      }
    }
  }
  return false;
}

// between two reads of a wave: the read it had is finished (whichever way it left the loop's body), and published anchors
// — long borders of reads already under way — come before a new read
TALC_DNC void edge_between_reads() {
  if (!X.holding) { X.holding = 1u; return; }
  if (lane_id() == 0) (void)aadd32(X.qwords + kQueueFinished, 1u);
  while (edge_task_steal(1)) { X.moreCells += X.cells; X.moreSteps += X.steps; }
}
// the end of a wave that has run out of reads; returns the 100 MHz ticks it slept (the profile build's utilisation figure)
TALC_DNC unsigned long long edge_linger(uint32_t n_work_) {
  const uint32_t n_work = (uint32_t)uni((int)n_work_);
  const int l = lane_id();
  unsigned long long idle = 0;
  while (true) {
    WSYNC();
    if (edge_task_steal(8)) { X.moreCells += X.cells; X.moreSteps += X.steps; continue; }
    uint32_t fin = 0;
    if (l == 0) fin = aload32(X.qwords + kQueueFinished);
    if ((uint32_t)uni((int)fin) >= n_work) break;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < 4; ++i) __builtin_amdgcn_s_sleep(127);
    idle += __builtin_amdgcn_s_memrealtime() - t0;
  }
  return idle;
}

TALC_DN bool search_edge(uint32_t& weakOutOff, uint32_t& weakOutLen, uint32_t& weakUsed) {
  const DevParams& P = X.P;
  const AnchorRec* anchors = uni_ptr(X.dirRight ? X.ancL : X.ancR);
  const int nAnch = uni(X.dirRight ? X.nAncL : X.nAncR);
  const int limit = min(nAnch, uni((int)P.MAX_START_ANCHORS));
  X.best2[0].have = false; X.best2[1].have = false; X.nEdges = 0; X.nanSeen = 0u;
  bool mayPublish = (X.boxes != nullptr) && limit <= EDGE_BOX_ANCHORS && !X.tracing && X.weakLen >= X.taskMinWeak;
  for (int s = 0; s < limit; ++s) {
    if (uni((int)mayPublish) && limit - s >= 2 && ((X.weakLen >= X.taskHeavy && X.qi < X.heavyUpTo) || queue_is_dry())) {
      const int folded = uni(edge_anchors_as_tasks(anchors, s, limit));
      if (X.overflow) return false;
      if (folded) break;
      mayPublish = false;
    }
    edge_anchor_search(anchors, s);
    if (X.overflow) return false;
  }
  bool found = false;
  if (X.best2[1].have || X.best2[0].have) {
    // sortOutBestBorder (Explorer.cpp:310-329): any long path beats every short one
    const int wi = X.best2[0].have ? 0 : 1;
    const EdgeCand w = X.best2[wi];
    const uint8_t* wseq = (wi == 0) ? X.edgeLong : X.edgeShort;
    const double diff = (double)X.weakLen - (double)w.len;
    double minScore;
    if ((X.weakLen >= 300) || X.complexRegion) minScore = fmax(0.75, P.MIN_BORDER);
    else minScore = P.MIN_BORDER;
    if (((diff < X.weakLen * 0.05) || ((X.weakLen < 6) & (w.len < 6))) & (w.idscore >= minScore)) {
      found = true;
      if (weakUsed + w.len > X.C.weakPool) { X.overflow |= OVF_WEAKPOOL; return false; }
      if (w.len) copy_bytes(X.weak + weakUsed, wseq, w.len, !X.dirRight);
      weakOutOff = weakUsed; weakOutLen = w.len;
      weakUsed += w.len;
      if (X.location == LOC_TAIL) X.Le = w.lanc; else X.Rs = w.ranc;
      WSYNC();
    }
  }
  return found;
}

TALC_DNC void trace_search_cold() {
  if (!X.tracing) return;
  const int locRef = X.location == LOC_HEAD ? 0 : X.location == LOC_INNER ? 1 : 2;   // Location enum of the reference
  const int nL = (X.location == LOC_HEAD) ? 0 : X.nAncL, nR = (X.location == LOC_TAIL) ? 0 : X.nAncR;
  trace_rec(TR_SEARCH, locRef, X.dirRight, nL, nR, 0.0, nullptr, 0, false);
  const int K = (int)X.P.K;
  for (int i = 0; i < nL; ++i) trace_rec(TR_ANCHOR, 0, (int)X.ancL[i].pos, (int)X.ancL[i].count, 0, 0.0, X.read + X.ancL[i].pos, K, false);
  for (int i = 0; i < nR; ++i) trace_rec(TR_ANCHOR, 1, (int)X.ancR[i].pos, (int)X.ancR[i].count, 0, 0.0, X.read + X.ancR[i].pos, K, false);
}

#ifndef TALC_SEARCH_WAVES_PER_SIMD
#define TALC_SEARCH_WAVES_PER_SIMD 5   /* 20 waves per CU: <= 96 VGPRs and <= 8 KB of LDS per wave */
#endif
__global__ void __launch_bounds__(64, TALC_SEARCH_WAVES_PER_SIMD)
k_search(DevParams P, TableView T, SearchCaps C, const uint8_t* __restrict__ codes, const uint64_t* __restrict__ offsets,
         const uint64_t* __restrict__ koff, const uint2* __restrict__ covAll, const CovWord* __restrict__ covWords,
         ReadState* __restrict__ state,
         const uint32_t* __restrict__ regions, const uint64_t* __restrict__ regoff, const uint32_t* __restrict__ headCovAll,
         uint8_t* __restrict__ outAll,
         const uint64_t* __restrict__ outoff, const uint32_t* __restrict__ order, uint32_t n_work,
         uint32_t* __restrict__ queue, uint8_t* __restrict__ scratchAll, uint64_t* __restrict__ counters, TraceBuf trace,
         uint32_t traceRead, uint32_t launchStamp, uint32_t flags, EdgeTaskArgs E) {
  __shared__ uint32_t s_next;
  const int l = lane_id();
  uint8_t* slot = scratchAll + (uint64_t)blockIdx.x * C.slotBytes;
  X.P = P; X.T = T; X.C = (const SearchLimits&)C;
  X.G[0] = make_set(slot + C.o_setA);
  X.G[1] = make_set(slot + C.o_setB);
  X.ia = 0;
  X.seqPool = slot + C.o_seqPool;
  X.refBuf = slot + C.o_ref; X.ref = X.refBuf; X.ancL = (AnchorRec*)(slot + C.o_ancL); X.ancR = (AnchorRec*)(slot + C.o_ancR);
  X.ancPos = (uint32_t*)(slot + C.o_ancPos);
  X.fullMeta = (FullMeta*)(slot + C.o_fullMeta); X.fullPool = slot + C.o_fullPoolB;
  X.edgeLong = slot + C.o_edgeLong; X.edgeShort = slot + C.o_edgeShort; X.edgeTmp = slot + C.o_edgeTmp;
  X.dpG = (int*)(slot + C.o_dp);
  X.wideBloom = (unsigned long long*)(slot + C.o_wideBloom); X.wideMask = 0;
  X.rowPool = (flags & 1u) ? nullptr : (int*)(slot + C.o_rowPool); X.rowStride = 0; X.rowAvail = 0;   // (flags bit 0: TALC_NO_ROWS)
  X.launchStamp = launchStamp;
  X.noForkStep = (flags >> 1) & 1u;
  X.childrenByLane = ((flags >> 2) & 1u) ^ 1u;
  // (over a graph that does not branch the queue's order alone keeps the waves busy 95 % of the launch, and the tasks'
  //  bookkeeping and the waves that stay cost config 2 0.7 %: the batch's fork share decides, k_order_scale)
  if (E.boxes != nullptr && E.autoSwitch != nullptr && *E.autoSwitch == 0u) E.boxes = nullptr;
  X.boxes = E.boxes; X.avail = E.avail; X.qwords = queue; X.boxSeqCap = E.seqCap; X.boxBytes = (uint32_t)edge_box_bytes(E.seqCap);
  X.nSlots = gridDim.x; X.mySlot = blockIdx.x; X.nWork = n_work; X.taskMinWeak = E.minWeak; X.nanSeen = 0u; X.taskTest = E.test;
  X.taskHeavy = E.heavy; X.heavyUpTo = (E.heavyRounds >= 0xFFFFu) ? 0xFFFFFFFFu : E.heavyRounds * gridDim.x;
  X.qi = 0; X.holding = 0u; X.stealSeq = 0u; X.moreCells = 0; X.moreSteps = 0;
  const uint32_t lingerMod = max(E.lingerMod, 1u);   // (5120 waves polling two words kept a memory channel busy: config 2 + 0.7 ms with nothing ever published)
  X.searchNo = 16u;   // (stamps below 16 << ROW_COV_BITS could be matrix values)
  {
    uint8_t* g = slot + C.o_gard;
    X.gScores = (double*)g; g += 8ull * (TCAP + 64);
    X.gDists = (double*)g; g += 8ull * (TCAP + 64);
    X.gVal = (ValIdx*)g; g += 16ull * (TCAP + 64);
    X.gRank = (Rank4*)g; g += 32ull * (TCAP + 64);
    X.gKept = (uint32_t*)g;
  }
  X.regS = (uint32_t*)(slot + C.o_regS); X.regE = (uint32_t*)(slot + C.o_regE); X.regH = (uint32_t*)(slot + C.o_regH);
  X.wOff = (uint32_t*)(slot + C.o_wOff); X.wLen = (uint32_t*)(slot + C.o_wLen);
  X.weak = slot + C.o_weak;
  if (l == 0) *(TraceBuf*)(slot + C.o_trace) = trace;
  X.tracep = (const TraceBuf*)(slot + C.o_trace);
  unsigned long long totCells = 0, totSteps = 0;
  PROF_DECL2;
#ifdef TALC_PROF
  if (l == 0) { for (int i = 0; i < PF_N; ++i) g_prof[i] = 0; for (int i = 0; i < 4; ++i) g_wprof[i] = 0; }
  const unsigned long long _pf_k0 = __builtin_amdgcn_s_memtime();
  const unsigned long long _pf_r0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz, the same counter on every CU
  unsigned long long _pf_rd0 = 0, _pf_lastStart = 0, _pf_loopEnd = 0, _pf_idle = 0;   // (_pf_idle: asleep without a read, waiting for anchors to be published)
  uint32_t _pf_prevQi = 0, _pf_prevR = 0;
#endif

  while (true) {
    WSYNC();
#ifdef TALC_PROF
    if (l == 0 && _pf_rd0) {   // duration of the previous read of this wave (100 MHz ticks)
      const unsigned long long dt = __builtin_amdgcn_s_memrealtime() - _pf_rd0;
      const int bk = dt < 25000ull ? 0 : dt < 100000ull ? 1 : dt < 400000ull ? 2 : dt < 1600000ull ? 3 : dt < 6400000ull ? 4 : 5;
      g_prof[PF_RD0 + bk] += 1;
      if (dt > g_prof[PF_RDMAX]) g_prof[PF_RDMAX] = (uint32_t)dt;
      state[_pf_prevR].pfTicks = (uint32_t)dt;
      _pf_lastStart = _pf_rd0; _pf_rd0 = 0;
    }
#endif
    if (X.boxes != nullptr) edge_between_reads();
    if (l == 0) s_next = atomicAdd(queue, 1u);
    WSYNC();
    const uint32_t qi = s_next;
    if (qi >= n_work) break;
    X.qi = qi;
#ifdef TALC_PROF
    _pf_rd0 = __builtin_amdgcn_s_memrealtime();
#endif
    PROF_BEGIN2();
    const uint32_t r = order[qi];
#ifdef TALC_PROF
    if (l == 0) state[r].pfStart = (uint32_t)_pf_rd0;
#endif
#ifdef TALC_PROF
    _pf_prevQi = qi; _pf_prevR = r;
#endif
    const uint64_t rb = offsets[r];
    const uint32_t L = (uint32_t)(offsets[r + 1] - rb);
    uint8_t* out = outAll + outoff[r];
    const uint32_t outCap = (uint32_t)(outoff[r + 1] - outoff[r]);
    ReadState st = state[r];
    X.read = codes + rb; X.L = L; X.n = L >= P.K ? L - P.K + 1 : 0; X.cov = covAll + koff[r]; X.covw = covWords + cov_word_base(koff[r], r); X.headCov = headCovAll + (uint64_t)r * kHeadCov; X.lambda = st.lambda;
    X.cells = 0; X.steps = 0; X.overflow = 0; X.complexRegion = false; X.ffPopped = false;
    X.tracing = (trace.recs != nullptr) && (r == traceRead);
    X.traceSteps = X.tracing && (trace.steps != 0);

    if (st.status != TALC_READ_CORRECTED || st.overflow) {
      // not corrected: pass the (encoded) read through (main.cpp:310 writes mySeqs[r] unchanged)
      copy_bytes(out, X.read, L, false);
      if (l == 0) { state[r].outLen = L; }
      continue;
    }
    const uint32_t K = P.K;
    const uint32_t R = st.nRegions;
    if (R > C.regCap) { X.overflow |= OVF_REGIONS; }
    else {
      const uint32_t* gS = regions + 3 * regoff[r];
      const uint32_t gCap = (uint32_t)(regoff[r + 1] - regoff[r]);
      const uint32_t* gE = gS + gCap;
      const uint32_t* gH = gE + gCap;
      for (uint32_t i = l; i < R; i += 64) { X.regS[i] = gS[i]; X.regE[i] = gE[i]; X.regH[i] = gH[i]; X.wLen[i] = 0xFFFFFFFFu; X.wOff[i] = 0; }
    }
    WSYNC();
    uint32_t weakUsed = 0;
#ifdef TALC_PROF
    uint32_t _pf_edge = 0; unsigned long long _pf_e0 = 0; (void)_pf_e0;
#endif
    PROF_END2(PF_PROLOG);
    // head / tail presence as set by setInitialStructure (Read.cpp:223-237)
    bool headPresent = false, tailPresent = false;
    uint32_t headLen = 0, tailLen = 0;
    bool headCorr = false, tailCorr = false;
    uint32_t headOff = 0, headCLen = 0, tailOff = 0, tailCLen = 0;
    if (!X.overflow) {
      headPresent = X.regS[0] > 0; headLen = X.regS[0];
      const uint32_t eLast = X.regE[R - 1];
      tailPresent = (eLast + 1 < X.n); tailLen = tailPresent ? (L - (eLast + K)) : 0;
      // ---- inner regions (Read.cpp:346-360)
      for (uint32_t reg = 0; reg + 1 < R && !X.overflow; ++reg) {
        uint32_t wo = 0, wl = 0;
        bool success = false;
        for (int attempt = 0; attempt < 2 && !success && !X.overflow; ++attempt) {
          // initializeINNER (Explorer.cpp:228-243)
          X.location = LOC_INNER; X.dirRight = (attempt == 0) ? 1 : 0;
          X.Ls = X.regS[reg]; X.Le = X.regE[reg]; X.Rs = X.regS[reg + 1]; X.Re = X.regE[reg + 1];
          X.LH = X.regH[reg]; X.RH = X.regH[reg + 1];
          X.weakLen = (X.Rs > X.Le + K) ? (X.Rs - (X.Le + K)) : 0;
          // (the LEFT attempt after a failed RIGHT one starts from the same two regions — a failed search changes
          //  neither — so anchorLEFTHandSide / anchorRIGHTHandSide, Explorer.cpp:239-240, would return the lists again)
          if (attempt == 0) { build_anchors(0); build_anchors(1); }
          if (uni((int)X.tracing)) trace_search_cold();
          PROF_BEGIN2(); success = search_bridge(wo, wl, weakUsed); PROF_END2(PF_SRCHB);
          if (uni((int)X.tracing)) {
            if (success) trace_rec(TR_RESULT, 1, 1, (int)X.Le, (int)X.Rs, 0.0, X.weak + wo, wl, false);
            else trace_rec(TR_RESULT, 1, 0, (int)X.regE[reg], (int)X.regS[reg + 1], 0.0, X.read + X.regE[reg] + K, X.weakLen, false);
          }
        }
        // updateINNER (Read.cpp:294-303)
        WSYNC();
        // (a region's start only ever moves inward: its hit index moves with it)
        if (success && l == 0) { X.regE[reg] = X.Le; X.regH[reg + 1] += X.Rs - X.regS[reg + 1]; X.regS[reg + 1] = X.Rs; X.wOff[reg] = wo; X.wLen[reg] = wl; }
        WSYNC();
      }
      // ---- head (Read.cpp:361-367)
      if (!X.overflow && headPresent && headLen <= P.MAX_BORDER_LEN) {
        X.location = LOC_HEAD; X.dirRight = 0;
        X.Rs = X.regS[0]; X.Re = X.regE[0]; X.Ls = 0; X.Le = 0; X.RH = X.regH[0]; X.LH = 0;
        X.weakLen = X.Rs;
        X.nAncL = 0;
        build_anchors(1);
        if (uni((int)X.tracing)) trace_search_cold();
        PROF_BEGIN2(); PF_EDGE_T0(); headCorr = search_edge(headOff, headCLen, weakUsed); PF_EDGE_T1(); PROF_END2(PF_SRCHE);
        if (uni((int)X.tracing)) {
          if (headCorr) trace_rec(TR_RESULT, 0, 1, 0, (int)X.Rs, 0.0, X.weak + headOff, headCLen, false);
          else trace_rec(TR_RESULT, 0, 0, 0, (int)X.regS[0], 0.0, X.read, X.weakLen, false);
        }
        WSYNC();
        if (headCorr && l == 0) { X.regH[0] += X.Rs - X.regS[0]; X.regS[0] = X.Rs; }   // updateHEAD (Read.cpp:305-311)
        WSYNC();
      }
      // ---- tail (Read.cpp:368-374)
      if (!X.overflow && tailPresent && tailLen <= P.MAX_BORDER_LEN) {
        X.location = LOC_TAIL; X.dirRight = 1;
        X.Ls = X.regS[R - 1]; X.Le = X.regE[R - 1]; X.Rs = 0; X.Re = 0; X.LH = X.regH[R - 1]; X.RH = 0;
        X.weakLen = L - (X.Le + K);
        X.nAncR = 0;
        build_anchors(0);
        if (uni((int)X.tracing)) trace_search_cold();
        PROF_BEGIN2(); PF_EDGE_T0(); tailCorr = search_edge(tailOff, tailCLen, weakUsed); PF_EDGE_T1(); PROF_END2(PF_SRCHE);
        if (uni((int)X.tracing)) {
          if (tailCorr) trace_rec(TR_RESULT, 2, 1, (int)X.Le, 0, 0.0, X.weak + tailOff, tailCLen, false);
          else trace_rec(TR_RESULT, 2, 0, (int)X.regE[R - 1], 0, 0.0, X.read + X.regE[R - 1] + K, X.weakLen, false);
        }
        WSYNC();
        if (tailCorr && l == 0) X.regE[R - 1] = X.Le;   // updateTAIL (Read.cpp:313-318)
        WSYNC();
      }
    }
    totCells += X.cells; totSteps += X.steps;
#ifdef TALC_PROF
    if (l == 0) { state[r].pfSteps = (uint32_t)X.steps; state[r].pfEdgeTicks = _pf_edge; state[r].pfAnchors = g_prof[PF_RANCH]; state[r].pfAnchorMax = g_prof[PF_RANCHMAX]; g_prof[PF_RANCH] = 0; g_prof[PF_RANCHMAX] = 0;
                  state[r].pfBridges = g_prof[PF_RBR]; state[r].pfBridgeMax = g_prof[PF_RBRMAX]; g_prof[PF_RBR] = 0; g_prof[PF_RBRMAX] = 0; }
#endif
    if (X.overflow) {
      copy_bytes(out, X.read, L, false);
      if (l == 0) { state[r].outLen = L; state[r].overflow = X.overflow; }
      continue;
    }
    // ---- updateCorrSeq (Read.cpp:320-326): head + (solid, weak)* + solid + tail
    // total length first
    unsigned long long total = 0;
    total += headPresent ? (headCorr ? headCLen : headLen) : 0;
    {
      unsigned long long part = 0;
      for (uint32_t i = l; i < R; i += 64) {
        part += (unsigned long long)X.regE[i] + K - X.regS[i];
        if (i + 1 < R) {
          if (X.wLen[i] != 0xFFFFFFFFu) part += X.wLen[i];
          else part += (X.regS[i + 1] > X.regE[i] + K) ? (X.regS[i + 1] - (X.regE[i] + K)) : 0;
        }
      }
      total += wave_sum_u64(part);
    }
    total += tailPresent ? (tailCorr ? tailCLen : tailLen) : 0;
    if (total > outCap) {
      copy_bytes(out, X.read, L, false);
      if (l == 0) { state[r].outLen = L; state[r].overflow = OVF_OUT; }
      continue;
    }
    uint32_t pos = 0;
    PROF_BEGIN2();
    if (headPresent) {
      if (headCorr) { copy_bytes(out + pos, X.weak + headOff, headCLen, false); pos += headCLen; }
      else { copy_bytes(out + pos, X.read, headLen, false); pos += headLen; }
    }
    for (uint32_t i = 0; i < R; ++i) {
      const uint32_t s = X.regS[i], e = X.regE[i];
      const uint32_t sl = e + K - s;
      copy_bytes(out + pos, X.read + s, sl, false); pos += sl;
      if (i + 1 < R) {
        if (X.wLen[i] != 0xFFFFFFFFu) { copy_bytes(out + pos, X.weak + X.wOff[i], X.wLen[i], false); pos += X.wLen[i]; }
        else if (X.regS[i + 1] > e + K) { const uint32_t wl = X.regS[i + 1] - (e + K); copy_bytes(out + pos, X.read + e + K, wl, false); pos += wl; }
      }
    }
    if (tailPresent) {
      if (tailCorr) { copy_bytes(out + pos, X.weak + tailOff, tailCLen, false); pos += tailCLen; }
      else { copy_bytes(out + pos, X.read + (L - tailLen), tailLen, false); pos += tailLen; }
    }
    {   // Read.cpp:423 on the regions as correct2 leaves them
      unsigned long long part = 0;
      for (uint32_t i = l; i < R; i += 64) part += (unsigned long long)X.regE[i] - X.regS[i] + 1;
      const uint32_t span = (uint32_t)wave_sum_u64(part);
      if (l == 0) { state[r].outLen = pos; state[r].inSpan = span; }
    }
    PROF_END2(PF_ASSEMBLE);
  }
  // no reads left: run published anchors of the edge searches still going on, until every read is finished
#ifdef TALC_PROF
  _pf_loopEnd = __builtin_amdgcn_s_memrealtime();
  if (l == 0) atomicMin((unsigned long long*)&counters[124], _pf_loopEnd);   // the first wave to find the queue dry
#endif
  if (X.boxes != nullptr && blockIdx.x % lingerMod == 0u) {
#ifdef TALC_PROF
    _pf_idle =
#endif
    edge_linger(n_work);
  }
  totCells += X.moreCells; totSteps += X.moreSteps;
  if (l == 0) {
    if (totSteps) atomicAdd((unsigned long long*)&counters[0], totSteps);
    if (totCells) atomicAdd((unsigned long long*)&counters[1], totCells);
#ifdef TALC_PROF
    g_prof[PF_TOTAL] = (uint32_t)(__builtin_amdgcn_s_memtime() - _pf_k0);
    g_prof[PF_XSTAGE] = g_wprof[0]; g_prof[PF_XLEV] = g_wprof[1]; g_prof[PF_XSEL] = g_wprof[2]; g_prof[PF_XNLEV] = g_wprof[3];
    {   // wave utilisation of the launch: sum of the waves' lifetimes against (last end - first start) x waves
      const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
      atomicAdd((unsigned long long*)&counters[125], r1 - _pf_r0 - _pf_idle);   // (counters[2 .. 2 + PF_N) are the categories)
      atomicMin((unsigned long long*)&counters[126], _pf_r0);
      atomicMax((unsigned long long*)&counters[127], r1);
      // the wave's last read: (queue position, read), (its start, the wave's end) — TALC_PROF_SLOW prints the waves that end last
      if (_pf_rd0) state[_pf_prevR].pfTicks = (uint32_t)(r1 - _pf_rd0);
      if (blockIdx.x < 8192u) {
        counters[128 + 2 * blockIdx.x] = ((unsigned long long)_pf_prevQi << 32) | _pf_prevR;
        counters[129 + 2 * blockIdx.x] = ((_pf_lastStart & 0xFFFFFFFFull) << 32) | (_pf_loopEnd & 0xFFFFFFFFull);   // (the end of its last read, not of its stay)
      }
      static_assert(2 + PF_N <= 124, "the profile categories run into the utilisation counters");
    }
    for (int i = 0; i < PF_N; ++i) {
      if (i == PF_RDMAX) atomicMax((unsigned long long*)&counters[2 + i], (unsigned long long)g_prof[i]);
      else atomicAdd((unsigned long long*)&counters[2 + i], (unsigned long long)g_prof[i]);
    }
#endif
  }
}

// ==================================================================== work-queue order
// Reads by descending cost estimate, without a sort: a 1024-bucket counting sort on a 10-bit logarithmic key (the order
// inside a bucket is whatever the atomics make it: records do not depend on the order reads are taken in).
// The key: the estimate, with its inner-gap part weighed by `gapScale` (in 1/256).  Over a branching graph (paralog
// families: 3.5 % of the solid k-mers of a batch fork, against 1 % over unique sequence with sequencing-error k-mers) a base
// of an inner gap costs ten to twenty times what it costs elsewhere — the walk carries several Trails — while the edges
// cost the same, and the edges are the part of a read other waves can take over (edge tasks): the reads with long inner
// gaps have to start first there.  k_order_scale: 1 up to a fork share of 1.2 %, + 1 per further 1 %, at most 3.
TALC_D uint32_t order_bucket(const ReadState& st, uint32_t gapScale) {
  if (st.status != TALC_READ_CORRECTED || st.overflow) return 1023u;   // passed through: last
  const unsigned long long key = (unsigned long long)st.costEst + (((unsigned long long)st.costGap * (gapScale - 256u)) >> 8);
  const uint32_t c = (uint32_t)min(key, 0xFFFFFFFFull) | 1u;
  const int e = 31 - __builtin_clz(c);                               // 0..31
  const uint32_t m = (e >= 5) ? ((c >> (e - 5)) & 31u) : ((c << (5 - e)) & 31u);
  return 1022u - min(1022u, (uint32_t)e * 32u + m);                  // heavy first
}
__global__ void __launch_bounds__(64) k_order_scale(uint32_t* __restrict__ batchStats, uint32_t fixedScale) {   // batchStats[128] := the gap scale, [129] := the graph branches
  const int l = lane_id();
  const unsigned long long fork = wave_sum_u64(batchStats[2 * l]), solid = wave_sum_u64(batchStats[2 * l + 1]);
  uint32_t scale = 256u;
  if (solid > 0) {
    const unsigned long long perMille10 = fork * 10000ull / solid;   // fork share in 1/10000
    if (perMille10 > 120ull) scale = 256u + (uint32_t)min((perMille10 - 120ull) * 256ull / 100ull, 512ull);
  }
  if (l == 0) batchStats[129] = scale > 256u ? 1u : 0u;   // (the edge tasks' switch)
  if (fixedScale) scale = fixedScale;
  if (l == 0) batchStats[128] = scale;
}
__global__ void k_order_hist(const ReadState* __restrict__ state, uint32_t n, uint32_t* __restrict__ hist, const uint32_t* __restrict__ batchStats) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n) atomicAdd(&hist[order_bucket(state[r], batchStats[128])], 1u);
}
__global__ void __launch_bounds__(1024) k_order_scan(uint32_t* __restrict__ hist) {   // hist[b] := first position of bucket b
  __shared__ uint32_t s[1024];
  const uint32_t t = threadIdx.x;
  const uint32_t v = hist[t];
  s[t] = v;
  __syncthreads();
  for (uint32_t off = 1; off < 1024; off <<= 1) {
    const uint32_t add = (t >= off) ? s[t - off] : 0u;
    __syncthreads();
    s[t] += add;
    __syncthreads();
  }
  hist[t] = s[t] - v;
}
__global__ void k_order_scatter(const ReadState* __restrict__ state, uint32_t n, uint32_t* __restrict__ cursor, uint32_t* __restrict__ order,
                                const uint32_t* __restrict__ batchStats) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n) order[atomicAdd(&cursor[order_bucket(state[r], batchStats[128])], 1u)] = r;
}

// ==================================================================== the count model's thresholds (DevParams.thr)
// One thread per count c < n: both predicates of isExpectedbyMyModel are monotone in nextc, so each is one number — found
// by bisection on the predicate ITSELF (the formula as the device evaluates it), not by solving it: the table cannot
// disagree with the formula it replaces.
__global__ void k_build_thresholds(double ALPHA, uint32_t n, uint32_t* __restrict__ thr) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n) return;
  uint32_t lo = 0, hi = 0x7FFFFFFFu;   // smallest nextc accepted by (nextc, c, false): "nextc >= bound"
  if (!is_expected_by_model(ALPHA, hi, c, false)) lo = hi = 0xFFFFFFFFu;
  while (lo < hi) { const uint32_t mid = lo + (hi - lo) / 2; if (is_expected_by_model(ALPHA, mid, c, false)) hi = mid; else lo = mid + 1; }
  thr[2u * c] = lo;
  lo = 0; hi = 0x7FFFFFFFu;            // number of nextc accepted by (nextc, c, true): "nextc <= bound"
  if (is_expected_by_model(ALPHA, hi, c, true)) lo = hi = 0xFFFFFFFFu;
  while (lo < hi) { const uint32_t mid = lo + (hi - lo) / 2; if (!is_expected_by_model(ALPHA, mid, c, true)) hi = mid; else lo = mid + 1; }
  thr[2u * c + 1u] = lo;               // the first nextc that is NOT accepted = how many are
}

// ==================================================================== k_pack
// dense output: codes -> ASCII, with the reverse complement of main.cpp:286 for corrected reads
// under -rev.  One block per (read, 4096-base chunk) of the OUTPUT.
__global__ void k_pack(const uint8_t* __restrict__ outAll, const uint64_t* __restrict__ outoff, const ReadState* __restrict__ state,
                       const uint64_t* __restrict__ dense_off, uint8_t* __restrict__ dense, uint32_t n_reads, int reverse) {
  const uint32_t r = blockIdx.x;
  if (r >= n_reads) return;
  const uint8_t* src = outAll + outoff[r];
  const uint32_t len = state[r].outLen;
  uint8_t* dst = dense + dense_off[r];
  const bool rc = reverse && state[r].status == TALC_READ_CORRECTED && state[r].overflow == 0;
  for (uint32_t i = threadIdx.x; i < len; i += blockDim.x) {
    if (rc) dst[i] = (uint8_t)code_to_ascii(complement_code(src[len - 1 - i]));
    else dst[i] = (uint8_t)code_to_ascii(src[i]);
  }
}

// ==================================================================== k_test_dp (test hook)
// mode 0: nw_score(a, b, match, mismatch, gap, freeBegin)          -> out[0]
// mode 1: seed_and_extension(ref=a, cand=b, xdrop, dirRight, true) -> out[0..4] = lenRefExt, lenHistExt,
//         posOnRef, score, stop
// mode 2: wave_find_window(a, pattern=b, wantLast=p3)              -> out[0]
// mode 5: seed_and_extension_multi(a, b, xHi = p0) against seed_and_extension(x) for every x  -> out[0..2]
// mode 4: edit_and_lcs(a, b, needEdit = !p0, acceptLcs = p1 if p0 == 2) -> out[0] = edit score (0 if not asked), out[1] = LCS
__global__ void __launch_bounds__(64, TALC_SEARCH_WAVES_PER_SIMD)   // (same register budget as k_search: they share the step functions)
k_test_dp(int mode, const uint8_t* a, int la, const uint8_t* b, int lb, int p0, int p1, int p2, int p3, int K,
          int* dpG, uint32_t dpCap, int* out, double alpha, double err, int minc) {
  if (lane_id() == 0) memset(&g_X, 0, sizeof g_X);
  WSYNC();
  X.P.K = (uint32_t)K; X.P.ALPHA = alpha; X.P.ERR = err; X.P.MIN_COUNT = (uint32_t)minc;
  X.C.dpCap = dpCap; X.dpG = dpG;
  X.dirRight = p1;
  if (mode == 0) {
    const int r = nw_score(a, la, b, lb, p0, p1, p2, p3 != 0);
    if (lane_id() == 0) { out[0] = r; out[5] = (int)X.overflow; }
  } else if (mode == 1) {
    const SeedExt e = seed_and_extension(a, la, b, lb, p0, true);
    if (lane_id() == 0) { out[0] = e.lenRefExt; out[1] = e.lenHistExt; out[2] = e.posOnRef; out[3] = e.score; out[4] = e.stop ? 1 : 0; out[5] = (int)X.overflow; }
  } else if (mode == 2) {
    const int r = wave_find_window(a, la, b, lb, p3 != 0);
    if (lane_id() == 0) out[0] = r;
  } else if (mode == 5) {
    // mode 5: every x in [0, p0] from seed_and_extension_multi against seed_and_extension(x):
    // out[0] = 1 if the multi form was available, out[1] = number of x that differ, out[2] = the first such x
    int* rc_ = X.dpG; int* rr_ = X.dpG + X.C.dpCap; int* rs_ = X.dpG + 2ull * X.C.dpCap;
    const bool ok = seed_and_extension_multi(a, la, b, lb, p0, rc_, rr_, rs_);
    int nbad = 0, first = -1;
    const int S5 = X.dirRight ? K - 1 : K;
    const bool sameAnchor = ballot64(lane_id() < S5 && lane_id() < la && lane_id() < lb && a[lane_id() < la ? lane_id() : 0] != b[lane_id() < lb ? lane_id() : 0]) == 0ull && la >= S5 && lb >= S5;
    if (ok) {
      for (int x = 0; x <= p0; ++x) {
        const SeedExt m = seedext_plain(la, lb, uni(rc_[x]), uni(rr_[x]), x);
        const int msc = uni(rs_[x]);
        const SeedExt e = seed_and_extension(a, la, b, lb, x, false);
        bool bad = (m.lenRefExt != e.lenRefExt || m.lenHistExt != e.lenHistExt || m.posOnRef != e.posOnRef || m.stop != e.stop || m.score != e.score);
        if (!e.stop && msc <= 0 && sameAnchor) {   // the score of the reported cell = what the scoring form returns
          const SeedExt es = seed_and_extension(a, la, b, lb, x, true);
          bad |= (es.score != msc);
        }
        if (bad) {
          if (first < 0) first = x;
          ++nbad;
        }
      }
    }
    if (lane_id() == 0) { out[0] = ok ? 1 : 0; out[1] = nbad; out[2] = first; out[5] = (int)X.overflow; }
  } else if (mode == 7) {
    // mode 7: scoreBridges' alignment continued row by row (wave_nw_rows) against the alignment from scratch (nw_score):
    // a = the reference, b = the candidate; for m = p0, p0 + p2, ... <= lb the kept row (behind the DP arrays) is
    // continued to m rows and its entry at tlen = min(la, m + p3) compared with nw_score(a[0, tlen), b[0, m), free begin).
    // out[0] = scorings, out[1] = those that differ, out[2] = the first such m
    int* row = X.dpG + 3ull * X.C.dpCap;
    int calls = 0, nbad = 0, first = -1, i0 = 0;
    const int B = (la + 63) >> 6;
    for (int m = p0; m <= lb && la >= 1 && la <= ROW_MAX_REF; m += max(p2, 1)) {
      const int tlen = min(la, m + p3);
      unsigned long long ncells = 0;
      int sc;
      WSYNC();
      if (B <= 2) sc = wave_nw_rows<2>(a, la, b, i0, m, 4, -3, -2, row, tlen, ncells);
      else if (B <= 4) sc = wave_nw_rows<4>(a, la, b, i0, m, 4, -3, -2, row, tlen, ncells);
      else if (B <= 6) sc = wave_nw_rows<6>(a, la, b, i0, m, 4, -3, -2, row, tlen, ncells);
      else if (B <= 8) sc = wave_nw_rows<8>(a, la, b, i0, m, 4, -3, -2, row, tlen, ncells);
      else if (B <= 12) sc = wave_nw_rows<12>(a, la, b, i0, m, 4, -3, -2, row, tlen, ncells);
      else if (B <= 16) sc = wave_nw_rows<16>(a, la, b, i0, m, 4, -3, -2, row, tlen, ncells);
      else if (B <= 24) sc = wave_nw_rows<24>(a, la, b, i0, m, 4, -3, -2, row, tlen, ncells);
      else if (B <= 32) sc = wave_nw_rows<32>(a, la, b, i0, m, 4, -3, -2, row, tlen, ncells);
      else if (B <= 48) sc = wave_nw_rows<48>(a, la, b, i0, m, 4, -3, -2, row, tlen, ncells);
      else if (B <= 64) sc = wave_nw_rows<64>(a, la, b, i0, m, 4, -3, -2, row, tlen, ncells);
      else if (B <= 96) sc = wave_nw_rows<96>(a, la, b, i0, m, 4, -3, -2, row, tlen, ncells);
      else sc = wave_nw_rows<128>(a, la, b, i0, m, 4, -3, -2, row, tlen, ncells);
      i0 = m;
      WSYNC();
      const int ex = nw_score(a, tlen, b, m, 4, -3, -2, true);
      if (uni(sc) != uni(ex)) { if (first < 0) first = m; ++nbad; }
      ++calls;
    }
    if (lane_id() == 0) { out[0] = calls; out[1] = nbad; out[2] = first; out[5] = (int)X.overflow; }
  } else if (mode == 8) {
    // mode 8: the kept wavefront (talc_wave.h: WfaKeep).  a = the reference, b = the candidate; the candidate's first
    // m = p0, p0 + p2, ... <= lb bases are scored as an edge search scores its Trail — x starts at p3 and follows the
    // scores (x = -score + 2, Explorer.cpp:713) — once through the leaf instances with a key (level kept / resumed from
    // call to call) and once through the general function (from level 0): out[0] = scorings, out[1] = those that
    // differ, out[2] = the first such m, out[3] = scorings that did resume
    int calls = 0, nbad = 0, first = -1, nres = 0;
    int x = p3;
    if (lane_id() == 0) g_keep.owner = 0u;
    WSYNC();
    for (int m = p0; m <= lb; m += max(p2, 1)) {
      const int lvBefore = ((uint32_t)uni((int)g_keep.owner) == 1u) ? uni(g_keep.level) : 0;
      SeedExt e1 = seed_and_extension_leaf(a, la, b, m, x, 1u);
      if (uni((int)e1.fallback) != 0) e1 = seed_and_extension(a, la, b, m, x, true);
      else if (lvBefore >= 1 && lvBefore < x) ++nres;
      const SeedExt e0 = seed_and_extension(a, la, b, m, x, true);
      const bool bad = (uni(e0.lenRefExt) != uni(e1.lenRefExt)) || (uni(e0.lenHistExt) != uni(e1.lenHistExt)) || (uni(e0.posOnRef) != uni(e1.posOnRef)) ||
                       (uni(e0.score) != uni(e1.score)) || (uni((int)e0.stop) != uni((int)e1.stop));
      if (bad) { if (first < 0) first = m; ++nbad; }
      ++calls;
      x = max(-uni(e0.score), 0) + 2;
    }
    if (lane_id() == 0) { out[0] = calls; out[1] = nbad; out[2] = first; out[3] = nres; out[5] = (int)X.overflow; }
  } else if (mode == 4) {
    // mode 4: edit_and_lcs(a, b) -> out[0] = global (0,-1,-1) score, out[1] = LCS length
    int es = 0, lcs = 0;
    edit_and_lcs(a, la, b, lb, es, lcs, p0 == 0, p0 == 2 ? p1 : 0);
    if (lane_id() == 0) { out[0] = es; out[1] = lcs; out[5] = (int)X.overflow; }
  } else {
    // mode 3: tag_next_nodes on the device.  a = 4 counts + 4 colours + count as 9 little-endian u32 (36 bytes),
    // p0 = complex; out[0..3] = tags, out[4..11] = the 4 distances as raw bits
    uint32_t w[9];
    for (int i = 0; i < 9; ++i) w[i] = (uint32_t)a[4 * i] | ((uint32_t)a[4 * i + 1] << 8) | ((uint32_t)a[4 * i + 2] << 16) | ((uint32_t)a[4 * i + 3] << 24);
    int tg[4]; double ds[4];
    tag_next_nodes(X.P.ALPHA, X.P.ERR, X.P.MIN_COUNT, w, w + 4, w[8], p0 != 0, tg, ds);
    if (lane_id() == 0) for (int i = 0; i < 4; ++i) { out[i] = tg[i]; long long bits = __double_as_longlong(ds[i]); out[4 + 2 * i] = (int)(bits & 0xffffffffll); out[5 + 2 * i] = (int)(bits >> 32); }
  }
}

#undef X
#endif  // __HIPCC__

}  // namespace talc
