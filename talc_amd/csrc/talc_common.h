// talc_common.h — types shared by the host side and the HIP kernels of libtalc_hip.so.
#pragma once
#include <stdint.h>

#include "talc_hip.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define TALC_HD __host__ __device__ __forceinline__
#define TALC_D __device__ __forceinline__
#ifdef TALC_NO_NOINLINE
#define TALC_DN __device__ __forceinline__
#define TALC_DNC __device__ __forceinline__
#else
#define TALC_DN __device__ __attribute__((noinline))
#define TALC_DNC __device__ __attribute__((noinline, cold))   /* rarely executed: optimised for size, laid out apart */
#endif
#else
#define TALC_HD inline
#define TALC_D inline
#endif

namespace talc {

#if defined(__HIPCC__)
// Explicit address spaces: pointers that travel through structs / LDS come back as generic pointers
// and would be accessed with flat_* instructions (which tie vmcnt and lgkmcnt together); these casts
// restore global_* / ds_* codegen.  Only ever applied to pointers known to be HBM (AS1) or LDS (AS3).
#define TALC_AS1 __attribute__((address_space(1)))
#define TALC_AS3 __attribute__((address_space(3)))
#define TALC_AS4 __attribute__((address_space(4)))   /* constant: uniform loads become scalar loads */
typedef const uint8_t TALC_AS1* gcu8;
typedef uint8_t TALC_AS1* gu8;
typedef uint32_t v2u32 __attribute__((ext_vector_type(2)));
typedef uint32_t v8u32 __attribute__((ext_vector_type(8)));
typedef uint32_t v16u32 __attribute__((ext_vector_type(16)));
typedef uint32_t v4u32 __attribute__((ext_vector_type(4)));   // plain vector: usable through AS-qualified pointers
#endif

// ------------------------------------------------------------------ successor-grouped k-mer table
// The reference's std::map<Dna5String, pair<uint,uint>> (Jellyfish.hpp:32-34) becomes two
// open-addressed tables of 32-byte buckets, one per walking direction:
//   RIGHT table: key = the (K-1)-prefix p of a k-mer;  cnt[b] / jc[b] = values of k-mer p+b
//   LEFT  table: key = the (K-1)-suffix s of a k-mer;  cnt[b] / jc[b] = values of k-mer b+s
// so the four successors of a Trail tip (getNextCounts, Jellyfish.cpp:308-321) are ONE aligned
// 32-byte HBM access instead of four random ones, and a point lookup (getCount) is one access
// to the RIGHT table.  cnt[b]==0 means "absent" (stored counts are >= min_count >= 1).
// Junction colours are < coloured_count_thr <= 65535 (Jellyfish.cpp:64,284), hence 16 bits.
struct __attribute__((aligned(32))) Bucket {
  uint64_t key;      // packed (K-1)-mer; kEmptyKey if unused
  uint32_t cnt[4];
  uint16_t jc[4];
};
static_assert(sizeof(Bucket) == 32, "bucket must be 32 bytes");

static const uint64_t kEmptyKey = ~0ULL;
// A key takes 2 (K - 1) <= 60 bits.  The top three bits of a RIGHT bucket's key word on the DEVICE hold the in-degree of
// its (K-1)-mer — how many of the four predecessors b + key have a count >= the table's MIN_COUNT, i.e. what the LEFT
// bucket of the same key would say (k_build_indegree, at upload) — so that the coverage kernel learns a k-mer's count
// and its left degree from ONE bucket.  Every key comparison masks them (an empty key never equals a masked key).
static const uint64_t kKeyMask = (1ULL << 61) - 1;
static const int kKeyDegShift = 61;

// coverage word pair per k-mer position: {count, colour (16 bits) | out-degrees}.  For a k-mer that is in the table
// (count != 0) k_coverage adds its out-degree towards RIGHT and towards LEFT (0..4 each) and the "known" flag.
static const uint32_t kCovColourMask = 0xFFFFu, kCovDegKnown = 1u << 22;
static const int kCovDegRShift = 16, kCovDegLShift = 19;

// The coverage of a read in HBM — the reference's vector<colouredCount> (Read.cpp:174-195), nine tenths of which is
// (0, 0) on a noisy read — is kept as the HITS only:
//   * one CovWord per 64 k-mer positions: bit i of `bits` = the k-mer at position 64 w + i is in the table (its count is
//     then >= MIN_COUNT: nothing below is ever stored, talc_table_host.h / talc_kernels_build.h, and a context only
//     runs with the table's own MIN_COUNT), `rank` = number of hits of the same TILE (kCovTile positions) before this
//     word;
//   * the hits' {count, colour | degrees} pairs, in position order, packed at the START of their tile's stretch of the
//     read's dense slot (pair i of the tile that starts at position p0 sits at index p0 + i), so the tiles of a read
//     — one workgroup each in k_coverage — need no offsets from each other.
// A read's words start at word (koff >> 6) + r of the batch's word array, koff = its first k-mer's index in the batch and
// r its number in the batch (that is at least the words of all reads before it).  The dense form exists only in
// talc_batch_fetch_coverage.  What this saves: the kernel the metric names wrote 1.25 GB of zeros per 100 k reads and the
// structure kernel read them back.
#ifndef TALC_COV_TILE
#define TALC_COV_TILE 512
#endif
struct __attribute__((aligned(16))) CovWord { uint64_t bits; uint32_t rank; uint32_t pad; };
TALC_HD uint64_t cov_word_base(uint64_t koff, uint32_t read_index) { return (koff >> 6) + read_index; }
TALC_HD uint64_t cov_words_total(uint64_t n_kmers, uint32_t n_reads) { return (n_kmers >> 6) + n_reads + 1; }

// Walk table (device only, derived from a finished Bucket table, same capacity and slot order): what a Trail that
// keeps following its only solid successor will meet over the next WALK_LEVELS steps, in one 32-byte record — the size
// of a bucket, so the two walk tables cost as much device memory as the two bucket tables (64-byte records with 32-bit
// levels, rounds 1-2, cost twice that: 141 GB behind 70 GB of buckets at 547 M k-mers) — so that the single-Trail
// fast-forward pays one dependent memory access per WALK_LEVELS steps instead of per step.
// Level 0 describes the bucket's own four counts, level j+1 the bucket reached from level j by appending level j's
// largest-count base.  One 16-bit word per level: bits 0-12 top = the largest count (0x1FFF: does not fit, the walk
// stops at this level), bit 13 single = "exactly one successor with count >= MIN_COUNT" (top >= MIN_COUNT > the largest
// of the other three) for the MIN_COUNT the table was filtered with — the only one a context may run it with
// (talc_ctx_create) —, bits 14-15 the base of `top`.  The only kind of step the fast-forward takes is a level with
// `single` set: that successor is the stored base and its count is top; a level whose bucket does not exist is all
// zero (no step passes it).  Whatever a record cannot express (a count beyond 13 bits) is left to the per-step form /
// the generic step: never wrong, only slower.
#define TALC_WALK_LEVELS 12
struct __attribute__((aligned(32))) WalkEntry {
  uint64_t key;                      // the bucket's key (kEmptyKey if unused)
  uint16_t lvl[TALC_WALK_LEVELS];    // top | single << 13 | base << 14 per level
};
static_assert(sizeof(WalkEntry) == 32, "walk entry must be 32 bytes");
static const uint32_t kWalkTopNone = 0x1FFFu;
static const uint32_t kWalkSingle = 1u << 13;
static const int kWalkBaseShift = 14;

struct TableView {
  const Bucket* right;   // device (or host) pointer
  const Bucket* left;
  uint64_t capacity;     // buckets per table (any size below 2^32; home = table_home(key, capacity))
  uint32_t k;
  // presence filter over the stored K-mers (blocked Bloom: three bits of one 64-bit word per k-mer, ~20 bits per
  // k-mer, 64-byte blocks chosen by the k-mer's minimizer): small enough to stay in the last-level cache, it answers
  // most lookups of k-mers that are NOT in the table (93 % of a noisy read's k-mers) without touching the table.
  // nullptr / 0 = no filter.  filterWords is a multiple of 8.
  const uint64_t* filter;
  uint64_t filterWords;
  const WalkEntry* walkRight;   // walk tables of `right` / `left`; nullptr = not built (the fast-forward probes per step)
  const WalkEntry* walkLeft;
};

TALC_HD uint64_t mix64(uint64_t x) {
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdULL;
  x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ULL;
  x ^= x >> 33;
  return x;
}

// Table hash: fold the high half of the key down, one multiply (every base of the key reaches the top bits); the
// home slot is the high product of the top 32 hash bits and the capacity (capacity < 2^32 buckets = 137 GB per
// table).  A lookup's address costs a handful of scalar instructions, which is what the path walk is made of.
TALC_HD uint64_t table_hash(uint64_t key) { const uint64_t x = key ^ (key >> 29); return x * 0x9E3779B97F4A7C15ULL; }
TALC_HD uint64_t table_slot(uint64_t h, uint64_t cap) { return ((h >> 32) * (uint64_t)(uint32_t)cap) >> 32; }
TALC_HD uint64_t table_home(uint64_t key, uint64_t cap) { return table_slot(table_hash(key), cap); }

// Presence filter over the stored k-mers: an array of 64-byte blocks (8 words); a k-mer sets / tests 3 bits of one
// word of one block.  Builder (k_build_filter) and prober (k_coverage) share these functions, so the filter has no
// false negatives by construction; a false positive only costs a bucket probe.
//   * The hash is built from 24-bit multiplies (full rate on the vector unit; 32- and 64-bit integer multiplies are
//     quarter rate, and the probe kernel turned out to be bound by them, not by memory).
//   * The BLOCK is chosen by the k-mer's minimizer (the smallest hash among its K - M + 1 M-mers), not by the k-mer's
//     own hash: consecutive k-mers of a read share their minimizer, hence their 64-byte block, for (K - M + 2) / 2
//     positions on average, and the lanes of a wave probe consecutive positions — their loads of one line become one
//     request to the L2.  Measured on config 2: the filter part of the probe kernel is bound by the number
//     of L2-missing requests (3.1 ms with one line per k-mer, 1.0 ms with shared lines).  M must be long enough that an
//     M-mer is (nearly) unique in the transcriptome: with M = 11 (4 M possible M-mers for 54 M k-mers) every popular
//     minimizer sent all its occurrences' k-mers to one block, the blocks' loads were badly skewed and the false
//     positives — each a wasted bucket probe — cost more than the shared lines saved.  M = K - 7 clamped to [12, 16].
//     TALC_FILTER_MINIMIZER = 0 builds the one-line-per-k-mer form (experiments).
#ifndef TALC_FILTER_MINIMIZER
#define TALC_FILTER_MINIMIZER 1
#endif
#ifndef TALC_MINIMIZER_SPAN
#define TALC_MINIMIZER_SPAN 7   /* M = K - span: a k-mer has span + 1 M-mers */
#endif
TALC_HD uint32_t filter_mmer_len(uint32_t K) {
  uint32_t m = K > TALC_MINIMIZER_SPAN ? K - TALC_MINIMIZER_SPAN : 1;
  m = m < 12 ? 12 : (m > 16 ? 16 : m);
  return m < K ? m : K;
}
TALC_HD uint32_t mul24(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umul24(a, b);
#else
  return (a & 0xFFFFFFu) * (b & 0xFFFFFFu);
#endif
}
TALC_HD uint32_t mmer_hash(uint32_t mmer) {   // M-mer of up to 32 bits -> 32-bit hash (0xFFFFFFFF is reserved: "contains N")
  uint32_t x = mul24(mmer, 0x9E3779u) ^ mul24(mmer >> 11, 0x85EBCBu) ^ (mmer >> 7);
  x ^= x >> 15; x = mul24(x, 0xC2B2AFu) ^ (x >> 9); x ^= x >> 13;
  return x == 0xFFFFFFFFu ? 0xFFFFFFFEu : x;
}
// minimizer hash of a packed k-mer (first base most significant): the prober takes the same minimum from LDS
TALC_HD uint32_t kmer_min_hash(uint64_t kmer, uint32_t K) {
  const uint32_t M = filter_mmer_len(K);
  const uint64_t mm = (M >= 32) ? ~0ULL : ((1ULL << (2 * M)) - 1);
  uint32_t best = 0xFFFFFFFFu;
  for (uint32_t i = 0; i + M <= K; ++i) {
    const uint32_t h = mmer_hash((uint32_t)((kmer >> (2 * (K - M - i))) & mm));
    best = h < best ? h : best;
  }
  return best;
}
// the block of a minimizer hash (the minimum of several hashes is not uniform: mixed again)
TALC_HD uint64_t filter_block_of_min(uint32_t mh, uint64_t nBlocks) {
  mh = mul24(mh, 0x2C1B3Du) ^ (mh >> 11); mh ^= mh >> 15; mh = mul24(mh, 0x297A2Du) ^ (mh >> 8);
  return ((uint64_t)mh * nBlocks) >> 32;
}
// two 32-bit hash words of a k-mer: .x selects the block (unless by minimizer), .y the word and the three bits
struct FilterHash { uint32_t x, y; };
TALC_HD FilterHash filter_hash(uint64_t kmer) {
  const uint32_t lo = (uint32_t)kmer, hi = (uint32_t)(kmer >> 32);
  uint32_t a = mul24(lo, 0x9E3779u) + mul24((lo >> 24) | (hi << 8), 0x85EBCBu) + mul24(hi >> 16, 0xC2B2AFu);
  a ^= a >> 15;
  uint32_t b = mul24(a, 0x2C1B3Du) ^ (a >> 9);
  b ^= b >> 13;
  const uint32_t c = mul24(b, 0x297A2Du) ^ ((a << 7) | (a >> 25)) ^ (b >> 11);
  FilterHash h; h.x = b; h.y = c;
  return h;
}
TALC_HD uint64_t filter_block(uint32_t h, uint64_t nBlocks) { return ((uint64_t)h * nBlocks) >> 32; }   // nBlocks < 2^32
TALC_HD uint64_t filter_index(uint64_t kmer, uint32_t K, FilterHash h, uint64_t nBlocks) {
#if TALC_FILTER_MINIMIZER
  return filter_block_of_min(kmer_min_hash(kmer, K), nBlocks) * 8 + (h.y >> 29);
#else
  return filter_block(h.x, nBlocks) * 8 + (h.y >> 29);
#endif
}
TALC_HD uint64_t filter_mask(FilterHash h) { return (1ULL << (h.y & 63)) | (1ULL << ((h.y >> 6) & 63)) | (1ULL << ((h.y >> 12) & 63)); }

// ------------------------------------------------------------------ Dna5 codes
// reads live in HBM as one byte per base: A=0 C=1 G=2 T=3 N=4 (SeqAn Dna5 ordinals)
enum : uint8_t { BASE_A = 0, BASE_C = 1, BASE_G = 2, BASE_T = 3, BASE_N = 4 };

TALC_HD uint8_t ascii_to_code(uint8_t c) {
  switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return 4;
  }
}
TALC_HD uint8_t complement_code(uint8_t c) { return c < 4 ? (uint8_t)(3 - c) : (uint8_t)4; }
TALC_HD char code_to_ascii(uint8_t c) { return c == 0 ? 'A' : c == 1 ? 'C' : c == 2 ? 'G' : c == 3 ? 'T' : 'N'; }

// Device copy of the parameters (talc_params + derived values)
struct DevParams {
  uint32_t K;
  uint32_t MIN_COUNT;
  double ALPHA;
  uint32_t WINDOW;
  double ERR;
  double MIN_INNER;
  double MIN_BORDER;
  uint32_t MAXB;
  int32_t reverse;
  uint32_t MIN_START_ANCHORS, MAX_START_ANCHORS, MAX_IN_COUNT, MAX_BORDER_PATHS, MAX_INNER_PATHS, CHECK_INTERVAL;
  double FAILURE_RATE;
  int32_t MAX_BORDER_FAILURES;
  uint32_t MAX_BORDER_LEN;
  // work estimate of an edge of h bases, in wave-cycles (only the order of the work queue depends on it; k_structure)
  uint32_t costEdgeLin, costEdgeQuad;
  // ... and the long-gap risk term: (costGapQuad + costGapFork x share of forking solid k-mers) x min(sum g^2, costGapCap^2)
  uint32_t costGapQuad, costGapFork, costGapCap, pad_;
  // the count model's thresholds by count (device memory, built per context by k_build_thresholds from the formula itself
  // for this ALPHA): thr[2 c] = the smallest nextc that isExpectedbyMyModel(nextc, c, false) accepts, thr[2 c + 1] = how many
  // nextc = 0, 1, ... isExpectedbyMyModel(nextc, c, true) accepts; c < thrN (counts beyond take the formula)
  const uint32_t* thr;
  uint32_t thrN, pad2_;
};

}  // namespace talc
