// talc_kernels_probe.h — device-side table probes and the k-mer coverage kernel
// (replaces getCountFromDBG / getNextCountsFromDBG / getLRCountsInSRFromDBG,
//  Jellyfish.cpp:407-413, 308-321, 485-496, and Read::reCoverage, Read.cpp:174-195).
#pragma once
#include "talc_common.h"

namespace talc {

TALC_D uint64_t dev_home(uint64_t key, uint64_t cap) { return table_home(key, cap); }

// One bucket = two 16-byte loads from the same 32-byte sector.
struct BucketRegs {
  uint64_t key;
  uint32_t cnt[4];
  uint32_t jc01, jc23;  // packed u16 pairs
  TALC_D uint32_t jc(int b) const { uint32_t w = (b < 2) ? jc01 : jc23; return (b & 1) ? (w >> 16) : (w & 0xffffu); }
  // cnt[b] for a lane-dependent b as selects (an indexed read of a register array goes through the private stack)
  TALC_D uint32_t count_of(int b) const { const uint32_t lo = (b & 1) ? cnt[1] : cnt[0], hi = (b & 1) ? cnt[3] : cnt[2]; return (b & 2) ? hi : lo; }
};

TALC_D BucketRegs load_bucket(const Bucket* p) {
  const v4u32 TALC_AS1* q = (const v4u32 TALC_AS1*)p;   // the table is always in HBM
  const v4u32 a = q[0], b = q[1];
  BucketRegs r;
  r.key = ((uint64_t)a.y << 32) | a.x;
  r.cnt[0] = a.z; r.cnt[1] = a.w; r.cnt[2] = b.x; r.cnt[3] = b.y;
  r.jc01 = b.z; r.jc23 = b.w;
  return r;
}

// Linear probing; returns true and fills `out` when the (K-1)-mer key is present.
TALC_D bool probe_bucket(const Bucket* tab, uint64_t cap, uint64_t key, BucketRegs& out) {
  uint64_t i = dev_home(key, cap);
  while (true) {
    BucketRegs r = load_bucket(tab + i);
    if ((r.key & kKeyMask) == key) { out = r; return true; }
    if (r.key == kEmptyKey) return false;
    if (++i == cap) i = 0;
  }
}

// the same, the home bucket `r` (slot i) having been loaded by the caller (several probes' first loads in flight together)
TALC_D bool probe_bucket_from(const Bucket* tab, uint64_t cap, uint64_t key, uint64_t i, BucketRegs& r) {
  while (true) {
    if ((r.key & kKeyMask) == key) return true;
    if (r.key == kEmptyKey) return false;
    if (++i == cap) i = 0;
    r = load_bucket(tab + i);
  }
}

// getCount (Jellyfish.cpp:407-413) for a packed K-mer
TALC_D void dev_get_count(const TableView& T, uint64_t kmer, uint32_t& cnt, uint32_t& jc) {
  BucketRegs r;
  cnt = 0; jc = 0;
  if (probe_bucket(T.right, T.capacity, kmer >> 2, r)) { const int b = (int)(kmer & 3); cnt = r.count_of(b); jc = r.jc(b); }
}

// getNextCounts (Jellyfish.cpp:308-321): the 4 successors of `kmer` walking RIGHT
// (drop first base, append b) or LEFT (prepend b, drop last), order A,C,G,T.
TALC_D void dev_next_counts(const TableView& T, uint64_t kmer, int dirRight, uint32_t cnt[4], uint32_t jc[4]) {
  const uint64_t m1 = (1ULL << (2 * (T.k - 1))) - 1;
  BucketRegs r;
  bool ok;
  if (dirRight) ok = probe_bucket(T.right, T.capacity, kmer & m1, r);
  else ok = probe_bucket(T.left, T.capacity, kmer >> 2, r);
#pragma unroll
  for (int b = 0; b < 4; ++b) { cnt[b] = ok ? r.cnt[b] : 0u; jc[b] = ok ? r.jc(b) : 0u; }
}

// ------------------------------------------------------------------ test-hook kernels
__global__ void k_lookup(TableView T, const uint64_t* kmers, uint64_t n, uint32_t* counts, uint32_t* jcounts) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t c, j;
  dev_get_count(T, kmers[i], c, j);
  counts[i] = c; jcounts[i] = j;
}
__global__ void k_next_counts(TableView T, const uint64_t* kmers, uint64_t n, int dirRight, uint32_t* counts4,
                              uint32_t* jcounts4) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t c[4], j[4];
  dev_next_counts(T, kmers[i], dirRight, c, j);
  for (int b = 0; b < 4; ++b) { counts4[4 * i + b] = c[b]; jcounts4[4 * i + b] = j[b]; }
}

// ------------------------------------------------------------------ encode
// raw ASCII -> Dna5 codes (SeqAn Dna5 conversion), with the -rev reverse complement of
// main.cpp:253 applied per read.  One block per (read, 4096-base chunk).
__global__ void k_encode(const uint8_t* __restrict__ raw, uint8_t* __restrict__ codes, const uint64_t* __restrict__ offsets,
                         const uint32_t* __restrict__ chunk_read, const uint32_t* __restrict__ chunk_start, int reverse) {
  const uint32_t r = chunk_read[blockIdx.x];
  const uint64_t b = offsets[r], e = offsets[r + 1];
  const uint64_t L = e - b;
  const uint64_t s0 = chunk_start[blockIdx.x];
  for (uint64_t i = s0 + threadIdx.x; i < s0 + 4096 && i < L; i += blockDim.x) {
    uint8_t c = ascii_to_code(raw[b + i]);
    if (reverse) codes[b + (L - 1 - i)] = complement_code(c);
    else codes[b + i] = c;
  }
}

// ------------------------------------------------------------------ coverage (the k-mer probe kernel)
// One block per tile of up to COV_TILE consecutive k-mer positions of ONE read, in two phases:
//  A. the tile's base window (COV_TILE + K - 1 codes) is staged into LDS as 2-bit packed words (first base most
//     significant, so a k-mer is one funnel shift away from its table form) plus an N bitmap; thread t takes positions
//     t, t+256, ...: k-mer, presence-filter test (one 8-byte word of a block its minimizer chooses: lanes share lines);
//     the 7-9 % that may be in the table go to an LDS queue, the rest leave no trace at all;
//  B. the queue is worked off with every lane busy: the 32-byte bucket of the k-mer's (K-1)-prefix — the one random HBM
//     access of a lookup: it holds the count, the k-mer's left degree (key word's top bits) and the right degree of the
//     position before (published through LDS; cov_count below).  The hits set their bit in the tile's 32 bitmap words
//     (LDS), the words' ranks are a 32-element scan, and every hit's pair goes to its rank at the start of the tile's
//     stretch (talc_common.h: CovWord): only hits are written, contiguously.
//  #{count > MIN_COUNT} (Read.cpp:190) is reduced per block and added to the read's counter.
#ifndef TALC_COV_EXP
#define TALC_COV_EXP 0   /* timing experiments (results wrong on purpose): bit 0 = no table traffic, bit 1 = no filter traffic */
#endif
#define COV_TILE TALC_COV_TILE
#ifndef COV_THREADS
#define COV_THREADS 64   /* one wave per tile of 512 positions: 32 independent tiles per CU, no wave waits at another's barrier (1.66 ms with 256 threads on 2048 positions, 1.37 ms so; config 2) */
#endif

// The table part of one lookup, for a k-mer that passed the filter: its count and colour, and — for a k-mer of the
// table — its out-degrees in both directions (getOutDegree, Jellyfish.cpp:383-393, for this MIN_COUNT), which ride in
// the colour word's upper half: the anchor search asks for them position by position (Explorer.cpp:449,515) and would
// otherwise probe, one dependent access at a time.  Three buckets would be involved — RIGHT[prefix] (the count),
// LEFT[prefix] (the predecessors' counts), RIGHT[suffix] (the successors' counts) — but RIGHT[suffix] of position p is
// RIGHT[prefix] of position p + 1, which that position's own lookup reads anyway when it is in the queue too (nine in
// ten positions inside a solid region): cov_count publishes the successor degree its bucket implies for the position
// before it, and cov_degrees only probes when nobody did.
TALC_D uint32_t bucket_degree(const BucketRegs& r, uint32_t min_count) {
  return (r.cnt[0] >= min_count) + (r.cnt[1] >= min_count) + (r.cnt[2] >= min_count) + (r.cnt[3] >= min_count);
}
// step 1: count, colour, left degree; degPrev = the right degree of the PREVIOUS position's k-mer.  ONE bucket: the
// RIGHT bucket of the k-mer's prefix holds the count, its key word's top bits the prefix's in-degree (= the k-mer's left
// degree: what LEFT[prefix] would say, talc_common.h), its four counts the previous position's right degree.
TALC_D void cov_count(const TableView& T, uint64_t kmer, uint32_t min_count, uint32_t& c, uint32_t& j, uint32_t& dL, uint32_t& degPrev) {
  c = 0; j = 0; dL = 0; degPrev = 0;
  const uint64_t kp = kmer >> 2;
  BucketRegs rc;
  if (!probe_bucket(T.right, T.capacity, kp, rc)) return;            // no successor of that (K-1)-mer at all
  degPrev = bucket_degree(rc, min_count);
  const int b = (int)(kmer & 3);
  c = rc.count_of(b); j = rc.jc(b);
  if (c != 0) dL = (uint32_t)(rc.key >> kKeyDegShift);
}
// step 2 (only for c != 0, when the next position did not publish it): the right degree by its own probe
TALC_D uint32_t cov_right_degree(const TableView& T, uint64_t kmer, uint32_t min_count) {
  const uint32_t K = T.k;
  const uint64_t m1 = (K >= 32) ? ~0ULL : ((1ULL << (2 * (K - 1))) - 1);
  BucketRegs br;
  return probe_bucket(T.right, T.capacity, kmer & m1, br) ? bucket_degree(br, min_count) : 0u;
}
#ifndef COV_WAVES_PER_SIMD
#define COV_WAVES_PER_SIMD 8
#endif
__global__ void __launch_bounds__(COV_THREADS, COV_WAVES_PER_SIMD * 64 / COV_THREADS > 0 ? COV_WAVES_PER_SIMD * 64 / COV_THREADS : 1)
k_coverage(TableView T, const uint8_t* __restrict__ codes, const uint64_t* __restrict__ offsets,
           const uint64_t* __restrict__ koff, const uint32_t* __restrict__ tile_read,
           const uint32_t* __restrict__ tile_start, uint2* __restrict__ cov, CovWord* __restrict__ covWords,
           int32_t* __restrict__ n_in, uint32_t min_count) {
  __shared__ uint64_t s_pack[(COV_TILE + 64) / 32 + 3];   // base i of the window at bits [63 - 2 (i % 32) - 1, 63 - 2 (i % 32)] of word i / 32
  __shared__ uint64_t s_nmask[(COV_TILE + 64) / 64 + 2];  // bit (i % 64) of word i / 64: base i is N
#if TALC_FILTER_MINIMIZER
  __shared__ uint32_t s_mh[COV_TILE + 64];                // hash of the M-mer starting at each window position
#endif
  __shared__ uint16_t s_queue[COV_TILE];                  // positions whose k-mer passed the filter
  __shared__ uint8_t s_degR[COV_TILE + 4];                // right degree of the k-mer at a position, published by the next position's lookup (0xFF: not)
  __shared__ unsigned long long s_bits[COV_TILE / 64];   // the tile's hit bitmap
  __shared__ uint32_t s_rank[COV_TILE / 64];             // hits of the tile before each word
  __shared__ uint32_t s_qn, s_anyN;
  __shared__ int s_nin;
  const uint32_t K = T.k;
  const uint32_t r = tile_read[blockIdx.x];
  const uint32_t p0 = tile_start[blockIdx.x];
  const uint64_t rb = offsets[r];
  const uint32_t L = (uint32_t)(offsets[r + 1] - rb);
  const uint32_t nk = L - K + 1;                       // k-mers in this read (host guarantees L >= K)
  const uint32_t cnt = min((uint32_t)COV_TILE, nk - p0);  // positions in this tile
  const uint32_t wlen = cnt + K - 1;                  // bases in the window
  const uint8_t TALC_AS1* src = (const uint8_t TALC_AS1*)(codes + rb + p0);

  if (threadIdx.x == 0) { s_nin = 0; s_qn = 0; s_anyN = 0; }
  __syncthreads();
  bool tileHasN = false;
  if (threadIdx.x < COV_TILE / 64) s_bits[threadIdx.x] = 0ull;
  for (uint32_t i = threadIdx.x; i < (COV_TILE + 4) / 4; i += COV_THREADS) reinterpret_cast<uint32_t*>(s_degR)[i] = 0xFFFFFFFFu;
  // stage: 16 bases -> one u32 of 2-bit codes (first base in the top bits) + 16 N bits, per thread per pass
  {
    uint32_t* pk32 = reinterpret_cast<uint32_t*>(s_pack);
    uint16_t* nm16 = reinterpret_cast<uint16_t*>(s_nmask);
    const uint32_t ngroups = (wlen + 15) / 16;
    for (uint32_t g = threadIdx.x; g < ngroups + 6; g += COV_THREADS) {   // (+ zeroed guard groups for the funnel shifts)
      uint32_t w = 0, nm = 0;
      const uint32_t base = g * 16;
      if (g < ngroups) {
        // 16 codes as two unaligned 8-byte loads (whole words only inside the window; the last group byte by byte)
        typedef uint64_t __attribute__((aligned(1))) u64u;
        uint64_t lo8 = 0, hi8 = 0;
        if (base + 16 <= wlen) {
          lo8 = *(const u64u TALC_AS1*)(src + base);
          hi8 = *(const u64u TALC_AS1*)(src + base + 8);
        } else {
          for (uint32_t j = 0; j < 16 && base + j < wlen; ++j) {
            const uint64_t c = (uint64_t)src[base + j];
            if (j < 8) lo8 |= c << (8 * j); else hi8 |= c << (8 * (j - 8));
          }
        }
#pragma unroll
        for (uint32_t j = 0; j < 16; ++j) {
          const uint32_t c = (uint32_t)(((j < 8) ? (lo8 >> (8 * j)) : (hi8 >> (8 * (j - 8)))) & 0xFFu);
          nm |= (c > 3u ? 1u : 0u) << j;
          w |= (c & 3u) << (30 - 2 * j);
        }
      }
      pk32[g ^ 1] = w;          // even group = high half of its 64-bit word
      nm16[g] = (uint16_t)nm;
      tileHasN |= (nm != 0u);
    }
  }
  // (nearly every tile is free of N: its positions then skip the N-mask reads of both passes)
  if (__ballot(tileHasN) != 0ull && (threadIdx.x & 63u) == 0u) s_anyN = 1u;
  __syncthreads();
  const bool anyN = s_anyN != 0u;

  // the 64 window bits that start with base q (first base most significant), and the N bits [q, q + 64)
  auto window = [&](uint32_t q) -> uint64_t {
    const uint32_t w = q >> 5, sh = 2 * (q & 31);
    const uint64_t hi = s_pack[w], lo = s_pack[w + 1];
    return (sh == 0) ? hi : ((hi << sh) | (lo >> (64 - sh)));
  };
  auto nbits = [&](uint32_t q) -> uint64_t {
    const uint32_t nw = q >> 6, nsh = q & 63;
    const uint64_t nlo = s_nmask[nw], nhi = s_nmask[nw + 1];
    return (nsh == 0) ? nlo : ((nlo >> nsh) | (nhi << (64 - nsh)));
  };
#if TALC_FILTER_MINIMIZER
  const uint32_t M = filter_mmer_len(K);
  const uint32_t nmm = wlen - M + 1;
  for (uint32_t q = threadIdx.x; q < nmm; q += COV_THREADS)
    s_mh[q] = (anyN && (nbits(q) & ((1ull << M) - 1))) ? 0xFFFFFFFFu : mmer_hash((uint32_t)(window(q) >> (64 - 2 * M)));
  __syncthreads();
  const uint32_t nwin = K - M + 1;                   // M-mers per k-mer
#endif

  const uint32_t kshift = 64 - 2 * K;
  const uint64_t nkmask = (K >= 64) ? ~0ULL : ((1ULL << K) - 1);
  const uint64_t nBlocks = T.filterWords >> 3;
  const uint64_t TALC_AS1* filter = (const uint64_t TALC_AS1*)T.filter;
  v2u32 TALC_AS1* out = (v2u32 TALC_AS1*)(cov + koff[r] + p0);
  int local_in = 0;
  // ---- phase A.  The filter words of all of a thread's positions are requested before the first is looked at (a tile
  // is at most COV_TILE / COV_THREADS = 8 positions per thread): one memory round trip per tile instead of one per pass.
  constexpr int NPASS = COV_TILE / COV_THREADS;
  uint64_t fword[NPASS], fmask[NPASS];
#pragma unroll
  for (int it = 0; it < NPASS; ++it) {
    const uint32_t p = (uint32_t)it * COV_THREADS + threadIdx.x;
    fword[it] = 0; fmask[it] = 1;                  // (p beyond the tile or an N in the k-mer: fails the test below)
    if (p < cnt && !(anyN && (nbits(p) & nkmask) != 0)) {     // no N among bases [p, p+K)
      fword[it] = ~0ULL;
      if (filter) {
        const uint64_t kmer = window(p) >> kshift;
        const FilterHash h = filter_hash(kmer);
        fmask[it] = filter_mask(h);
#if TALC_FILTER_MINIMIZER
        uint32_t mh = s_mh[p];
        for (uint32_t i = 1; i < nwin; ++i) mh = min(mh, s_mh[p + i]);
        const uint64_t idx = filter_block_of_min(mh, nBlocks) * 8 + (h.y >> 29);
#else
        const uint64_t idx = filter_block(h.x, nBlocks) * 8 + (h.y >> 29);
#endif
#if (TALC_COV_EXP & 2)   /* timing experiment: no filter traffic, the same share of survivors */
        fword[it] = ((h.x & 15u) == 0u) ? ~0ULL : 0ULL; (void)idx;
#else
        fword[it] = filter[idx];
#endif
      }
    }
  }
#pragma unroll
  for (int it = 0; it < NPASS; ++it) {
    const uint32_t pb = (uint32_t)it * COV_THREADS;
    if (pb >= cnt) break;
    const bool maybe = (fword[it] & fmask[it]) == fmask[it];
    // queue the survivors: one LDS atomic per wave
    const unsigned long long bal = __ballot(maybe);
    if (bal) {
      const uint32_t lane = threadIdx.x & 63u;
      uint32_t base = 0;
      if (lane == 0) base = atomicAdd(&s_qn, (uint32_t)__popcll(bal));
      base = (uint32_t)__shfl((int)base, 0, 64);
      if (maybe) s_queue[base + (uint32_t)__popcll(bal & ((1ull << lane) - 1))] = (uint16_t)(pb + threadIdx.x);
    }
  }
  __syncthreads();
  // ---- phase B.  A thread keeps the results of its (at most NPASS, nearly always one) queue entries in registers until
  // the tile's ranks are known.
  const uint32_t qn = s_qn;
  uint32_t myP[NPASS], myC[NPASS], myJ[NPASS];
#pragma unroll
  for (int it = 0; it < NPASS; ++it) {
    myP[it] = 0; myC[it] = 0; myJ[it] = 0;
    const uint32_t qi = (uint32_t)it * COV_THREADS + threadIdx.x;
    if ((uint32_t)it * COV_THREADS >= qn) break;   // (block-uniform)
    if (qi < qn) {
      const uint32_t p = s_queue[qi];
      const uint64_t kmer = window(p) >> kshift;
      uint32_t c = 0, j = 0, dL = 0;
#if (TALC_COV_EXP & 1)   /* timing experiment: no table traffic */
      (void)dL; (void)kmer;
#else
      uint32_t degPrev;
      cov_count(T, kmer, min_count, c, j, dL, degPrev);
      if (p > 0) s_degR[p - 1] = (uint8_t)degPrev;
#endif
      if (c != 0) {
        atomicOr(&s_bits[p >> 6], 1ull << (p & 63u));
        j |= kCovDegKnown | (dL << kCovDegLShift);
      }
      myP[it] = p; myC[it] = c; myJ[it] = j;
    }
  }
  __syncthreads();
  if (threadIdx.x < 64) {   // ranks: exclusive scan of the 32 words' populations (first wave)
    const uint32_t w = threadIdx.x;
    const uint32_t pc = (w < COV_TILE / 64) ? (uint32_t)__popcll(s_bits[w < COV_TILE / 64 ? w : 0]) : 0u;
    uint32_t incl = pc;
#pragma unroll
    for (int off = 1; off < COV_TILE / 64; off <<= 1) {
      const uint32_t o = (uint32_t)__shfl_up((int)incl, off, 64);
      if ((int)w >= off) incl += o;
    }
    if (w < COV_TILE / 64) s_rank[w] = incl - pc;
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < NPASS; ++it) {
    if ((uint32_t)it * COV_THREADS >= qn) break;
    const uint32_t c = myC[it];
    if (c != 0) {
      const uint32_t p = myP[it];
      uint32_t dR = s_degR[p];
      if (dR == 0xFFu) dR = cov_right_degree(T, window(p) >> kshift, min_count);
      const uint32_t idx = s_rank[p >> 6] + (uint32_t)__popcll(s_bits[p >> 6] & ((1ull << (p & 63u)) - 1ull));
      out[idx] = v2u32{c, myJ[it] | (dR << kCovDegRShift)};
      local_in += (c > min_count) ? 1 : 0;
    }
  }
  {   // the tile's words of the read's bitmap (every word of the tile, hit or not)
    const uint32_t nw = (cnt + 63u) >> 6;
    if (threadIdx.x < nw) {
      v4u32 TALC_AS1* wout = (v4u32 TALC_AS1*)(covWords + cov_word_base(koff[r], r) + (p0 >> 6));
      const unsigned long long bw = s_bits[threadIdx.x];
      wout[threadIdx.x] = v4u32{(uint32_t)bw, (uint32_t)(bw >> 32), s_rank[threadIdx.x], 0u};   // {bits, rank, pad}
    }
  }
  // block reduction of local_in
  for (int off = 32; off > 0; off >>= 1) local_in += __shfl_down(local_in, off, 64);
  if ((threadIdx.x & 63) == 0 && local_in) atomicAdd(&s_nin, local_in);
  __syncthreads();
  if (threadIdx.x == 0 && s_nin) atomicAdd(&n_in[r], s_nin);
}

}  // namespace talc
