// talc_kernels_build.h — the k-mer table built on the device (replaces the insert loop of buildCDBG,
// Jellyfish.cpp:251-269, for arrays of packed k-mers).
//
// Semantics to keep: only k-mers with count >= min_count are stored, and of several lines with the same
// k-mer the FIRST one wins (std::map::insert, Jellyfish.cpp:262).  Four passes over the n entries:
//   claim     every kept entry finds or claims the bucket of its (K-1)-mer key in both tables (64-bit CAS, linear
//             probing from the same home slot the lookups use) and bids for its count slot with atomicMin(index+1)
//             — the count words, preset to 0xFFFFFFFF, hold the smallest bidding index after this pass;
//   resolve   an entry whose index is not the one left in the slot lost to an earlier line: it forgets its slot;
//   write     the winners store their counts;
//   finalize  untouched count words and colour fields become 0; stored k-mers and buckets are counted.
// Bucket placement along a probe sequence depends on the order the CASes land in, which is not the host builder's
// order; lookups do not depend on it (no deletions, probing stops at the first empty bucket).
#pragma once
#include "talc_common.h"
#include "talc_kernels_probe.h"

namespace talc {

static constexpr uint32_t kNoSlot = 0xFFFFFFFFu;

TALC_D uint32_t build_claim_bucket(Bucket* tab, uint64_t cap, uint64_t key) {
  uint64_t i = table_home(key, cap);
  while (true) {
    unsigned long long* kp = (unsigned long long*)&tab[i].key;
    unsigned long long cur = __atomic_load_n(kp, __ATOMIC_RELAXED);
    if (cur == kEmptyKey) cur = atomicCAS(kp, (unsigned long long)kEmptyKey, (unsigned long long)key) == kEmptyKey ? key : __atomic_load_n(kp, __ATOMIC_RELAXED);
    if (cur == key) return (uint32_t)i;
    if (++i == cap) i = 0;
  }
}

__global__ void k_build_claim(Bucket* right, Bucket* left, uint64_t cap, uint32_t K, const uint64_t* __restrict__ kmers,
                              const uint32_t* __restrict__ counts, uint64_t n, uint32_t minc, uint32_t* __restrict__ slotR,
                              uint32_t* __restrict__ slotL) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (counts[i] < minc) { slotR[i] = kNoSlot; slotL[i] = kNoSlot; return; }
  const uint64_t km = kmers[i];
  const uint64_t m1 = (1ULL << (2 * (K - 1))) - 1;
  const uint32_t sr = build_claim_bucket(right, cap, km >> 2);
  atomicMin(&right[sr].cnt[km & 3], (uint32_t)(i + 1));
  const uint32_t sl = build_claim_bucket(left, cap, km & m1);
  atomicMin(&left[sl].cnt[(km >> (2 * (K - 1))) & 3], (uint32_t)(i + 1));
  slotR[i] = sr; slotL[i] = sl;
}

__global__ void k_build_resolve(const Bucket* __restrict__ right, const Bucket* __restrict__ left, uint32_t K,
                                const uint64_t* __restrict__ kmers, uint64_t n, uint32_t* __restrict__ slotR,
                                uint32_t* __restrict__ slotL) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t km = kmers[i];
  const uint32_t sr = slotR[i], sl = slotL[i];
  if (sr != kNoSlot && right[sr].cnt[km & 3] != (uint32_t)(i + 1)) slotR[i] = kNoSlot;
  if (sl != kNoSlot && left[sl].cnt[(km >> (2 * (K - 1))) & 3] != (uint32_t)(i + 1)) slotL[i] = kNoSlot;
}

__global__ void k_build_write(Bucket* right, Bucket* left, uint32_t K, const uint64_t* __restrict__ kmers,
                              const uint32_t* __restrict__ counts, uint64_t n, const uint32_t* __restrict__ slotR,
                              const uint32_t* __restrict__ slotL) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t km = kmers[i];
  const uint32_t sr = slotR[i], sl = slotL[i];
  if (sr != kNoSlot) right[sr].cnt[km & 3] = counts[i];
  if (sl != kNoSlot) left[sl].cnt[(km >> (2 * (K - 1))) & 3] = counts[i];
}

// stats: [0] stored k-mers (RIGHT table), [1] buckets RIGHT, [2] buckets LEFT
__global__ void k_build_finalize(Bucket* right, Bucket* left, uint64_t cap, unsigned long long* stats) {
  const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long nk = 0, nr = 0, nl = 0;
  if (j < cap) {
    for (int which = 0; which < 2; ++which) {
      Bucket* b = (which == 0 ? right : left) + j;
      const bool used = b->key != kEmptyKey;
      for (int c = 0; c < 4; ++c) {
        uint32_t v = b->cnt[c];
        if (!used || v == 0xFFFFFFFFu) v = 0;
        b->cnt[c] = v; b->jc[c] = 0;
        if (which == 0 && v != 0) ++nk;
      }
      if (used) { if (which == 0) ++nr; else ++nl; }
    }
  }
  // block reduction (256 threads)
  __shared__ unsigned long long s[3];
  if (threadIdx.x == 0) { s[0] = s[1] = s[2] = 0; }
  __syncthreads();
  for (int off = 32; off > 0; off >>= 1) {
    nk += (unsigned long long)__shfl_down((long long)nk, off, 64);
    nr += (unsigned long long)__shfl_down((long long)nr, off, 64);
    nl += (unsigned long long)__shfl_down((long long)nl, off, 64);
  }
  if ((threadIdx.x & 63) == 0) { atomicAdd(&s[0], nk); atomicAdd(&s[1], nr); atomicAdd(&s[2], nl); }
  __syncthreads();
  if (threadIdx.x == 0) { if (s[0]) atomicAdd(&stats[0], s[0]); if (s[1]) atomicAdd(&stats[1], s[1]); if (s[2]) atomicAdd(&stats[2], s[2]); }
}

// ------------------------------------------------------------------ junction colours on the device (Jellyfish.cpp:273-290)
// The reference walks the junction dump line by line and, for the k-mer and then for its reverse complement, sets the
// colour if the junction count is below colouredCountThr and the k-mer is in the table (:278-289) — so when several
// lines reach the same k-mer the LAST one in (line, strand) order wins.  Entry e = 2 * line + strand.  Three passes:
//   claim   every entry that passes the threshold and finds its k-mer bids for it with atomicMax(e + 1) in a scratch
//           hash keyed by the k-mer's count slot (RIGHT bucket index * 4 + last base);
//   write   the entry whose bid is the one left there stores its colour in both tables.
// (16-bit stores: the four colours of a bucket share two dwords, byte-enabled writes do not disturb their neighbours.)
TALC_D uint64_t dev_revcomp(uint64_t km, uint32_t K) {
  // complement = 3 - base = bitwise not of the 2-bit code; then reverse the 2-bit groups of the K-mer
  uint64_t x = ~km;
  x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2);
  x = ((x >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((x & 0x0F0F0F0F0F0F0F0FULL) << 4);
  x = __builtin_bswap64(x);
  return x >> (64 - 2 * K);
}
// slot of `key` in tab, or ~0 if absent
TALC_D uint64_t dev_find_slot(const Bucket* tab, uint64_t cap, uint64_t key) {
  uint64_t i = table_home(key, cap);
  while (true) {
    const uint64_t k = ((const uint64_t TALC_AS1*)&tab[i].key)[0];
    if ((k & kKeyMask) == key) return i;
    if (k == kEmptyKey) return ~0ULL;
    if (++i == cap) i = 0;
  }
}
static constexpr uint64_t kNoId = ~0ULL;

__global__ void k_colour_claim(const Bucket* __restrict__ right, uint64_t cap, uint32_t K, const uint64_t* __restrict__ jk,
                               const int64_t* __restrict__ jc, uint64_t n, uint32_t thr, uint64_t* __restrict__ ids,
                               unsigned long long* __restrict__ hkeys, uint32_t* __restrict__ hseq, uint64_t hmask) {
  const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= 2 * n) return;
  const uint64_t i = e >> 1;
  uint64_t id = kNoId;
  if ((unsigned int)(int)jc[i] < thr) {   // Jellyfish.cpp:284: int compared with unsigned
    const uint64_t km = (e & 1) ? dev_revcomp(jk[i], K) : jk[i];
    const uint64_t sr = dev_find_slot(right, cap, km >> 2);
    if (sr != ~0ULL && right[sr].cnt[km & 3] != 0) id = sr * 4 + (km & 3);
  }
  ids[e] = id;
  if (id == kNoId) return;
  uint64_t h = mix64(id) & hmask;
  while (true) {
    unsigned long long cur = __atomic_load_n(&hkeys[h], __ATOMIC_RELAXED);
    if (cur == kNoId) { const unsigned long long old = atomicCAS(&hkeys[h], (unsigned long long)kNoId, (unsigned long long)id); cur = (old == kNoId) ? id : old; }
    if (cur == id) break;
    h = (h + 1) & hmask;
  }
  atomicMax(&hseq[h], (uint32_t)(e + 1));
}

__global__ void k_colour_write(Bucket* __restrict__ right, Bucket* __restrict__ left, uint64_t cap, uint32_t K,
                               const uint64_t* __restrict__ jk, const int64_t* __restrict__ jc, uint64_t n,
                               const uint64_t* __restrict__ ids, const unsigned long long* __restrict__ hkeys,
                               const uint32_t* __restrict__ hseq, uint64_t hmask) {
  const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= 2 * n) return;
  const uint64_t id = ids[e];
  if (id == kNoId) return;
  uint64_t h = mix64(id) & hmask;
  while (hkeys[h] != id) h = (h + 1) & hmask;
  if (hseq[h] != (uint32_t)(e + 1)) return;   // a later line colours this k-mer
  const uint64_t i = e >> 1;
  const uint64_t km = (e & 1) ? dev_revcomp(jk[i], K) : jk[i];
  const uint16_t c = (uint16_t)(int)jc[i];
  right[id >> 2].jc[id & 3] = c;
  const uint64_t m1 = (1ULL << (2 * (K - 1))) - 1;
  const uint64_t sl = dev_find_slot(left, cap, km & m1);
  const uint32_t fb = (uint32_t)((km >> (2 * (K - 1))) & 3);
  if (sl != ~0ULL && left[sl].cnt[fb] != 0) left[sl].jc[fb] = c;
}

// decolourRepeatsFromDBG (utils.cpp:658-669): the colour of the four homopolymer k-mers becomes 0
__global__ void k_decolour_repeats(Bucket* __restrict__ right, Bucket* __restrict__ left, uint64_t cap, uint32_t K) {
  const uint32_t b = threadIdx.x;
  if (b >= 4) return;
  uint64_t km = 0;
  for (uint32_t i = 0; i < K; ++i) km = (km << 2) | b;
  const uint64_t m1 = (1ULL << (2 * (K - 1))) - 1;
  const uint64_t sr = dev_find_slot(right, cap, km >> 2);
  if (sr != ~0ULL && right[sr].cnt[b] != 0) right[sr].jc[b] = 0;
  const uint64_t sl = dev_find_slot(left, cap, km & m1);
  if (sl != ~0ULL && left[sl].cnt[b] != 0) left[sl].jc[b] = 0;
}

// presence filter (TableView::filter; talc_common.h: blocks by minimizer) from the RIGHT table, one thread per bucket
__global__ void k_build_filter(const Bucket* __restrict__ right, uint64_t cap, uint32_t K, unsigned long long* __restrict__ filter,
                               uint64_t nWords) {
  const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= cap) return;
  const BucketRegs r = load_bucket(right + j);
  if (r.key == kEmptyKey) return;
  const uint64_t key = r.key & kKeyMask;
  const uint64_t nBlocks = nWords >> 3;
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    if (r.cnt[b] == 0) continue;
    const uint64_t km = (key << 2) | (uint64_t)b;
    const FilterHash h = filter_hash(km);
    atomicOr(&filter[filter_index(km, K, h, nBlocks)], (unsigned long long)filter_mask(h));
  }
}

// in-degree of every RIGHT bucket's (K-1)-mer into the top bits of its key word (talc_common.h), one thread per bucket
__global__ void k_build_indegree(Bucket* __restrict__ right, const Bucket* __restrict__ left, uint64_t cap, uint32_t min_count) {
  const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= cap) return;
  unsigned long long TALC_AS1* kp = (unsigned long long TALC_AS1*)&right[j].key;
  const uint64_t raw = *kp;
  if (raw == kEmptyKey) return;
  const uint64_t key = raw & kKeyMask;
  BucketRegs rl;
  uint64_t deg = 0;
  if (probe_bucket(left, cap, key, rl)) deg = (rl.cnt[0] >= min_count) + (rl.cnt[1] >= min_count) + (rl.cnt[2] >= min_count) + (rl.cnt[3] >= min_count);
  *kp = key | (deg << kKeyDegShift);
}

// ------------------------------------------------------------------ walk tables (WalkEntry, talc_common.h)
// One thread per bucket of either table: level 0 from the bucket's own counts, then up to WALK_LEVELS-1 dependent
// probes along the largest-count successor.  Runs once per upload.
TALC_D uint32_t walk_level(const uint32_t c[4], uint32_t min_count, uint32_t& top) {
  uint32_t am = 0;
  top = c[0];
#pragma unroll
  for (uint32_t b = 1; b < 4; ++b) if (c[b] > top) { top = c[b]; am = b; }
  uint32_t nx = 0;
#pragma unroll
  for (uint32_t b = 0; b < 4; ++b) if (b != am && c[b] > nx) nx = c[b];
  const bool fits = top < kWalkTopNone;
  const bool single = fits && top >= min_count && nx < min_count;
  return (fits ? top : kWalkTopNone) | (single ? kWalkSingle : 0u) | (am << kWalkBaseShift);
}

__global__ void k_build_walk(const Bucket* __restrict__ right, const Bucket* __restrict__ left, uint64_t cap, uint32_t K,
                             uint32_t min_count, WalkEntry* __restrict__ walkRight, WalkEntry* __restrict__ walkLeft) {
  const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= 2 * cap) return;
  const bool dirRight = j < cap;
  const uint64_t s = dirRight ? j : j - cap;
  const Bucket* tab = dirRight ? right : left;
  const uint64_t m1 = (1ULL << (2 * (K - 1))) - 1;
  BucketRegs r = load_bucket(tab + s);
  const uint64_t key0 = (r.key == kEmptyKey) ? kEmptyKey : (r.key & kKeyMask);
  uint32_t lv[TALC_WALK_LEVELS];
#pragma unroll
  for (int i = 0; i < TALC_WALK_LEVELS; ++i) lv[i] = 0;
  if (key0 != kEmptyKey) {
    uint64_t key = key0;
#pragma unroll
    for (int lev = 0; lev < TALC_WALK_LEVELS; ++lev) {
      uint32_t top;
      const uint32_t w = walk_level(r.cnt, min_count, top);
      lv[lev] = w;
      if (top == 0 || top >= kWalkTopNone || lev == TALC_WALK_LEVELS - 1) break;   // nothing usable beyond this level
      const uint64_t am = w >> kWalkBaseShift;
      // successor key: RIGHT appends the base to the (K-1)-mer and drops its first base, LEFT prepends and drops the last
      key = dirRight ? (((key << 2) | am) & m1) : ((am << (2 * (K - 2))) | (key >> 2));
      if (!probe_bucket(tab, cap, key, r)) break;   // no such bucket: the remaining levels stay zero
    }
  }
  static_assert(TALC_WALK_LEVELS == 12, "two 16-byte stores: the key and twelve 16-bit levels");
  v4u32 TALC_AS1* out = (v4u32 TALC_AS1*)((dirRight ? walkRight : walkLeft) + s);
  out[0] = v4u32{(uint32_t)key0, (uint32_t)(key0 >> 32), lv[0] | (lv[1] << 16), lv[2] | (lv[3] << 16)};
  out[1] = v4u32{lv[4] | (lv[5] << 16), lv[6] | (lv[7] << 16), lv[8] | (lv[9] << 16), lv[10] | (lv[11] << 16)};
}


// ==================================================================== an imported image's invariant
// "Every stored count is >= MIN_COUNT" is what k_coverage (a hit is a table k-mer) and the walk tables' `single` flag rely
// on, and an image carries no parameters of its own: the importer checks the smallest non-zero count of the RIGHT table and
// the widest key against the parameters it was given (out[0] = that minimum, out[1] = the OR of all keys).
__global__ void k_image_check(const Bucket* __restrict__ right, uint64_t cap, unsigned long long* __restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long mn = ~0ull, orKeys = 0ull;
  if (i < cap && right[i].key != kEmptyKey) {
    orKeys = right[i].key & kKeyMask;
#pragma unroll
    for (int b = 0; b < 4; ++b) { const uint32_t c = right[i].cnt[b]; if (c != 0u && (unsigned long long)c < mn) mn = c; }
  }
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned long long m2 = (unsigned long long)__shfl_down((long long)mn, off, 64), o2 = (unsigned long long)__shfl_down((long long)orKeys, off, 64);
    mn = m2 < mn ? m2 : mn; orKeys |= o2;
  }
  if ((threadIdx.x & 63) == 0) { if (mn != ~0ull) atomicMin(&out[0], mn); if (orKeys) atomicOr(&out[1], orKeys); }
}

// ==================================================================== the text dump parsed on the device
// `jellyfish dump -c` writes one canonical line per k-mer — K letters of ACGT, one blank, one to nine digits, a newline —
// and a 500 M-line dump is 19 GB of them: parsed on the host (128 threads over the mapped file) that took 6.6 s of a
// 12.5 s run, most of it page faults and vector growth, not parsing.  Here the file's bytes go to the GPU as they are
// (several reader threads, page-locked staging buffers) and two kernels turn them into the builder's arrays, line i of the
// file at index i (Jellyfish.cpp:251-269: the first line of a k-mer wins, so the line number is what the builder's
// atomicMin compares): k_parse_count counts the lines that start in each 16 KB tile, the host turns the counts into the
// tiles' first line numbers, k_parse_lines parses every line from its start.  Any line that is not canonical (other white
// space, another length, a letter outside ACGT, no count, a count of ten digits, no newline at the end of the file) raises
// a flag and the whole file goes through the host's tokeniser instead, which is the one that knows what the reference
// does with such lines; the builder kernels apply the MIN_COUNT filter themselves.
constexpr int kParseThreads = 256, kParseSlice = 64, kParseTile = kParseThreads * kParseSlice;
struct ParseStats { unsigned long long kept, flags; };   // flags != 0: a line the device parser does not take

TALC_D bool parse_line_start(const uint8_t* __restrict__ text, uint64_t size, uint64_t p) {
  return p < size && (p == 0 || text[p - 1] == (uint8_t)'\n');
}

__global__ void __launch_bounds__(kParseThreads)
k_parse_count(const uint8_t* __restrict__ text, uint64_t size, uint32_t* __restrict__ tileCount) {
  const uint64_t s0 = (uint64_t)blockIdx.x * kParseTile + (uint64_t)threadIdx.x * kParseSlice;
  uint32_t n = 0;
  if (s0 < size) {
    // newline bytes of [s0 - 1, s0 + 63): a line starts after each of them
    const uint64_t lo = s0 == 0 ? 0 : s0 - 1, hi = min(size, s0 + (uint64_t)kParseSlice) - 1;   // the last byte of the file starts nothing
    if (s0 == 0) n = 1;
    for (uint64_t q = lo; q < hi; ++q) n += text[q] == (uint8_t)'\n' ? 1u : 0u;
  }
  __shared__ uint32_t s_n;
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  for (int off = 32; off > 0; off >>= 1) n += __shfl_down((int)n, off, 64);
  if ((threadIdx.x & 63) == 0 && n) atomicAdd(&s_n, n);
  __syncthreads();
  if (threadIdx.x == 0) tileCount[blockIdx.x] = s_n;
}

__global__ void __launch_bounds__(kParseThreads)
k_parse_lines(const uint8_t* __restrict__ text, uint64_t size, const uint64_t* __restrict__ tileFirstLine, uint32_t K, uint32_t minc,
              uint64_t* __restrict__ kmers, uint32_t* __restrict__ counts, ParseStats* __restrict__ stats) {
  const uint64_t s0 = (uint64_t)blockIdx.x * kParseTile + (uint64_t)threadIdx.x * kParseSlice;
  // the line starts of this thread's slice (a line is at least K + 3 >= 21 bytes: at most four per slice)
  uint64_t st[4];
  uint32_t n = 0;
  bool bad = false;
  if (s0 < size) {
    const uint64_t hi = min(size, s0 + (uint64_t)kParseSlice);
    for (uint64_t q = s0; q < hi; ++q) {
      if (parse_line_start(text, size, q)) { if (n < 4) st[n] = q; else bad = true; ++n; }
    }
  }
  // rank of the thread's first line within the tile: exclusive scan over the block
  __shared__ uint32_t s_scan[kParseThreads];
  s_scan[threadIdx.x] = n;
  __syncthreads();
  for (int off = 1; off < kParseThreads; off <<= 1) {
    const uint32_t add = (threadIdx.x >= (unsigned)off) ? s_scan[threadIdx.x - off] : 0u;
    __syncthreads();
    s_scan[threadIdx.x] += add;
    __syncthreads();
  }
  uint64_t line = tileFirstLine[blockIdx.x] + (s_scan[threadIdx.x] - n);
  uint32_t kept = 0;
  for (uint32_t j = 0; j < min(n, 4u); ++j, ++line) {
    const uint64_t p = st[j];
    uint64_t v = 0;
    uint32_t cv = 0;
    bool ok = p + K + 2 < size;
    if (ok) {
      for (uint32_t i = 0; i < K; ++i) { const uint8_t c = ascii_to_code(text[p + i]); ok &= c < 4; v = (v << 2) | (uint64_t)(c & 3u); }
      ok &= text[p + K] == (uint8_t)' ' || text[p + K] == (uint8_t)'\t';
      uint64_t d = p + K + 1;
      int nd = 0;
      while (d < size && (uint32_t)(text[d] - (uint8_t)'0') < 10u && nd < 9) { cv = cv * 10u + (uint32_t)(text[d] - (uint8_t)'0'); ++d; ++nd; }
      ok &= nd > 0 && d < size && text[d] == (uint8_t)'\n';
    }
    if (ok) { kmers[line] = v; counts[line] = cv; kept += cv >= minc ? 1u : 0u; }
    else bad = true;
  }
  if (bad) atomicOr(&stats->flags, 1ull);
  for (int off = 32; off > 0; off >>= 1) kept += __shfl_down((int)kept, off, 64);
  if ((threadIdx.x & 63) == 0 && kept) atomicAdd(&stats->kept, (unsigned long long)kept);
}

}  // namespace talc
