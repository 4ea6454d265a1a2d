// talc_table_host.h — host-side construction of the successor-grouped k-mer table
// (replaces buildCDBG, Jellyfish.cpp:236-295, and decolourRepeatsFromDBG, utils.cpp:658-669).
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <cstring>
#include <omp.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "talc_common.h"
#include "talc_jf.h"

namespace talc {

struct DeviceCopy {
  Bucket* right = nullptr;
  Bucket* left = nullptr;
  uint64_t* filter = nullptr;
  uint64_t filterWords = 0;
  WalkEntry* walkRight = nullptr;
  WalkEntry* walkLeft = nullptr;
};

struct HostTable {
  talc_params p;
  uint64_t capacity = 0;     // buckets per table
  uint64_t nkmers = 0;       // stored k-mers == SR_DBG.size()
  uint64_t nbuckets_right = 0, nbuckets_left = 0;
  Bucket* right = nullptr;   // host images (calloc'ed; released after the last upload on request)
  Bucket* left = nullptr;
  std::map<int, DeviceCopy> dev;
  bool frozen = false;

  ~HostTable() { free(right); free(left); }

  uint64_t k1mask() const { return (p.k - 1 >= 32) ? ~0ULL : ((1ULL << (2 * (p.k - 1))) - 1); }
  uint64_t kmask() const { return (p.k >= 32) ? ~0ULL : ((1ULL << (2 * p.k)) - 1); }

  static inline uint64_t home(uint64_t key, uint64_t cap) {
    return table_home(key, cap);
  }

  // Buckets per table for nKept stored k-mers.  Load factor <= 0.5 always; a table built for a GPU whose memory it leaves
  // mostly idle is made sparser — 4 or 3 buckets per k-mer — because what a lookup costs is the number of lines its
  // linear-probing chain touches, and the random-read rate of the memory system is what k_coverage is bound by (a 217 M
  // k-mer table: 38.3 % of the HBM peak at 2 buckets per k-mer, 39.6 % at 2.7, 42.2 % at 4; k_search -1.7 %;
  // profiles/r04).  deviceBytes = 0 (a host table): 2.  The four tables of a copy (two bucket tables, two walk tables:
  // 128 bytes per bucket) may take 40 % of the device at the density chosen.  TALC_TABLE_SLOTS_X10 overrides (20 .. 80).
  static uint64_t capacity_for(uint64_t nKept, uint64_t deviceBytes = 0) {
    uint64_t x10 = 20;
    for (uint64_t cand : {40ull, 30ull}) {
      if (deviceBytes && (nKept * cand / 10) * 128ull <= deviceBytes / 10 * 4) { x10 = cand; break; }
    }
    if (const char* e = getenv("TALC_TABLE_SLOTS_X10")) { const uint64_t v = strtoull(e, nullptr, 10); if (v >= 20 && v <= 80) x10 = v; }
    uint64_t cap = nKept * x10 / 10 + 64;
    if (cap >= (1ULL << 32)) cap = nKept * 2 + 64;   // (table_slot() addresses 2^32 buckets per table)
    return cap;
  }
  bool allocate(uint64_t nKept, bool clear = true) {
    capacity = capacity_for(nKept);
    if (capacity >= (1ULL << 32)) return false;   // table_slot() addresses 2^32 buckets (137 GB) per table
    right = (Bucket*)malloc(capacity * sizeof(Bucket));
    left = (Bucket*)malloc(capacity * sizeof(Bucket));
    if (!right || !left) return false;
    if (!clear) return true;   // (the device builder overwrites both images)
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)capacity; ++i) {
      right[i].key = kEmptyKey; left[i].key = kEmptyKey;
      for (int b = 0; b < 4; ++b) { right[i].cnt[b] = 0; right[i].jc[b] = 0; left[i].cnt[b] = 0; left[i].jc[b] = 0; }
    }
    return true;
  }

  // find the bucket of `key` (or the empty bucket where it would go).  bounded: stay inside
  // [.., limit) — used by the parallel phase, where a thread must never touch another thread's
  // bucket range — and report false if the probe sequence would leave it.
  static inline bool findSlotBounded(Bucket* tab, uint64_t cap, uint64_t key, uint64_t limit, uint64_t& slot) {
    uint64_t i = home(key, cap);
    while (i < limit) {
      if ((tab[i].key & kKeyMask) == key || tab[i].key == kEmptyKey) { slot = i; return true; }
      ++i;
    }
    return false;
  }
  static inline void findSlot(Bucket* tab, uint64_t cap, uint64_t key, uint64_t& slot) {
    uint64_t i = home(key, cap);
    while (true) {
      if ((tab[i].key & kKeyMask) == key || tab[i].key == kEmptyKey) { slot = i; return; }
      if (++i == cap) i = 0;
    }
  }
  static inline const Bucket* find(const Bucket* tab, uint64_t cap, uint64_t key) {
    uint64_t i = home(key, cap);
    while (true) {
      if ((tab[i].key & kKeyMask) == key) return &tab[i];   // (an image that came back from a device carries degree bits)
      if (tab[i].key == kEmptyKey) return nullptr;
      if (++i == cap) i = 0;
    }
  }

  // Insert the (already count-filtered) k-mers; first duplicate wins (std::map::insert,
  // Jellyfish.cpp:262).  Parallel over hash ranges: each thread owns a contiguous bucket range
  // and takes the keys whose home bucket falls in it, in array order, so the winner among
  // duplicates is the same as in a serial pass; keys whose probe sequence would cross the
  // range end are deferred to a serial pass.
  void insertAll(const uint64_t* kmers, const uint32_t* counts, uint64_t n) {
    const uint64_t m1 = k1mask();
    const uint32_t minc = p.min_count;
    for (int which = 0; which < 2; ++which) {
      Bucket* tab = which == 0 ? right : left;
      int T = omp_get_max_threads();
      if (T > 16) T = 16;   // every thread scans the whole array: more threads only add redundant scanning
      if (n < 100000) T = 1;
      std::vector<std::vector<uint64_t>> deferred(T);
      std::vector<uint64_t> added(T, 0), newb(T, 0);
#pragma omp parallel num_threads(T)
      {
        const int t = omp_get_thread_num();
        const uint64_t lo = capacity / T * t, hi = (t == T - 1) ? capacity : capacity / T * (t + 1);
        uint64_t nadd = 0, nb = 0;
        for (uint64_t i = 0; i < n; ++i) {
          if (counts[i] < minc) continue;
          const uint64_t km = kmers[i];
          const uint64_t key = which == 0 ? (km >> 2) : (km & m1);
          const uint64_t h = home(key, capacity);
          if (h < lo || h >= hi) continue;
          const uint32_t b = which == 0 ? (uint32_t)(km & 3) : (uint32_t)((km >> (2 * (p.k - 1))) & 3);
          uint64_t slot;
          if (!findSlotBounded(tab, capacity, key, hi, slot)) { deferred[t].push_back(i); continue; }
          if (tab[slot].key == kEmptyKey) { tab[slot].key = key; nb++; }
          if (tab[slot].cnt[b] == 0) { tab[slot].cnt[b] = counts[i]; nadd++; }
        }
        added[t] = nadd; newb[t] = nb;
      }
      uint64_t nadd = 0, nb = 0;
      for (int t = 0; t < T; ++t) { nadd += added[t]; nb += newb[t]; }
      for (int t = 0; t < T; ++t)
        for (uint64_t i : deferred[t]) {
          const uint64_t km = kmers[i];
          const uint64_t key = which == 0 ? (km >> 2) : (km & m1);
          const uint32_t b = which == 0 ? (uint32_t)(km & 3) : (uint32_t)((km >> (2 * (p.k - 1))) & 3);
          uint64_t slot;
          findSlot(tab, capacity, key, slot);
          if (tab[slot].key == kEmptyKey) { tab[slot].key = key; nb++; }
          if (tab[slot].cnt[b] == 0) { tab[slot].cnt[b] = counts[i]; nadd++; }
        }
      if (which == 0) { nkmers = nadd; nbuckets_right = nb; } else nbuckets_left = nb;
    }
  }

  // (count, colour) of a full k-mer, host side
  bool lookup(uint64_t km, uint32_t& cnt, uint32_t& jc) const {
    const Bucket* b = find(right, capacity, km >> 2);
    cnt = 0; jc = 0;
    if (!b) return false;
    cnt = b->cnt[km & 3]; jc = b->jc[km & 3];
    return cnt != 0;
  }
  void setColour(uint64_t km, uint16_t c) {
    Bucket* b = const_cast<Bucket*>(find(right, capacity, km >> 2));
    if (b && b->cnt[km & 3] != 0) b->jc[km & 3] = c;
    Bucket* l = const_cast<Bucket*>(find(left, capacity, km & k1mask()));
    const uint32_t fb = (uint32_t)((km >> (2 * (p.k - 1))) & 3);
    if (l && l->cnt[fb] != 0) l->jc[fb] = c;
  }
  uint64_t revcomp(uint64_t km) const {
    uint64_t r = 0;
    for (uint32_t i = 0; i < p.k; ++i) { r = (r << 2) | (3 - (km & 3)); km >>= 2; }
    return r;
  }
  // Jellyfish.cpp:278-289 in list order: the k-mer, then its reverse complement
  void colour(const uint64_t* jkmers, const int64_t* jcounts, uint64_t n) {
    for (uint64_t i = 0; i < n; ++i) {
      const int jc = (int)jcounts[i];
      if (!((unsigned int)jc < p.coloured_count_thr)) continue;  // int vs unsigned compare of the reference
      uint32_t c, j;
      if (lookup(jkmers[i], c, j)) setColour(jkmers[i], (uint16_t)jc);
      const uint64_t rc = revcomp(jkmers[i]);
      if (lookup(rc, c, j)) setColour(rc, (uint16_t)jc);
    }
  }
  // utils.cpp:658-669
  void decolourRepeats() {
    for (uint64_t b = 0; b < 4; ++b) {
      uint64_t km = 0;
      for (uint32_t i = 0; i < p.k; ++i) km = (km << 2) | b;
      uint32_t c, j;
      if (lookup(km, c, j)) setColour(km, 0);
    }
  }
};

// text -> packed; false if not exactly K letters of ACGT (any case)
static inline bool packText(const char* s, size_t len, uint32_t K, uint64_t& out) {
  if (len != K) return false;
  uint64_t v = 0;
  for (size_t i = 0; i < len; ++i) {
    uint8_t c = ascii_to_code((uint8_t)s[i]);
    if (c > 3) return false;
    v = (v << 2) | c;
  }
  out = v;
  return true;
}

// "kmer count" lines (Jellyfish.cpp:251-269: whitespace-separated tokens, count by std::stoi, kept if
// (unsigned)count >= min_count, Jellyfish.cpp:260) from a memory-mapped file, parsed in parallel over chunks cut at
// line ends; the kept entries come out in file order (first duplicate wins later on).  Lines without two tokens are
// counted as bad; kept k-mers that are not K letters of ACGT are not representable and dropped.
// wantCounts false: the second token is returned as a signed count for every line (junction dumps).
struct DumpStats { int64_t nread = 0, nkept = 0, nbad = 0; };
// A file that starts with a Jellyfish 2 header is read as that tool's binary count file (talc_jf.h) under the same
// contract; what does not verify there returns false with *err set (a file that cannot be opened leaves *err empty).
static inline bool parseDumpFile(const char* path, uint32_t K, uint32_t minc, bool filter, std::vector<uint64_t>& kmers,
                                 std::vector<uint32_t>* counts, std::vector<int64_t>* scounts, DumpStats& st,
                                 std::string* err = nullptr) {
  const int fd = open(path, O_RDONLY);
  if (fd < 0) return false;
  struct stat sb;
  if (fstat(fd, &sb) != 0) { close(fd); return false; }
  const size_t size = (size_t)sb.st_size;
  if (size == 0) { close(fd); return true; }
  const char* base = (const char*)mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
  close(fd);
  if (base == MAP_FAILED) return false;
  if (jfLooksLike(base, size)) {
    JfStats js;
    std::string why;
    const bool ok = jfParseImage(base, size, K, minc, filter, kmers, counts, scounts, js, why);
    munmap((void*)base, size);
    if (!ok && err) *err = std::string(path) + ": " + why;
    st.nread += js.nread; st.nkept += js.nkept;
    return ok;
  }
  int T = omp_get_max_threads();
  if (T > 128) T = 128;
  if (size < (1u << 20)) T = 1;
  std::vector<size_t> cut(T + 1, size);
  cut[0] = 0;
  for (int t = 1; t < T; ++t) {
    size_t p = size / T * t;
    while (p < size && base[p - 1] != '\n') ++p;
    cut[t] = p;
  }
  std::vector<std::vector<uint64_t>> lk(T);
  std::vector<std::vector<uint32_t>> lc(T);
  std::vector<std::vector<int64_t>> ls(T);
  std::vector<DumpStats> lst(T);
#pragma omp parallel num_threads(T)
  {
    const int t = omp_get_thread_num();
    const char* p = base + cut[t];
    const char* end = base + cut[t + 1];
    auto isws = [](char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\v' || c == '\f'; };
    DumpStats s;
    {   // the chunk's line count bounds its entries: one allocation per thread instead of a dozen doublings with copies
      size_t lines = 1;
      for (const char* q = p; q < end;) { const char* nl = (const char*)memchr(q, '\n', (size_t)(end - q)); if (!nl) break; ++lines; q = nl + 1; }
      lk[t].reserve(lines);
      if (counts) lc[t].reserve(lines);
      if (scounts) ls[t].reserve(lines);
    }
    // code of a letter (0..3) or 0x80; the fast path below takes the canonical line of `jellyfish dump -c` — exactly K
    // letters of ACGT, one blank, one to nine digits, newline — in one forward pass; anything else goes through the
    // tokeniser, which gives such a line the same answer
    uint8_t lut[256];
    for (int i = 0; i < 256; ++i) { const uint8_t c = ascii_to_code((uint8_t)i); lut[i] = c > 3 ? 0x80 : c; }
    while (p < end) {
      if (p + K + 2 < end) {
        uint64_t v = 0;
        uint32_t bad = 0;
        for (uint32_t i = 0; i < K; ++i) { const uint32_t c = lut[(uint8_t)p[i]]; bad |= c; v = (v << 2) | (c & 3u); }
        if (!(bad & 0x80u) && (p[K] == ' ' || p[K] == '\t')) {
          const char* d = p + K + 1;
          uint32_t cv = 0; int nd = 0;
          while (d < end && (unsigned)(*d - '0') < 10u && nd < 9) { cv = cv * 10u + (uint32_t)(*d - '0'); ++d; ++nd; }
          if (nd > 0 && d < end && *d == '\n') {
            s.nread++;
            if (!filter || cv >= minc) {
              if (filter) s.nkept++;
              lk[t].push_back(v);
              if (counts) lc[t].push_back(cv);
              if (scounts) ls[t].push_back((int64_t)cv);
            }
            p = d + 1;
            continue;
          }
        }
      }
      const char* eol = (const char*)memchr(p, '\n', (size_t)(end - p));
      if (!eol) eol = end;
      const char* q = p;
      while (q < eol && isws(*q)) ++q;
      const char* k0 = q;
      while (q < eol && !isws(*q)) ++q;
      const char* k1 = q;
      while (q < eol && isws(*q)) ++q;
      const char* c0 = q;
      while (q < eol && !isws(*q)) ++q;
      const char* c1 = q;
      if (k1 > k0 && c1 > c0) {
        // atoi: optional sign, then digits; anything else stops the number (0 if none)
        long long v = 0; bool neg = false; const char* d = c0;
        if (d < c1 && (*d == '+' || *d == '-')) { neg = (*d == '-'); ++d; }
        while (d < c1 && *d >= '0' && *d <= '9') { v = v * 10 + (*d - '0'); if (v > 0x7fffffffLL) v = 0x7fffffffLL; ++d; }
        const int c = (int)(neg ? -v : v);
        s.nread++;
        if (!filter || (unsigned int)c >= minc) {
          if (filter) s.nkept++;
          uint64_t packed;
          if (packText(k0, (size_t)(k1 - k0), K, packed)) {
            lk[t].push_back(packed);
            if (counts) lc[t].push_back((uint32_t)c);
            if (scounts) ls[t].push_back((int64_t)c);
          }
        }
      } else {
        s.nbad++;   // (fgets + sscanf counted every line that did not yield two tokens, blank ones included)
      }
      p = eol + 1;
    }
    lst[t] = s;
  }
  munmap((void*)base, size);
  size_t total = 0;
  for (int t = 0; t < T; ++t) total += lk[t].size();
  kmers.resize(total);
  if (counts) counts->resize(total);
  if (scounts) scounts->resize(total);
  std::vector<size_t> offs(T + 1, 0);
  for (int t = 0; t < T; ++t) {
    offs[t + 1] = offs[t] + lk[t].size();
    st.nread += lst[t].nread; st.nkept += lst[t].nkept; st.nbad += lst[t].nbad;
  }
#pragma omp parallel for num_threads(T) schedule(static, 1)
  for (int t = 0; t < T; ++t) {   // file order is kept: chunk t's entries follow chunk t-1's
    if (lk[t].empty()) continue;
    memcpy(kmers.data() + offs[t], lk[t].data(), lk[t].size() * 8);
    if (counts) memcpy(counts->data() + offs[t], lc[t].data(), lc[t].size() * 4);
    if (scounts) memcpy(scounts->data() + offs[t], ls[t].data(), ls[t].size() * 8);
    std::vector<uint64_t>().swap(lk[t]);
  }
  return true;
}

}  // namespace talc
