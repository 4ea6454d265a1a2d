// synth.cpp — deterministic synthetic inputs for the TALC hot path.
//
// Host-only helper library (libtalc_synth.so): a random transcriptome, the matching
// `jellyfish dump -c` style k-mer table (as packed arrays or as text), an optional junction
// k-mer dump, and ONT-like long reads with substitutions / insertions / deletions.  Every read
// is generated from (seed, read index) alone, so any rank can produce its own shard.
// All randomness is splitmix64 / xoshiro256** seeded as stated; no external data.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace {

struct Rng {
  uint64_t s[4];
  static uint64_t splitmix(uint64_t& x) {
    uint64_t z = (x += 0x9e3779b97f4a7c15ULL);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
  }
  explicit Rng(uint64_t seed) { for (int i = 0; i < 4; ++i) s[i] = splitmix(seed); }
  static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
  uint64_t next() {  // xoshiro256**
    const uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
    return r;
  }
  double uniform() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
  uint64_t below(uint64_t n) { return n ? next() % n : 0; }
  double normal() {
    double u1 = uniform(), u2 = uniform();
    if (u1 < 1e-300) u1 = 1e-300;
    return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
  }
  uint32_t poisson(double lambda) {
    if (lambda < 30.0) {
      const double L = std::exp(-lambda);
      uint32_t k = 0; double p = 1.0;
      do { ++k; p *= uniform(); } while (p > L);
      return k - 1;
    }
    double v = lambda + std::sqrt(lambda) * normal();
    return v < 0 ? 0u : (uint32_t)(v + 0.5);
  }
};

struct Spec {
  uint64_t target_kmers;   // total transcript bases (~ number of distinct k-mers)
  uint32_t k;
  uint32_t mean_len, sd_len, min_len;   // read length ~ N(mean, sd) clipped [min_len, transcript]
  int32_t mixed_lengths;   // config 5: log-uniform 500..20000
  double sub_rate, ins_rate, del_rate;
  double frac_short, frac_random;
  double extra_error_frac, count1_frac;
  int32_t junction_period; // ~300
  uint64_t seed;
  // paralog families (off by default, so every other field keeps producing the data it always did): with
  // probability paralog_frac a new transcript is a mutated copy of an earlier one (substitutions at rate
  // paralog_div, insertions and deletions at a quarter of it each), which puts bubbles and forks into the
  // de Bruijn graph: several live Trails, gardening, bridge scoring
  double paralog_frac, paralog_div;
};

struct Synth {
  Spec sp;
  std::vector<uint8_t> tx;            // concatenated transcripts, codes 0..3
  std::vector<uint64_t> tstart;       // transcript start offsets (+ sentinel)
  std::vector<double> tlambda;
  std::vector<double> tcum;           // cumulative weight length*lambda
  // mixed_lengths (config 5): transcripts by ascending length and the cumulative weights in that order, so that a
  // read of length L is drawn (again with probability ~ length*lambda) among the transcripts that can hold it
  std::vector<uint32_t> byLen;
  std::vector<uint64_t> lenSorted;
  std::vector<double> cumSorted;
  std::vector<uint64_t> dkeys;
  std::vector<uint32_t> dcounts;
  std::vector<uint64_t> jkeys;
  std::vector<int64_t> jcounts;
  bool dumpBuilt = false, juncBuilt = false;
};

const char D[4] = {'A', 'C', 'G', 'T'};

void buildTranscriptome(Synth& S) {
  Rng rng(S.sp.seed ^ 0x7A1C0001ULL);
  Rng prng(S.sp.seed ^ 0x7A1C0009ULL);   // separate stream: the default (no paralogs) consumes none of it
  uint64_t total = 0;
  S.tstart.push_back(0);
  while (total < S.sp.target_kmers) {
    // config 5 (mixed read lengths up to 20 kb) needs transcripts that long: a wider length distribution there
    double len = S.sp.mixed_lengths ? std::exp(std::log(4000.0) + 1.0 * rng.normal()) : std::exp(std::log(1500.0) + 0.6 * rng.normal());
    if (len < 300) len = 300;
    if (len > 30000) len = 30000;
    uint64_t L = (uint64_t)len;
    if (total + L > S.sp.target_kmers && S.sp.target_kmers - total >= 300) L = S.sp.target_kmers - total;
    const size_t nHave = S.tstart.size() - 1;
    if (S.sp.paralog_frac > 0 && nHave > 0 && prng.uniform() < S.sp.paralog_frac) {
      const size_t u = (size_t)prng.below(nHave);
      const uint64_t b = S.tstart[u], e = S.tstart[u + 1];
      const double dv = S.sp.paralog_div;
      uint64_t made = 0;
      for (uint64_t i = b; i < e; ++i) {
        const double r = prng.uniform();
        if (r < dv) { S.tx.push_back((uint8_t)((S.tx[i] + 1 + prng.below(3)) & 3)); ++made; }
        else if (r < 1.25 * dv) { /* deletion */ }
        else if (r < 1.5 * dv) { S.tx.push_back((uint8_t)(prng.next() >> 62)); S.tx.push_back(S.tx[i]); made += 2; }
        else { S.tx.push_back(S.tx[i]); ++made; }
      }
      while (made < 300) { S.tx.push_back((uint8_t)(prng.next() >> 62)); ++made; }
      L = made;
    } else {
      for (uint64_t i = 0; i < L; ++i) S.tx.push_back((uint8_t)(rng.next() >> 62));
    }
    total += L;
    S.tstart.push_back(total);
    double lam = std::exp(std::log(30.0) + 1.0 * rng.normal());
    if (lam < 4) lam = 4;
    if (lam > 5000) lam = 5000;
    S.tlambda.push_back(lam);
  }
  double c = 0;
  for (size_t t = 0; t < S.tlambda.size(); ++t) {
    c += (double)(S.tstart[t + 1] - S.tstart[t]) * S.tlambda[t];
    S.tcum.push_back(c);
  }
  if (S.sp.mixed_lengths) {
    const size_t n = S.tlambda.size();
    S.byLen.resize(n);
    for (size_t t = 0; t < n; ++t) S.byLen[t] = (uint32_t)t;
    std::stable_sort(S.byLen.begin(), S.byLen.end(), [&](uint32_t a, uint32_t b) {
      return S.tstart[a + 1] - S.tstart[a] < S.tstart[b + 1] - S.tstart[b];
    });
    double cs = 0;
    for (size_t i = 0; i < n; ++i) {
      const uint32_t t = S.byLen[i];
      const uint64_t len = S.tstart[t + 1] - S.tstart[t];
      cs += (double)len * S.tlambda[t];
      S.lenSorted.push_back(len);
      S.cumSorted.push_back(cs);
    }
  }
}

// gcd for the affine line permutation below
uint64_t gcd64(uint64_t a, uint64_t b) { while (b) { const uint64_t t = a % b; a = b; b = t; } return a; }

// The same kind of dump for the big configurations (>= 100 M k-mers: BASELINE configs 3-5), built in parallel: one
// generator per transcript / per block of extra lines instead of one stream for the whole file, and the line order
// scattered by an affine permutation of the line index instead of a serial Fisher-Yates.  Deterministic for a given
// spec whatever the number of threads.  (Smaller configurations keep the serial form below: the committed golden
// fixtures pin its exact output.)
void buildDumpParallel(Synth& S) {
  const uint32_t K = S.sp.k;
  const uint64_t mask = (K >= 32) ? ~0ULL : ((1ULL << (2 * K)) - 1);
  const size_t nT = S.tlambda.size();
  std::vector<uint64_t> off(nT + 1, 0);
  for (size_t t = 0; t < nT; ++t) {
    const uint64_t len = S.tstart[t + 1] - S.tstart[t];
    off[t + 1] = off[t] + (len >= K ? len - K + 1 : 0);
  }
  const uint64_t nTrue = off[nT];
  const uint64_t nErr = (uint64_t)(S.sp.extra_error_frac * (double)nTrue);
  const uint64_t nOne = (uint64_t)(S.sp.count1_frac * (double)nTrue);
  const uint64_t N = nTrue + nErr + nOne;
  std::vector<uint64_t> keys(N);
  std::vector<uint32_t> counts(N);
#pragma omp parallel for schedule(dynamic, 64)
  for (long t = 0; t < (long)nT; ++t) {
    const uint64_t b = S.tstart[t], e = S.tstart[t + 1];
    if (e - b < K) continue;
    Rng rng((S.sp.seed ^ 0x7A1C0002ULL) + 0x9E3779B97F4A7C15ULL * (uint64_t)(t + 1));
    uint64_t km = 0, w = off[t];
    for (uint64_t i = b; i < e; ++i) {
      km = ((km << 2) | S.tx[i]) & mask;
      if (i - b + 1 >= K) {
        uint32_t c = rng.poisson(S.tlambda[t]);
        if (c < 2) c = 2;
        keys[w] = km; counts[w] = c; ++w;
      }
    }
  }
  const uint64_t BLK = 1u << 16;
  const uint64_t nExtra = nErr + nOne;
#pragma omp parallel for schedule(dynamic, 4)
  for (long blk = 0; blk < (long)((nExtra + BLK - 1) / BLK); ++blk) {
    Rng rng((S.sp.seed ^ 0x7A1C0012ULL) + 0x9E3779B97F4A7C15ULL * (uint64_t)(blk + 1));
    const uint64_t lo = (uint64_t)blk * BLK, hi = std::min(nExtra, lo + BLK);
    for (uint64_t i = lo; i < hi; ++i) {
      if (i < nErr) {   // one substitution of a true k-mer, count in {2,3}
        uint64_t km = keys[rng.below(nTrue)];
        const uint32_t o = (uint32_t)rng.below(K);
        const uint64_t cur = (km >> (2 * o)) & 3, nb = (cur + 1 + rng.below(3)) & 3;
        km = (km & ~(3ULL << (2 * o))) | (nb << (2 * o));
        keys[nTrue + i] = km; counts[nTrue + i] = 2 + (uint32_t)rng.below(2);
      } else {          // count 1: dropped by MIN_COUNT
        keys[nTrue + i] = rng.next() & mask; counts[nTrue + i] = 1;
      }
    }
  }
  // line i of the dump = entry (a i + b) mod N, a coprime to N
  uint64_t a = 0x9E3779B97F4A7C15ULL % N, b0 = (S.sp.seed * 0xD1B54A32D192ED03ULL + 12345) % N;
  if (a < 2) a = 2;
  while (gcd64(a, N) != 1) ++a;
  S.dkeys.resize(N); S.dcounts.resize(N);
#pragma omp parallel for schedule(static)
  for (long i = 0; i < (long)N; ++i) {
    const uint64_t j = (uint64_t)(((unsigned __int128)a * (uint64_t)i + b0) % N);
    S.dkeys[i] = keys[j]; S.dcounts[i] = counts[j];
  }
  S.dumpBuilt = true;
}

void buildDump(Synth& S) {
  if (S.dumpBuilt) return;
  if (S.sp.target_kmers >= 100000000ULL) { buildDumpParallel(S); return; }
  const uint32_t K = S.sp.k;
  const uint64_t mask = (K >= 32) ? ~0ULL : ((1ULL << (2 * K)) - 1);
  Rng rng(S.sp.seed ^ 0x7A1C0002ULL);
  std::vector<uint64_t>& keys = S.dkeys;
  std::vector<uint32_t>& counts = S.dcounts;
  for (size_t t = 0; t + 1 < S.tstart.size(); ++t) {
    const uint64_t b = S.tstart[t], e = S.tstart[t + 1];
    if (e - b < K) continue;
    uint64_t km = 0;
    for (uint64_t i = b; i < e; ++i) {
      km = ((km << 2) | S.tx[i]) & mask;
      if (i - b + 1 >= K) {
        uint32_t c = rng.poisson(S.tlambda[t]);
        if (c < 2) c = 2;
        keys.push_back(km);
        counts.push_back(c);
      }
    }
  }
  const uint64_t nTrue = keys.size();
  // "error" k-mers: one substitution of a true k-mer, count in {2,3}
  const uint64_t nErr = (uint64_t)(S.sp.extra_error_frac * (double)nTrue);
  for (uint64_t i = 0; i < nErr; ++i) {
    uint64_t km = keys[rng.below(nTrue)];
    uint32_t off = (uint32_t)rng.below(K);
    uint64_t cur = (km >> (2 * off)) & 3, nb = (cur + 1 + rng.below(3)) & 3;
    km = (km & ~(3ULL << (2 * off))) | (nb << (2 * off));
    keys.push_back(km);
    counts.push_back(2 + (uint32_t)rng.below(2));
  }
  // lines with count 1 (must be dropped by MIN_COUNT): random k-mers
  const uint64_t nOne = (uint64_t)(S.sp.count1_frac * (double)nTrue);
  for (uint64_t i = 0; i < nOne; ++i) {
    keys.push_back(rng.next() & mask);
    counts.push_back(1);
  }
  // shuffle line order (Fisher-Yates)
  for (uint64_t i = keys.size(); i > 1; --i) {
    uint64_t j = rng.below(i);
    std::swap(keys[i - 1], keys[j]);
    std::swap(counts[i - 1], counts[j]);
  }
  S.dumpBuilt = true;
}

void buildJunctions(Synth& S) {
  if (S.juncBuilt) return;
  const uint32_t K = S.sp.k;
  const uint64_t mask = (K >= 32) ? ~0ULL : ((1ULL << (2 * K)) - 1);
  Rng rng(S.sp.seed ^ 0x7A1C0004ULL);
  const int period = S.sp.junction_period > 0 ? S.sp.junction_period : 300;
  for (size_t t = 0; t + 1 < S.tstart.size(); ++t) {
    const uint64_t b = S.tstart[t], e = S.tstart[t + 1];
    uint64_t pos = b + period / 2 + rng.below(period);
    while (pos + K < e && pos >= b + K) {
      // the K-1 k-mers spanning the junction between pos-1 and pos
      const bool big = rng.uniform() < 0.05;
      for (uint64_t s = pos - K + 1; s < pos; ++s) {
        uint64_t km = 0;
        for (uint32_t i = 0; i < K; ++i) km = ((km << 2) | S.tx[s + i]) & mask;
        S.jkeys.push_back(km);
        S.jcounts.push_back(big ? 10000 + (int64_t)rng.below(5000) : 2 + (int64_t)rng.below(200));
      }
      pos += period / 2 + rng.below(period);
    }
  }
  S.juncBuilt = true;
}

// one read from (seed, index): appended to out as ASCII
void makeRead(const Synth& S, uint64_t index, std::string& out) {
  out.clear();
  Rng rng((S.sp.seed ^ 0x7A1C0003ULL) + 0x9E3779B97F4A7C15ULL * (index + 1));
  const uint32_t K = S.sp.k;
  const double u = rng.uniform();
  if (u < S.sp.frac_short) {  // shorter than K+1
    uint32_t L = 1 + (uint32_t)rng.below(K);
    for (uint32_t i = 0; i < L; ++i) out.push_back(D[rng.next() >> 62]);
    return;
  }
  uint64_t L;
  if (S.sp.mixed_lengths) L = (uint64_t)std::exp(std::log(500.0) + rng.uniform() * (std::log(20000.0) - std::log(500.0)));
  else {
    double l = (double)S.sp.mean_len + (double)S.sp.sd_len * rng.normal();
    if (l < S.sp.min_len) l = S.sp.min_len;
    L = (uint64_t)l;
  }
  if (u < S.sp.frac_short + S.sp.frac_random) {  // pure random read
    for (uint64_t i = 0; i < L; ++i) out.push_back(D[rng.next() >> 62]);
    return;
  }
  // transcript chosen with probability ~ length * lambda (mixed lengths: among those at least L long, so that the
  // read-length distribution really reaches 20 kb instead of being clipped to short transcripts)
  size_t t;
  if (S.sp.mixed_lengths) {
    if (L > S.lenSorted.back()) L = S.lenSorted.back();
    const size_t i0 = (size_t)(std::lower_bound(S.lenSorted.begin(), S.lenSorted.end(), L) - S.lenSorted.begin());
    const double base = i0 ? S.cumSorted[i0 - 1] : 0.0;
    const double w = base + rng.uniform() * (S.cumSorted.back() - base);
    size_t i = (size_t)(std::lower_bound(S.cumSorted.begin() + (long)i0, S.cumSorted.end(), w) - S.cumSorted.begin());
    if (i >= S.byLen.size()) i = S.byLen.size() - 1;
    t = S.byLen[i];
  } else {
    const double w = rng.uniform() * S.tcum.back();
    t = (size_t)(std::lower_bound(S.tcum.begin(), S.tcum.end(), w) - S.tcum.begin());
    if (t >= S.tlambda.size()) t = S.tlambda.size() - 1;
  }
  const uint64_t b = S.tstart[t], e = S.tstart[t + 1];
  if (L > e - b) L = e - b;
  const uint64_t start = b + rng.below(e - b - L + 1);
  const double ps = S.sp.sub_rate, pi = S.sp.ins_rate, pd = S.sp.del_rate;
  for (uint64_t i = start; i < start + L; ++i) {
    const double r = rng.uniform();
    if (r < pd) continue;                                   // deletion
    if (r < pd + pi) out.push_back(D[rng.next() >> 62]);    // insertion before the base
    uint8_t c = S.tx[i];
    if (r >= pd + pi && r < pd + pi + ps) c = (uint8_t)((c + 1 + rng.below(3)) & 3);  // substitution
    out.push_back(D[c]);
  }
}

std::string unpack(uint64_t km, uint32_t K) {
  std::string s(K, 'A');
  for (uint32_t i = 0; i < K; ++i) s[K - 1 - i] = D[(km >> (2 * i)) & 3];
  return s;
}

}  // namespace

extern "C" {

struct synth_spec {
  uint64_t target_kmers;
  uint32_t k;
  uint32_t mean_len, sd_len, min_len;
  int32_t mixed_lengths;
  double sub_rate, ins_rate, del_rate;
  double frac_short, frac_random;
  double extra_error_frac, count1_frac;
  int32_t junction_period;
  uint64_t seed;
  double paralog_frac, paralog_div;
};

void synth_spec_default(synth_spec* s) {
  s->target_kmers = 5000000; s->k = 21; s->mean_len = 2000; s->sd_len = 400; s->min_len = 500;
  s->mixed_lengths = 0; s->sub_rate = 0.04; s->ins_rate = 0.04; s->del_rate = 0.04;
  s->frac_short = 0.001; s->frac_random = 0.005; s->extra_error_frac = 0.10; s->count1_frac = 0.02;
  s->junction_period = 300; s->seed = 0; s->paralog_frac = 0.0; s->paralog_div = 0.03;
}

void* synth_create(const synth_spec* sp) {
  Synth* S = new Synth();
  S->sp.target_kmers = sp->target_kmers; S->sp.k = sp->k; S->sp.mean_len = sp->mean_len; S->sp.sd_len = sp->sd_len;
  S->sp.min_len = sp->min_len; S->sp.mixed_lengths = sp->mixed_lengths; S->sp.sub_rate = sp->sub_rate;
  S->sp.ins_rate = sp->ins_rate; S->sp.del_rate = sp->del_rate; S->sp.frac_short = sp->frac_short;
  S->sp.frac_random = sp->frac_random; S->sp.extra_error_frac = sp->extra_error_frac;
  S->sp.count1_frac = sp->count1_frac; S->sp.junction_period = sp->junction_period; S->sp.seed = sp->seed;
  S->sp.paralog_frac = sp->paralog_frac; S->sp.paralog_div = sp->paralog_div;
  buildTranscriptome(*S);
  return S;
}
void synth_destroy(void* h) { delete (Synth*)h; }

uint64_t synth_transcriptome_bases(void* h) { return ((Synth*)h)->tx.size(); }
uint64_t synth_num_transcripts(void* h) { return ((Synth*)h)->tlambda.size(); }

uint64_t synth_dump_size(void* h) { buildDump(*(Synth*)h); return ((Synth*)h)->dkeys.size(); }
void synth_dump_arrays(void* h, uint64_t* keys, uint32_t* counts) {
  Synth& S = *(Synth*)h; buildDump(S);
  memcpy(keys, S.dkeys.data(), S.dkeys.size() * 8);
  memcpy(counts, S.dcounts.data(), S.dcounts.size() * 4);
}
// free the internal copy once the caller has taken the arrays (big configs)
void synth_dump_release(void* h) {
  Synth& S = *(Synth*)h;
  std::vector<uint64_t>().swap(S.dkeys); std::vector<uint32_t>().swap(S.dcounts); S.dumpBuilt = false;
}
uint64_t synth_junction_size(void* h) { buildJunctions(*(Synth*)h); return ((Synth*)h)->jkeys.size(); }
void synth_junction_arrays(void* h, uint64_t* keys, int64_t* jcounts) {
  Synth& S = *(Synth*)h; buildJunctions(S);
  memcpy(keys, S.jkeys.data(), S.jkeys.size() * 8);
  memcpy(jcounts, S.jcounts.data(), S.jcounts.size() * 8);
}

// reads [first, first+n): pass bases=NULL to get the total size in offsets[n].  Every read depends on (seed, index)
// only, so large requests are generated in parallel.
void synth_reads(void* h, uint64_t first, uint32_t n, char* bases, uint64_t* offsets) {
  const Synth& S = *(Synth*)h;
  if (n < 4096) {
    std::string r;
    uint64_t pos = 0;
    for (uint32_t i = 0; i < n; ++i) {
      makeRead(S, first + i, r);
      offsets[i] = pos;
      if (bases) memcpy(bases + pos, r.data(), r.size());
      pos += r.size();
    }
    offsets[n] = pos;
    return;
  }
  if (bases) {   // offsets[] hold the result of the sizing call: fill in place
#pragma omp parallel
    {
      std::string r;
#pragma omp for schedule(dynamic, 256)
      for (long i = 0; i < (long)n; ++i) { makeRead(S, first + (uint64_t)i, r); memcpy(bases + offsets[i], r.data(), r.size()); }
    }
    return;
  }
  std::vector<uint64_t> len(n);
#pragma omp parallel
  {
    std::string r;
#pragma omp for schedule(dynamic, 256)
    for (long i = 0; i < (long)n; ++i) { makeRead(S, first + (uint64_t)i, r); len[i] = r.size(); }
  }
  uint64_t pos = 0;
  for (uint32_t i = 0; i < n; ++i) { offsets[i] = pos; pos += len[i]; }
  offsets[n] = pos;
}

int synth_write_dump(void* h, const char* path) {
  Synth& S = *(Synth*)h; buildDump(S);
  FILE* f = fopen(path, "w");
  if (!f) return -1;
  for (size_t i = 0; i < S.dkeys.size(); ++i) fprintf(f, "%s %u\n", unpack(S.dkeys[i], S.sp.k).c_str(), S.dcounts[i]);
  fclose(f);
  return 0;
}
int synth_write_junctions(void* h, const char* path) {
  Synth& S = *(Synth*)h; buildJunctions(S);
  FILE* f = fopen(path, "w");
  if (!f) return -1;
  for (size_t i = 0; i < S.jkeys.size(); ++i) fprintf(f, "%s %lld\n", unpack(S.jkeys[i], S.sp.k).c_str(), (long long)S.jcounts[i]);
  fclose(f);
  return 0;
}
int synth_write_fasta(void* h, const char* path, uint64_t first, uint32_t n) {
  const Synth& S = *(Synth*)h;
  FILE* f = fopen(path, "w");
  if (!f) return -1;
  std::string r;
  for (uint32_t i = 0; i < n; ++i) {
    makeRead(S, first + i, r);
    fprintf(f, ">read_%07llu\n%s\n", (unsigned long long)(first + i), r.c_str());
  }
  fclose(f);
  return 0;
}

}  // extern "C"
