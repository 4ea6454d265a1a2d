// talc_oracle.cpp — CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).  See talc_oracle.hpp.
// parity unpinned (no reference golden vectors exist; SeqAn2 absent) — SURVEY.md §8c.
//
// Deliberate, documented interpretation choices where the reference is undefined:
//  * Explorer.cpp:705 `for(unsigned int(j); j<...` has no initialiser -> j starts at 0.
//  * std::pow(x,2) (Explorer.cpp:1195-1196,1213-1214) is evaluated as x*x: g++ folds
//    pow(x,2.0) to a multiplication at every optimisation level the reference uses (-O3).
//  * abs(double) in Explorer.cpp:1247 resolves to the floating-point overload.
//  * Explorer.cpp:852 reads newrankings[MAX] even when size()<=MAX; guarded (result is
//    and-ed with false anyway).
//  * stdout debug prints (DEBUG_READ/DEBUG_TEST/DEBUG_USER) are not reproduced.
#include "talc_oracle.hpp"
#include <omp.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>

#include "seqan_shim.hpp"

namespace talc_oracle {

static UBCounters g_ub;
UBCounters& ubCounters() { return g_ub; }

static const char Dict[4] = {'A', 'C', 'G', 'T'};  // Jellyfish.cpp:63, Explorer.cpp:104

// ---------------------------------------------------------------- segment helpers
static TSeq infixS(const TSeq& s, long b, long e) {
  if (b < 0 || e < b || (size_t)e > s.size()) {
    g_ub.infixClamped++;
    if (b < 0) b = 0;
    if ((size_t)e > s.size()) e = (long)s.size();
    if (e < b) return TSeq();
  }
  return s.substr((size_t)b, (size_t)(e - b));
}
static TSeq prefixS(const TSeq& s, long e) { return infixS(s, 0, e); }
static TSeq suffixS(const TSeq& s, long b) { return infixS(s, b, (long)s.size()); }

// ---------------------------------------------------------------- Table
static inline int baseCode(char c) {
  switch (c) {
    case 'A': return 0;
    case 'C': return 1;
    case 'G': return 2;
    case 'T': return 3;
    default: return -1;
  }
}
static bool packKmer(const TSeq& kmer, uint64_t& out) {
  uint64_t v = 0;
  for (char c : kmer) {
    int b = baseCode(c);
    if (b < 0) return false;
    v = (v << 2) | (uint64_t)b;
  }
  out = v;
  return true;
}
static inline uint64_t mix64(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return x;
}
static const uint64_t kEmpty = ~0ULL;

void Table::flatGrow() {
  size_t ncap = fkeys_.empty() ? (1u << 16) : fkeys_.size() * 2;
  std::vector<uint64_t> ok; ok.swap(fkeys_);
  std::vector<colouredCount> ov; ov.swap(fvals_);
  fkeys_.assign(ncap, kEmpty);
  fvals_.assign(ncap, colouredCount(0, 0));
  for (size_t i = 0; i < ok.size(); ++i) {
    if (ok[i] == kEmpty) continue;
    uint64_t s = mix64(ok[i]) & (ncap - 1);
    while (fkeys_[s] != kEmpty) s = (s + 1) & (ncap - 1);
    fkeys_[s] = ok[i];
    fvals_[s] = ov[i];
  }
}
bool Table::flatFind(uint64_t key, uint64_t& slot) const {
  if (fkeys_.empty()) return false;
  uint64_t mask = fkeys_.size() - 1, s = mix64(key) & mask;
  while (true) {
    if (fkeys_[s] == key) { slot = s; return true; }
    if (fkeys_[s] == kEmpty) { slot = s; return false; }
    s = (s + 1) & mask;
  }
}
bool Table::insert(const TSeq& kmer, colouredCount v) {
  if (backend_ == MAP) return map_.insert(std::make_pair(kmer, v)).second;
  uint64_t key;
  if (fK_ == 0 && !kmer.empty()) fK_ = (unsigned)kmer.size();
  if (kmer.size() != fK_ || kmer.size() > 31 || !packKmer(kmer, key))
    return fother_.insert(std::make_pair(kmer, v)).second;
  if ((fcount_ + 1) * 2 > fkeys_.size()) flatGrow();
  uint64_t slot;
  if (flatFind(key, slot)) return false;
  fkeys_[slot] = key; fvals_[slot] = v; fcount_++;
  return true;
}
bool Table::contains(const TSeq& kmer) const {
  if (backend_ == MAP) return map_.count(kmer) > 0;
  uint64_t key, slot;
  if (kmer.size() != fK_ || kmer.size() > 31 || !packKmer(kmer, key)) return fother_.count(kmer) > 0;
  return flatFind(key, slot);
}
colouredCount Table::at(const TSeq& kmer) const {
  if (backend_ == MAP) {
    auto it = map_.find(kmer);
    return it == map_.end() ? colouredCount(0, 0) : it->second;
  }
  uint64_t key, slot;
  if (kmer.size() != fK_ || kmer.size() > 31 || !packKmer(kmer, key)) {
    auto it = fother_.find(kmer);
    return it == fother_.end() ? colouredCount(0, 0) : it->second;
  }
  return flatFind(key, slot) ? fvals_[slot] : colouredCount(0, 0);
}
void Table::setColour(const TSeq& kmer, unsigned int c) {
  if (backend_ == MAP) {
    auto it = map_.find(kmer);
    if (it != map_.end()) it->second.second = c;
    return;
  }
  uint64_t key, slot;
  if (kmer.size() != fK_ || kmer.size() > 31 || !packKmer(kmer, key)) {
    auto it = fother_.find(kmer);
    if (it != fother_.end()) it->second.second = c;
    return;
  }
  if (flatFind(key, slot)) fvals_[slot].second = c;
}
size_t Table::size() const {
  return backend_ == MAP ? map_.size() : (size_t)fcount_ + fother_.size();
}
static TSeq unpackKmer(uint64_t key, unsigned int K) {
  TSeq s(K, 'A');
  for (unsigned int i = 0; i < K; ++i) s[K - 1 - i] = Dict[(key >> (2 * i)) & 3];
  return s;
}
void Table::insertPacked(const uint64_t* keys, const uint32_t* counts, const uint32_t* jcounts, uint64_t n,
                         unsigned int K, bool sortedHint) {
  if (backend_ == MAP) {
    if (sortedHint) {
      // keys ascending => k-mer text ascending (A<C<G<T): O(1) amortised hinted insertion
      for (uint64_t i = 0; i < n; ++i)
        map_.emplace_hint(map_.end(), unpackKmer(keys[i], K), colouredCount(counts[i], jcounts ? jcounts[i] : 0));
    } else {
      for (uint64_t i = 0; i < n; ++i)
        map_.insert(std::make_pair(unpackKmer(keys[i], K), colouredCount(counts[i], jcounts ? jcounts[i] : 0)));
    }
    return;
  }
  if (fK_ == 0) fK_ = K;
  if (fkeys_.empty() && n > (1u << 15)) {   // sized once for the whole array (same content, no regrowth)
    size_t cap = 1u << 16;
    while (cap < 2 * (size_t)n + 2) cap *= 2;
    fkeys_.assign(cap, kEmpty);
    fvals_.assign(cap, colouredCount(0, 0));
    // Parallel over slot ranges: a thread takes, in array order, the keys whose home slot lies in its range, so the
    // winner among duplicates is the one a serial pass picks (std::map::insert, Jellyfish.cpp:262); a probe sequence
    // that would leave the range is put off to the serial pass below.
    int T = omp_get_max_threads();
    if (T > 32) T = 32;
    if (n < (1u << 20)) T = 1;
    std::vector<std::vector<uint64_t>> later(T);
    std::vector<uint64_t> added(T, 0);
    const uint64_t mask = cap - 1;
#pragma omp parallel num_threads(T)
    {
      const int t = omp_get_thread_num();
      const uint64_t lo = cap / T * t, hi = (t == T - 1) ? cap : cap / T * (t + 1);
      uint64_t nadd = 0;
      for (uint64_t i = 0; i < n; ++i) {
        uint64_t sl = mix64(keys[i]) & mask;
        if (sl < lo || sl >= hi) continue;
        while (sl < hi && fkeys_[sl] != kEmpty && fkeys_[sl] != keys[i]) ++sl;
        if (sl >= hi) { later[t].push_back(i); continue; }
        if (fkeys_[sl] == keys[i]) continue;
        fkeys_[sl] = keys[i];
        fvals_[sl] = colouredCount(counts[i], jcounts ? jcounts[i] : 0);
        ++nadd;
      }
      added[t] = nadd;
    }
    for (int t = 0; t < T; ++t) fcount_ += added[t];
    std::vector<uint64_t> rest;
    for (int t = 0; t < T; ++t) rest.insert(rest.end(), later[t].begin(), later[t].end());
    std::sort(rest.begin(), rest.end());   // array order again
    for (uint64_t i : rest) {
      uint64_t slot;
      if (flatFind(keys[i], slot)) continue;
      fkeys_[slot] = keys[i];
      fvals_[slot] = colouredCount(counts[i], jcounts ? jcounts[i] : 0);
      fcount_++;
    }
    return;
  }
  for (uint64_t i = 0; i < n; ++i) {
    if ((fcount_ + 1) * 2 > fkeys_.size()) flatGrow();
    uint64_t slot;
    if (flatFind(keys[i], slot)) continue;
    fkeys_[slot] = keys[i];
    fvals_[slot] = colouredCount(counts[i], jcounts ? jcounts[i] : 0);
    fcount_++;
  }
}

// ---------------------------------------------------------------- Dna5 / I/O
TSeq toDna5(const std::string& raw) {
  TSeq s(raw.size(), 'N');
  for (size_t i = 0; i < raw.size(); ++i) {
    switch (raw[i]) {
      case 'A': case 'a': s[i] = 'A'; break;
      case 'C': case 'c': s[i] = 'C'; break;
      case 'G': case 'g': s[i] = 'G'; break;
      case 'T': case 't': s[i] = 'T'; break;
      default: s[i] = 'N';
    }
  }
  return s;
}
TSeq reverseComplement(const TSeq& s) {
  TSeq r(s.size(), 'N');
  for (size_t i = 0; i < s.size(); ++i) {
    char c = s[s.size() - 1 - i];
    r[i] = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : 'N';
  }
  return r;
}

// io.cpp:26-48 + SeqFileIn::readRecords semantics (FASTA '>' / FASTQ '@', multi-line FASTA,
// id = whole header line after the marker, qualities discarded).
int loadSeqData(std::vector<std::string>& ids, std::vector<TSeq>& seqs, const std::string& file) {
  std::ifstream in(file);
  if (!in) {
    std::cerr << "ERROR: Could not open file " << file << "\n";
    return 1;
  }
  auto chomp = [](std::string& l) { while (!l.empty() && (l.back() == '\r' || l.back() == '\n')) l.pop_back(); };
  std::string line;
  // format from the first non-empty line
  bool fastq = false, started = false, have = false;
  std::string id, raw;
  while (std::getline(in, line)) {
    chomp(line);
    if (!started) {
      if (line.empty()) continue;
      started = true;
      fastq = (line[0] == '@');
      if (!fastq && line[0] != '>') return 1;
    }
    if (fastq) {
      if (line.empty()) continue;
      if (line[0] != '@') return 1;
      id = line.substr(1);
      raw.clear();
      while (std::getline(in, line)) { chomp(line); if (!line.empty() && line[0] == '+') break; raw += line; }
      size_t got = 0;
      while (got < raw.size() && std::getline(in, line)) { chomp(line); got += line.size(); }
      ids.push_back(id);
      seqs.push_back(toDna5(raw));
    } else {
      if (!line.empty() && line[0] == '>') {
        if (have) { ids.push_back(id); seqs.push_back(toDna5(raw)); }
        id = line.substr(1); raw.clear(); have = true;
      } else if (have) {
        raw += line;
      }
    }
  }
  if (!fastq && have) { ids.push_back(id); seqs.push_back(toDna5(raw)); }
  return 0;
}

// io.cpp:50-75 + SeqFileOut::writeRecords to *.fa: '>' id '\n', sequence wrapped at 70 columns.
int outputSeqData(const std::vector<std::string>& ids, const std::vector<TSeq>& seqs, const std::string& file) {
  std::ofstream out(file, std::ios_base::trunc);
  if (!out) {
    std::cerr << "ERROR: Could not open the file " << file << "\n";
    return 1;
  }
  for (size_t r = 0; r < ids.size(); ++r) {
    out << '>' << ids[r] << '\n';
    const TSeq& s = seqs[r];
    for (size_t p = 0; p < s.size(); p += 70) out << s.substr(p, 70) << '\n';
  }
  return 0;
}

void throwToLog(const std::string& seqName, const std::string& chaine, const std::string& outFile) {
  std::ofstream outputFile;
  outputFile.open(outFile, std::ios_base::app);
  outputFile << "[Read: " << seqName << " ]: " << chaine << std::endl;
  outputFile.close();
}

// ---------------------------------------------------------------- table build (Jellyfish.cpp:236-295)
BuildStats buildCDBG(Table& dBG, const std::string& countsTable, const std::string& junctionCountsTable,
                     const Params& P) {
  BuildStats st;
  std::ifstream in(countsTable);
  std::string line;
  while (std::getline(in, line)) {
    std::istringstream iss(line);
    std::string kmer0, count;
    if (iss >> kmer0 >> count) {
      TSeq kmer = toDna5(kmer0);
      // std::stoi(count) >= gp_MIN_COUNT : int vs unsigned comparison (negative -> huge)
      if ((unsigned int)std::stoi(count) >= P.gp_MIN_COUNT) {
        dBG.insert(kmer, colouredCount((unsigned int)std::stoi(count), 0));
        st.actualCounter++;
      }
      st.onlineCounter++;
    } else {
      st.badLines++;
    }
  }
  in.close();
  if (P.gp_useJunctions) {
    std::ifstream jin(junctionCountsTable);
    std::vector<std::pair<TSeq, long>> js;
    while (std::getline(jin, line)) {
      std::istringstream iss2(line);
      std::string jmer0, jcount;
      if (iss2 >> jmer0 >> jcount) js.push_back(std::make_pair(toDna5(jmer0), (long)std::stoi(jcount)));
      else st.badLines++;
    }
    colourJunctions(dBG, js, P);
  }
  return st;
}

// Jellyfish.cpp:278-289: for the junction k-mer and its reverse complement, in file order
void colourJunctions(Table& dBG, const std::vector<std::pair<TSeq, long>>& junctions, const Params& P) {
  for (const auto& j : junctions) {
    TSeq jmer = j.first;
    int jc = (int)j.second;
    bool below = ((unsigned int)jc < P.colouredCountThr);  // int < unsigned: negative never passes
    if (below & dBG.contains(jmer)) dBG.setColour(jmer, (unsigned int)jc);
    jmer = reverseComplement(jmer);
    if (below & dBG.contains(jmer)) dBG.setColour(jmer, (unsigned int)jc);
  }
}

// utils.cpp:658-669
void decolourRepeatsFromDBG(Table& dBG, const Params& P) {
  for (unsigned int b = 0; b < 4; b++) {
    TSeq kmerToRemove(P.K, Dict[b]);
    if (dBG.contains(kmerToRemove)) dBG.setColour(kmerToRemove, 0);
  }
}

// ---------------------------------------------------------------- utils.cpp live helpers
TSeq formNextKmer(const TSeq& kmer, char new_base, Direction direction) {  // utils.cpp:370-387
  TSeq newKmer;
  if (direction == RIGHT) {
    newKmer = suffixS(kmer, kmer.empty() ? 0 : 1);
    newKmer.push_back(new_base);
  } else {
    TSeq tmp(1, new_base);
    tmp += kmer;
    newKmer = prefixS(tmp, (long)kmer.size());
  }
  return newKmer;
}
TSeq getKmerAt(const TSeq& ref, unsigned int position, unsigned int kmerSize) {  // utils.cpp:625
  return infixS(ref, position, (long)position + kmerSize);
}
TSeq extractSolidSequence(const TSeq& ref, unsigned int s, unsigned int e, unsigned int k) {  // :632
  return infixS(ref, s, (long)e + k);
}
TSeq extractWeakSequence(const TSeq& ref, unsigned int e, unsigned int s, unsigned int k) {  // :638
  return infixS(ref, (long)e + k, s);
}
TSeq extractWeakBorderSequence(const TSeq& ref, unsigned int kmPos, unsigned int k, Location loc) {  // :644
  if (loc == HEAD) return prefixS(ref, kmPos);
  return suffixS(ref, (long)kmPos + k);
}

// ---------------------------------------------------------------- queries (Jellyfish.cpp)
std::vector<TSeq> getKmers(const TSeq& Seq, unsigned int kmerSize) {  // :69-82
  unsigned readLength = (unsigned)Seq.size();
  std::vector<TSeq> kmers;
  for (unsigned startPosition = 0; startPosition < readLength - kmerSize + 1; startPosition++)
    kmers.push_back(infixS(Seq, startPosition, (long)startPosition + kmerSize));
  return kmers;
}
std::vector<TSeq> getSuccessors(const TSeq& kmer, Direction direction) {  // :116-126
  std::vector<TSeq> result;
  for (unsigned int b = 0; b < 4; b++) result.push_back(formNextKmer(kmer, Dict[b], direction));
  return result;
}
std::vector<colouredCount> getNextCounts(const Ctx& C, const TSeq& kmer, Direction direction) {  // :308-321
  std::vector<colouredCount> colouredCounts;
  std::vector<TSeq> nextKmers = getSuccessors(kmer, direction);
  for (unsigned int k = 0; k < nextKmers.size(); k++) {
    if (C.dBG->contains(nextKmers[k])) colouredCounts.push_back(C.dBG->at(nextKmers[k]));
    else colouredCounts.push_back(std::make_pair(0u, 0u));
  }
  return colouredCounts;
}
int getOutDegree(const Ctx& C, const TSeq& kmer, Direction direction) {  // :383-393
  std::vector<colouredCount> nextCounts = getNextCounts(C, kmer, direction);
  int outDegree = 0;
  for (unsigned int i = 0; i < nextCounts.size(); i++)
    if (nextCounts[i].first >= C.P.gp_MIN_COUNT) ++outDegree;
  return outDegree;
}
colouredCount getCount(const Ctx& C, const TSeq& kmer) {  // :407-413
  if (C.dBG->contains(kmer)) return C.dBG->at(kmer);
  return std::make_pair(0u, 0u);
}
std::vector<colouredCount> getLRCountsInSR(const Ctx& C, const TSeq& Seq) {  // :485-496
  std::vector<colouredCount> counts;
  std::vector<TSeq> kmers = getKmers(Seq, C.P.K);
  for (unsigned int position = 0; position < kmers.size(); position++) {
    if (C.dBG->contains(kmers[position])) counts.push_back(C.dBG->at(kmers[position]));
    else counts.push_back(std::make_pair(0u, 0u));
  }
  return counts;
}

// ---------------------------------------------------------------- count model (Explorer.cpp:1185-1217)
bool isExpectedbyMyModel(const Params& P, unsigned int nextc, unsigned int cc, Status classe) {
  const double gp_ALPHA = P.gp_ALPHA;
  if ((cc <= 3) & (classe == UNEXPECTED))
    return ((double)nextc <= ((double)(cc + 0.5) + gp_ALPHA * sqrt((double)(cc + 0.5))));
  else if ((cc <= 3) & (classe == EXPECTED))
    return ((double)nextc >= ((double)(cc - 0.5) + (1 - gp_ALPHA) * sqrt((double)(cc - 0.5))));
  else if ((cc > 3) & (classe == UNEXPECTED)) {
    double t = (gp_ALPHA / 2 + sqrt((double)(cc + 0.96)));
    return ((double)nextc <= t * t);
  } else {
    double t = (gp_ALPHA / 2 - sqrt((double)(cc + 0.02)));
    return ((double)nextc >= t * t);
  }
}
bool isExpectedbyMyLastNode(const Params& P, unsigned int nextc, unsigned int cc) {
  const double gp_ALPHA = P.gp_ALPHA;
  bool isExpected = true;
  if (cc <= 3) {
    isExpected &= ((double)nextc <= ((double)(cc + 0.5) + gp_ALPHA * sqrt((double)(cc + 0.5))));
    isExpected &= ((double)nextc >= ((double)(cc - 0.5) + (1 - gp_ALPHA) * sqrt((double)(cc - 0.5))));
  }
  if (cc > 3) {
    double t1 = (gp_ALPHA / 2 + sqrt((double)(cc + 0.96)));
    double t2 = (gp_ALPHA / 2 - sqrt((double)(cc + 0.02)));
    isExpected &= ((double)nextc <= t1 * t1);
    isExpected &= ((double)nextc >= t2 * t2);
  }
  return isExpected;
}

// Explorer.cpp:1226-1298
void tagNextNodes(const Params& P, std::vector<std::pair<Status, double>>& nodeTags,
                  std::vector<colouredCount>& nextCounts, unsigned int count, bool complex) {
  nodeTags.clear();
  int counter = 0;
  double dist = 0;
  unsigned int nextc = 0;
  unsigned int lambda_noise = 0;
  unsigned int nbExpected = 0, nbBreakpoints = 0, nbUnexpected = 0;

  for (unsigned int i = 0; i < nextCounts.size(); i++)
    if ((unsigned int)(int)nextCounts[i].first >= P.gp_MIN_COUNT) counter++;
  if (counter > 0) {
    lambda_noise = (unsigned int)(int)((double)count * P.gp_SR_ERROR_RATE);
    for (unsigned int b = 0; b < nextCounts.size(); b++) {
      nextc = nextCounts[b].first;
      dist = std::fabs((double)count - (double)nextc) / sqrt((double)count);
      if (nextc >= P.gp_MIN_COUNT) {
        if (isExpectedbyMyModel(P, nextc, count, EXPECTED) || (counter == 1)) {
          nodeTags.push_back(std::make_pair(EXPECTED, dist));
          ++nbExpected;
        } else if (lambda_noise >= P.gp_MIN_COUNT) {
          if (!isExpectedbyMyModel(P, nextc, lambda_noise, UNEXPECTED) || (nextCounts[b].second > 0)) {
            nodeTags.push_back(std::make_pair(BREAKPOINT, dist));
            ++nbBreakpoints;
          } else {
            nodeTags.push_back(std::make_pair(UNEXPECTED, dist));
            ++nbUnexpected;
          }
        } else {
          nodeTags.push_back(std::make_pair(BREAKPOINT, dist));
          ++nbBreakpoints;
        }
      } else
        nodeTags.push_back(std::make_pair(UNEXPECTED, dist));
    }
  }
  if ((nbExpected == 0) & (nbBreakpoints == 1)) {
    for (unsigned int t = 0; t < nodeTags.size(); t++)
      if (nodeTags[t].first == BREAKPOINT) nodeTags[t].first = EXPECTED;
  }
  if ((nbExpected == 1) & (nbUnexpected > 0) & !complex) {
    counter = 0;
    unsigned int index = 0;
    for (unsigned int i = 0; i < nextCounts.size(); i++) {
      if (nodeTags[i].first == UNEXPECTED) {
        if (counter == 0) index = i;
        counter += nextCounts[i].first;
        if (nextCounts[index].first < nextCounts[i].first) index = i;
      }
    }
    if (!isExpectedbyMyModel(P, (unsigned int)counter, lambda_noise, UNEXPECTED)) nodeTags[index].first = BREAKPOINT;
  }
}

// ---------------------------------------------------------------- Read.cpp free functions
bool findINRegions(const Params& P, std::vector<kmerStretch>& StartEndKmers,
                   const std::vector<colouredCount>& counts) {  // Read.cpp:440-489
  unsigned int pos = 0, current_startKmer = 0, current_count = 0;
  bool state = false;
  if (counts.size() > 1) {
    while (pos < counts.size()) {
      current_count = counts[pos].first;
      if ((current_count >= P.gp_MIN_COUNT) & (state == false)) {
        current_startKmer = pos;
        state = true;
      } else if ((current_count < P.gp_MIN_COUNT) & (state == true)) {
        StartEndKmers.push_back(std::make_tuple(current_startKmer, pos - 1, UNEXPECTED));
        state = false;
      }
      ++pos;
    }
    if (state == true) StartEndKmers.push_back(std::make_tuple(current_startKmer, pos - 1, UNEXPECTED));
  }
  return !StartEndKmers.empty();
}

double computeSeqErrorThreshold(const Params& P, const std::vector<colouredCount>& counts) {  // Read.cpp:493-518
  std::vector<unsigned int> INcounts;
  double robMean(P.gp_MIN_COUNT);
  unsigned int first, last;
  for (unsigned int pos = 0; pos < counts.size(); pos++)
    if (counts[pos].first >= P.gp_MIN_COUNT) INcounts.push_back(counts[pos].first);
  std::sort(INcounts.begin(), INcounts.end());
  INcounts.size() > 10 ? first = (unsigned int)(0.15 * (double)INcounts.size()) : first = 0;
  INcounts.size() > 10 ? last = (unsigned int)(0.90 * (double)INcounts.size()) : last = (unsigned int)INcounts.size();
  for (unsigned int i = first; i < last; i++) robMean += INcounts[i];
  robMean /= last - first;
  return robMean * P.gp_SR_ERROR_RATE;
}

void analyzeINRegions(const Ctx& C, std::vector<kmerStretch>& StartEndKmers, const TSeq& refSequence,
                      const std::vector<colouredCount>& counts, double solidityThr) {  // Read.cpp:524-600
  const unsigned int K = C.P.K;
  std::vector<kmerStretch> newStartEndKmers;
  TSeq kmer;
  bool OK = true;
  unsigned int new_start_pos = 0, new_end_pos = 0;
  Status regionStatus = UNEXPECTED;
  unsigned int c = 0;
  int span = 0;
  for (unsigned int reg = 0; reg < StartEndKmers.size(); reg++) {
    span = 0;
    OK = true;
    new_start_pos = std::get<0>(StartEndKmers[reg]);
    kmer = getKmerAt(refSequence, new_start_pos, K);
    if (!((StartEndKmers.size() == reg + 1) & newStartEndKmers.empty()) & (getOutDegree(C, kmer, LEFT) == 0) &
        (new_start_pos != 0)) {
      OK = false;
      while ((new_start_pos < std::get<1>(StartEndKmers[reg])) & !OK) {
        ++new_start_pos;
        kmer = getKmerAt(refSequence, new_start_pos, K);
        if (getOutDegree(C, kmer, LEFT) > 1) OK = true;
      }
    }
    new_end_pos = std::get<1>(StartEndKmers[reg]);
    if (OK & !((StartEndKmers.size() == reg + 1) & newStartEndKmers.empty())) {
      kmer = getKmerAt(refSequence, new_end_pos, K);
      if ((getOutDegree(C, kmer, RIGHT) == 0) & (new_end_pos != counts.size() - 1)) {
        OK = false;
        while ((new_end_pos > std::get<0>(StartEndKmers[reg])) & !OK) {
          --new_end_pos;
          kmer = getKmerAt(refSequence, new_end_pos, K);
          if (getOutDegree(C, kmer, RIGHT) > 1) OK = true;
        }
      }
    }
    if (OK) {
      if (reg + 1 < StartEndKmers.size())
        span = ((int)std::get<0>(StartEndKmers[reg + 1]) - (int)(new_end_pos + K));
      if (span < 0) {
        if ((int)std::get<1>(StartEndKmers[reg + 1]) + span >= (int)std::get<0>(StartEndKmers[reg + 1]))
          std::get<0>(StartEndKmers[reg + 1]) -= span;  // unsigned -= negative int
        else {
          std::get<0>(StartEndKmers[reg + 1]) = new_start_pos;
          OK = false;
        }
      }
      if (OK) {
        c = 0;
        for (unsigned int i = new_start_pos; i <= new_end_pos; i++) c < counts[i].first ? c = counts[i].first : c += 0;
        !isExpectedbyMyModel(C.P, (unsigned int)c, (unsigned int)solidityThr, UNEXPECTED) ? regionStatus = EXPECTED
                                                                                          : regionStatus = LOWCOUNT;
        if (regionStatus == EXPECTED) newStartEndKmers.push_back(std::make_tuple(new_start_pos, new_end_pos, regionStatus));
      }
    }
  }
  if (!newStartEndKmers.empty()) StartEndKmers = newStartEndKmers;
}

// ---------------------------------------------------------------- Trail
Trail::Trail()
    : m_lastStep(TSeq(), colouredCount(0, 0)), m_lastScore(0), m_nbFailuresInARow(0), m_nbBreakpoints(0),
      m_distance(0), m_leftAnchor(-1), m_rightAnchor(-1) {}
Trail::Trail(const TSeq& kmer, const colouredCount& ccount)
    : m_sequence(kmer), m_lastStep(kmer, ccount), m_lastScore(0), m_nbFailuresInARow(0), m_nbBreakpoints(0),
      m_distance(0), m_leftAnchor(-1), m_rightAnchor(-1) {}

// Trail.cpp:313-330 addNewBase, :332-338 makeTipNode
static TSeq addNewBase(const TSeq& sequence, char newBase, Direction direction) {
  if (direction == LEFT) return TSeq(1, newBase) + sequence;
  TSeq s = sequence;
  s.push_back(newBase);
  return s;
}
static tipNode makeTipNode(const TSeq& kmer, char base, Direction direction, unsigned int count) {
  return std::make_pair(formNextKmer(kmer, base, direction), std::make_pair(count, 0u));
}
Trail::Trail(const Trail& path, char newBase, Direction direction, unsigned int count)
    : m_sequence(addNewBase(path.getSeq(), newBase, direction)),
      m_lastStep(makeTipNode(path.getLastKmer(), newBase, direction, count)),
      m_lastScore(path.getLastScore()), m_nbFailuresInARow(path.getNbFailuresInARow()),
      m_nbBreakpoints(path.getNbBreakpoints()), m_distance(path.getDistance()),
      m_leftAnchor(path.getLeftAnchor()), m_rightAnchor(path.getRightAnchor()) {}

std::vector<colouredCount> Trail::whatsNext(const Ctx& C, Direction direction) const {
  return getNextCounts(C, m_lastStep.first, direction);
}

void Trail::Overlapscore(const TSeq& reference, Direction direction) {  // Trail.cpp:145-174
  // sequence 0 = reference, sequence 1 = candidate in both branches of :150-159
  const shim::SimpleScore sc = {4, -3, -2};
  switch (direction) {
    case LEFT:   // AlignConfig<false,false,true,true>
      m_lastScore = shim::globalAlignmentScore(reference, m_sequence, sc, false, false, true, true);
      break;
    case RIGHT:  // AlignConfig<true,true,false,false>
      m_lastScore = shim::globalAlignmentScore(reference, m_sequence, sc, true, true, false, false);
      break;
  }
}

bool Trail::seedAndExtend(const Ctx& C, const TSeq& reference, Direction direction, int xdrop,
                          unsigned int MAX_FAILURES) {  // Trail.cpp:193-216
  TSeq candidate(m_sequence);
  bool ok = true;
  std::tuple<TSeq, TSeq, int, double, bool> extension_res =
      getSeedAndExtension(reference, candidate, xdrop, direction, C.P.K);
  ok = (std::get<1>(extension_res).size() == candidate.size());
  if (!ok) m_nbFailuresInARow++;
  else m_nbFailuresInARow = 0;
  m_lastScore = std::get<3>(extension_res);
  if (direction == RIGHT) m_rightAnchor = std::get<2>(extension_res);
  else m_leftAnchor = std::get<2>(extension_res);
  ok = (m_nbFailuresInARow <= MAX_FAILURES);
  ok &= !std::get<4>(extension_res);
  return ok;
}

bool Trail::checkAims(const std::vector<anchorTuple>& aims, Direction direction) {  // Trail.cpp:273-285
  bool isEqual = false;
  unsigned int i = 0;
  while ((!isEqual) & (i < aims.size())) {
    isEqual = (m_lastStep.first == std::get<0>(aims[i]));
    ++i;
  }
  if (isEqual) {
    if (direction == RIGHT) m_rightAnchor = (int)std::get<1>(aims[i - 1]);
    else m_leftAnchor = (int)std::get<1>(aims[i - 1]);
  }
  return isEqual;
}

bool Trail::ThinkIveAlreadyGotThere(const TSeq& history) const {  // Trail.cpp:289-302
  long alreadyOccurred = -1;
  if (history.size() > m_lastStep.first.size()) alreadyOccurred = shim::findFirst(history, m_lastStep.first);
  return alreadyOccurred > 0;
}

// Trail.cpp:341-437
std::tuple<TSeq, TSeq, int, double, bool> getSeedAndExtension(const TSeq& reference, const TSeq& candidate,
                                                              int xdrop, Direction direction,
                                                              unsigned int seedSize) {
  TSeq seq1, seq2, refExtension, histExtension;
  int posOnRef = -1;
  bool state = true;
  const shim::SimpleScore scoringScheme = {0, -1, -1};
  bool stopThere = false;
  double score = 0;

  if (reference.size() < candidate.size()) {
    seq1 = candidate;
    seq2 = reference;
    state = false;
  } else {
    seq1 = reference;
    seq2 = candidate;
  }
  if (seq1.size() < seedSize || seq2.size() < seedSize) {
    // the reference would build a seed with wrapped-around coordinates: undefined
    g_ub.seedTooShort++;
    return std::make_tuple(refExtension, histExtension, posOnRef, (double)((-1) * xdrop), true);
  }
  if (direction == RIGHT) {
    shim::Seed seedR = {0, 0, (long)seedSize - 1, (long)seedSize - 1};
    shim::extendSeed(seedR, seq1, seq2, shim::EXTEND_RIGHT, scoringScheme, xdrop);
    if (state) {
      histExtension = prefixS(candidate, seedR.endV);
      refExtension = prefixS(reference, seedR.endH);
      posOnRef = (int)seedR.endH;
    } else {
      histExtension = prefixS(candidate, seedR.endH);
      refExtension = prefixS(reference, seedR.endV);
      posOnRef = (int)seedR.endV;
    }
  } else {
    shim::Seed seedL = {(long)seq1.size() - (long)seedSize, (long)seq2.size() - (long)seedSize,
                        (long)seq1.size() - 1, (long)seq2.size() - 1};
    shim::extendSeed(seedL, seq1, seq2, shim::EXTEND_LEFT, scoringScheme, xdrop);
    if (state) {
      histExtension = suffixS(candidate, seedL.beginV);
      refExtension = suffixS(reference, seedL.beginH);
      posOnRef = (int)seedL.beginH;
    } else {
      histExtension = suffixS(candidate, seedL.beginH);
      refExtension = suffixS(reference, seedL.beginV);
      posOnRef = (int)seedL.beginV;
    }
  }
  if (std::max(refExtension.size(), histExtension.size()) >= seedSize) {
    if (histExtension.size() <= refExtension.size()) score = shim::globalAlignmentScore(refExtension, histExtension, scoringScheme);
    else score = shim::globalAlignmentScore(histExtension, refExtension, scoringScheme);
  } else {
    score = (-1) * xdrop;
    stopThere = true;
  }
  return std::make_tuple(refExtension, histExtension, posOnRef, score, stopThere);
}

// ---------------------------------------------------------------- Trajectory
Trajectory::Trajectory()
    : m_leftAnchor(0), m_rightAnchor(0), m_score(-200000), m_IDscore(-1), m_nbBreakpoints(0), m_lastScore(0),
      m_distance(0) {}
Trajectory::Trajectory(const Trail& trail)
    : m_sequence(trail.getSeq()), m_leftAnchor((unsigned int)trail.getLeftAnchor()),
      m_rightAnchor((unsigned int)trail.getRightAnchor()), m_score(-200000), m_IDscore(0),
      m_nbBreakpoints(trail.getNbBreakpoints()), m_lastScore(trail.getLastScore()),
      m_distance(trail.getDistance() / (trail.getLength() + 0.01)) {}

void Trajectory::trim(unsigned int minSize, unsigned int intervalLength, unsigned int nbFailuresInARow,
                      Direction direction) {  // Trajectory.cpp:89-112
  unsigned int nbBases = nbFailuresInARow * intervalLength;
  TSeq newEdge;
  if (getLength() >= nbBases + minSize) {
    if (direction == RIGHT) newEdge = prefixS(m_sequence, (long)getLength() - (long)nbBases);
    else newEdge = suffixS(m_sequence, nbBases);
  } else
    newEdge = m_sequence;
  m_sequence = newEdge;
}

void Trajectory::reshape(const TSeq& reference, unsigned int kmerSize, Direction direction, bool shorter) {  // :114-155
  std::tuple<TSeq, TSeq, int, double> extension_results;
  TSeq newSeq, tmp;
  int xdrop1 = 0;
  tmp = m_sequence;
  xdrop1 = (int)m_lastScore * (-1);
  if (!shorter) {
    extension_results = findStopPosition(tmp, reference, xdrop1, direction, kmerSize);
    if (direction == LEFT) newSeq = suffixS(tmp, std::get<2>(extension_results));
    else newSeq = prefixS(tmp, std::get<2>(extension_results));
  } else {
    extension_results = findStopPosition(reference, tmp, xdrop1, direction, kmerSize);
    if (direction == LEFT) {
      newSeq = prefixS(reference, std::get<2>(extension_results));
      newSeq += tmp;
    } else {
      tmp = suffixS(reference, std::get<2>(extension_results));
      newSeq = m_sequence;
      newSeq += tmp;
    }
  }
  m_IDscore = computePercentID(std::get<0>(extension_results), std::get<1>(extension_results));
  m_score = std::get<3>(extension_results);
  m_sequence = newSeq;
}

bool Trajectory::cutAnchors(Location location, unsigned int limit, unsigned int kmerSize) {  // :157-211
  TSeq truncSeq;
  const TSeq oldSeq = m_sequence;
  bool isOK = true;
  unsigned int len = getLength();
  switch (location) {
    case HEAD:
      if (len > kmerSize) truncSeq = prefixS(oldSeq, (long)len - kmerSize);
      break;
    case TAIL:
      if (len > kmerSize) truncSeq = suffixS(oldSeq, kmerSize);
      break;
    case INNER:
      if (len >= 2 * kmerSize) {
        truncSeq = infixS(oldSeq, kmerSize, (long)len - kmerSize);
      } else if ((len < 2 * kmerSize) & (len > kmerSize)) {
        if (m_rightAnchor + 2 * kmerSize - len <= limit) m_rightAnchor = m_rightAnchor + 2 * kmerSize - len;
        else isOK = false;
      } else
        isOK = false;
      break;
    default: break;
  }
  m_sequence = truncSeq;
  return isOK;
}

void Trajectory::scoreSequence(const TSeq& reference) {  // :239-243
  m_score = computeEditDistance(reference, m_sequence);
  m_IDscore = computeIDScore(reference, m_sequence) / std::max(reference.size(), m_sequence.size());
}

unsigned int findBestBridge(const std::vector<Trajectory>& trajectories) {  // :282-303
  unsigned int index = 0;
  std::vector<unsigned int> exAequo1;
  for (unsigned int i = 1; i < trajectories.size(); i++)
    if (trajectories[i].getScore() > trajectories[index].getScore()) index = i;
  for (unsigned int i = index + 1; i < trajectories.size(); i++)
    if (trajectories[i].getScore() == trajectories[index].getScore()) exAequo1.push_back(i);
  if (!exAequo1.empty()) {
    for (unsigned int i = 0; i < exAequo1.size(); i++)
      if (trajectories[exAequo1[i]].getMeanDistance() > trajectories[index].getMeanDistance()) index = exAequo1[i];
  }
  return index;
}

std::pair<bool, unsigned int> findBestBORDER(const std::vector<Trajectory>& trajectories) {  // :306-334
  unsigned int index = 0;
  std::vector<unsigned int> exAequo1;
  bool isConvenient = true;
  if (trajectories.empty()) isConvenient = false;
  else {
    for (unsigned int i = 1; i < trajectories.size(); i++)
      if (trajectories[i].getScore() > trajectories[index].getScore()) index = i;
    for (unsigned int i = index + 1; i < trajectories.size(); i++)
      if (trajectories[i].getScore() == trajectories[index].getScore()) exAequo1.push_back(i);
    if (!exAequo1.empty()) {
      for (unsigned int i = 0; i < exAequo1.size(); i++)
        if (trajectories[exAequo1[i]].getMeanDistance() > trajectories[index].getMeanDistance()) index = exAequo1[i];
    }
  }
  return std::make_pair(isConvenient, index);
}

double computeIDScore(const TSeq& gap, const TSeq& history) {  // :337-384
  double score = -1;
  unsigned int len1 = (unsigned)gap.size(), len2 = (unsigned)history.size();
  if ((len1 > 0) & (len2 > 0)) {
    const shim::SimpleScore sc = {1, 0, 0};
    if (len1 > len2) score = shim::localAlignmentScore(gap, history, sc);
    else score = shim::localAlignmentScore(history, gap, sc);
  } else
    score = -1;
  return score;
}

double computeEditDistance(const TSeq& reference, const TSeq& history) {  // :386-428
  double score = -100000;
  if ((history.size() > 0) & (reference.size() > 0)) {
    const shim::SimpleScore sc = {0, -1, -1};
    if (reference.size() >= history.size()) score = shim::globalAlignmentScore(reference, history, sc);
    else score = shim::globalAlignmentScore(history, reference, sc);
  } else
    score = -100000;
  return score;
}

std::tuple<TSeq, TSeq, int, double> findStopPosition(const TSeq& reference, const TSeq& shorterPath, int xdrop,
                                                     Direction direction, unsigned int kmerSize) {  // :482-503
  int xdrop1 = xdrop;
  bool goFurther = true;
  std::tuple<TSeq, TSeq, int, double, bool> extension_results, new_extension_results;
  new_extension_results = getSeedAndExtension(reference, shorterPath, xdrop1, direction, kmerSize);
  do {
    --xdrop1;
    extension_results = new_extension_results;
    new_extension_results = getSeedAndExtension(reference, shorterPath, xdrop1, direction, kmerSize);
    if (std::get<1>(new_extension_results).size() < std::get<1>(extension_results).size()) goFurther = false;
  } while (goFurther & (xdrop1 > 0));
  return std::make_tuple(std::get<0>(extension_results), std::get<1>(extension_results),
                         std::get<2>(extension_results), std::get<3>(extension_results));
}

double computePercentID(const TSeq& seq1, const TSeq& seq2) {  // :505-528
  const shim::SimpleScore sc = {1, 0, 0};
  double score, len;
  if (seq2.size() <= seq1.size()) {
    len = (double)seq1.size();
    score = shim::localAlignmentScore(seq1, seq2, sc) / len;
  } else {
    len = (double)seq2.size();
    score = shim::localAlignmentScore(seq2, seq1, sc) / len;
  }
  return score;
}

// ---------------------------------------------------------------- scoring / gardening (Explorer.cpp:689-865)
void scoreBridges(const Params& P, std::vector<Trail>& newCompetingPaths, unsigned int stepCounter,
                  const TSeq& reference, Direction direction) {
  TSeq truncatedReference;
  int bound = 0;
  if (direction == RIGHT) {
    bound = (int)(P.K + stepCounter + P.gp_WINDOW_SIZE);
    if ((size_t)bound >= reference.size()) truncatedReference = reference;
    else truncatedReference = prefixS(reference, bound);
  } else {
    bound = (int)reference.size() - (int)P.K - (int)stepCounter - (int)P.gp_WINDOW_SIZE;
    if (bound < 0) truncatedReference = reference;
    else truncatedReference = suffixS(reference, bound);
  }
  // Explorer.cpp:705: `for(unsigned int(j); j<...` — uninitialised in the reference; j=0 here.
  for (unsigned int j = 0; j < newCompetingPaths.size(); j++) newCompetingPaths[j].Overlapscore(truncatedReference, direction);
}

typedef std::tuple<unsigned int, double, double> IdxSD;
typedef std::tuple<unsigned int, unsigned int, unsigned int, unsigned int> Rank4;

bool doABitOfGardening(const Params& P, std::vector<unsigned int>& indexOfKeptPaths,
                       std::vector<Trail>& newCompetingPaths) {
  const unsigned int MAXB = P.gp_MAX_NB_COMPETING_PATHS;
  indexOfKeptPaths.clear();
  std::vector<IdxSD> indexOfTrails1, indexOfTrails2;
  std::vector<unsigned int> rankWithTies1, rankWithTies2;
  std::vector<Rank4> rankings, newrankings;
  bool ties = true, isComplex = false;
  unsigned int nb, s = 0;

  nb = (unsigned int)newCompetingPaths.size();
  nb = std::min(nb, MAXB);
  for (unsigned t = 0; t < newCompetingPaths.size(); t++) {
    rankings.push_back(std::make_tuple(t, 0u, 0u, 0u));
    rankWithTies1.push_back(t);
    rankWithTies2.push_back(t);
    indexOfTrails1.push_back(std::make_tuple(t, newCompetingPaths[t].getLastScore(), newCompetingPaths[t].getDistance()));
    indexOfTrails2.push_back(std::make_tuple(t, newCompetingPaths[t].getLastScore(), newCompetingPaths[t].getDistance()));
  }
  if (rankings.empty()) return false;  // reference would index [0] of an empty vector; callers never pass one

  std::sort(indexOfTrails1.begin(), indexOfTrails1.end(),
            [](const IdxSD& lhs, const IdxSD& rhs) { return std::get<1>(lhs) > std::get<1>(rhs); });
  std::get<1>(rankings[std::get<0>(indexOfTrails1[0])]) = rankWithTies1[0];
  for (unsigned int t = 1; t < rankWithTies1.size(); t++) {
    if (std::get<1>(indexOfTrails1[t]) == std::get<1>(indexOfTrails1[t - 1])) rankWithTies1[t] = rankWithTies1[t - 1];
    else rankWithTies1[t] = rankWithTies1[t - 1] + 1;
    std::get<1>(rankings[std::get<0>(indexOfTrails1[t])]) = rankWithTies1[t];
  }
  std::sort(indexOfTrails2.begin(), indexOfTrails2.end(),
            [](const IdxSD& lhs, const IdxSD& rhs) { return std::get<2>(lhs) < std::get<2>(rhs); });
  std::get<2>(rankings[std::get<0>(indexOfTrails2[0])]) = rankWithTies2[0];
  for (unsigned int t = 1; t < rankWithTies2.size(); t++) {
    if (std::get<2>(indexOfTrails2[t]) == std::get<2>(indexOfTrails2[t - 1])) rankWithTies2[t] = rankWithTies2[t - 1];
    else rankWithTies2[t] = rankWithTies2[t - 1] + 1;
    std::get<2>(rankings[std::get<0>(indexOfTrails2[t])]) = rankWithTies2[t];
  }
  for (unsigned int t = 0; t < rankings.size(); t++) {
    std::get<3>(rankings[t]) = std::get<1>(rankings[t]) + std::get<2>(rankings[t]);
    if ((std::get<3>(rankings[t]) == 0) || (newCompetingPaths.size() <= MAXB)) newrankings.push_back(rankings[t]);
  }
  if (newrankings.empty()) {
    std::sort(rankings.begin(), rankings.end(),
              [](const Rank4& lhs, const Rank4& rhs) { return std::get<1>(lhs) < std::get<1>(rhs); });
    s = 0;
    ties = false;
    do {
      if ((s <= nb) || ties) newrankings.push_back(rankings[s]);
      if (s < rankings.size() - 1) ties = (std::get<1>(rankings[s + 1]) == std::get<1>(rankings[s]));
      ++s;
    } while (((s <= nb) || ties) & (s < rankings.size()));

    if (newrankings.size() > MAXB) {
      if (std::get<1>(newrankings[0]) != std::get<1>(newrankings[MAXB])) {
        newrankings.pop_back();
        ties = true;
        while ((newrankings.size() >= MAXB) & ties) {
          ties = (std::get<1>(newrankings.back()) == std::get<1>(newrankings[newrankings.size() - 2]));
          ties |= (newrankings.size() >= MAXB);
          if (ties) newrankings.pop_back();
        }
      }
      bool sameAsMax = false;
      if (newrankings.size() > MAXB) sameAsMax = (std::get<1>(newrankings[0]) == std::get<1>(newrankings[MAXB]));
      else g_ub.gardeningOOB++;
      if ((newrankings.size() > MAXB) & sameAsMax) {
        isComplex = true;
        std::sort(newrankings.begin(), newrankings.end(),
                  [](const Rank4& lhs, const Rank4& rhs) { return std::get<2>(lhs) < std::get<2>(rhs); });
        for (unsigned int t = 0; t < MAXB; t++) indexOfKeptPaths.push_back(std::get<0>(newrankings[t]));
      }
    }
    for (unsigned int t = 0; t < newrankings.size(); t++) indexOfKeptPaths.push_back(std::get<0>(newrankings[t]));
  } else
    for (unsigned int t = 0; t < newrankings.size(); t++) indexOfKeptPaths.push_back(std::get<0>(newrankings[t]));
  return isComplex;
}

// ---------------------------------------------------------------- Explorer
static kmerStretch make_empty_KmPos() { return std::make_tuple(0u, 0u, UNCORRECTED); }

Explorer::Explorer(const Ctx& c, const TSeq& refSequence, const std::vector<colouredCount>& coverage, double lambda,
                   Trace* trace)
    : C(c), m_trace(trace), m_sequence(refSequence), m_weakSequence(TSeq(), UNCORRECTED), m_coverage(coverage),
      m_priorLambda_noise(lambda), m_LEFT_KMpositions(make_empty_KmPos()), m_RIGHT_KMpositions(make_empty_KmPos()),
      m_location(UNKNOWN), m_direction(RIGHT), m_complexRegion(false) {}

void Explorer::reset() {  // Explorer.cpp:155-174 — m_complexRegion is NOT reset
  m_weakSequence = std::make_pair(TSeq(), UNCORRECTED);
  m_LEFT_KMpositions = make_empty_KmPos();
  m_RIGHT_KMpositions = make_empty_KmPos();
  m_LEFT_anchors.clear();
  m_RIGHT_anchors.clear();
  m_location = UNKNOWN;
  m_direction = RIGHT;
  m_fullPaths.clear();
  m_longPaths.clear();
  m_shortPaths.clear();
}

void Explorer::setWeakSequence() {  // :218-226
  const unsigned int K = C.P.K;
  if (m_location == INNER)
    m_weakSequence.first = extractWeakSequence(m_sequence, std::get<1>(m_LEFT_KMpositions), std::get<0>(m_RIGHT_KMpositions), K);
  else if (m_location == HEAD)
    m_weakSequence.first = extractWeakBorderSequence(m_sequence, std::get<0>(m_RIGHT_KMpositions), K, m_location);
  else
    m_weakSequence.first = extractWeakBorderSequence(m_sequence, std::get<1>(m_LEFT_KMpositions), K, m_location);
  m_weakSequence.second = UNCORRECTED;
}

void Explorer::initializeINNER(kmerStretch startKmPos, kmerStretch endKmPos, Direction direction) {  // :228-243
  reset();
  m_location = INNER;
  m_direction = direction;
  m_LEFT_KMpositions = startKmPos;
  m_RIGHT_KMpositions = endKmPos;
  setWeakSequence();
  anchorLEFTHandSide();
  anchorRIGHTHandSide();
  traceSearch();
}
void Explorer::initializeHEAD(kmerStretch kmPos) {  // :245-257
  reset();
  m_location = HEAD;
  m_direction = LEFT;
  m_RIGHT_KMpositions = kmPos;
  setWeakSequence();
  anchorRIGHTHandSide();
  traceSearch();
}
void Explorer::initializeTAIL(kmerStretch kmPos) {  // :259-271
  reset();
  m_location = TAIL;
  m_direction = RIGHT;
  m_LEFT_KMpositions = kmPos;
  setWeakSequence();
  anchorLEFTHandSide();
  traceSearch();
}

void Explorer::traceSearch() {
  if (!m_trace || !m_trace->enabled) return;
  m_trace->ev.push_back({TR_SEARCH, (long)m_location, (long)m_direction, (long)m_LEFT_anchors.size(),
                         (long)m_RIGHT_anchors.size(), 0.0, ""});
  for (auto& a : m_LEFT_anchors)
    m_trace->ev.push_back({TR_ANCHOR, 0, (long)std::get<1>(a), (long)std::get<2>(a), 0, 0.0, std::get<0>(a)});
  for (auto& a : m_RIGHT_anchors)
    m_trace->ev.push_back({TR_ANCHOR, 1, (long)std::get<1>(a), (long)std::get<2>(a), 0, 0.0, std::get<0>(a)});
}
void Explorer::traceResult(bool success) {
  if (!m_trace || !m_trace->enabled) return;
  m_trace->ev.push_back({TR_RESULT, (long)m_location, (long)success, (long)std::get<1>(m_LEFT_KMpositions),
                         (long)std::get<0>(m_RIGHT_KMpositions), 0.0, m_weakSequence.first});
}

// Explorer.cpp:402-411
static void sortAnchorsByNearest(double cc, std::vector<anchorTuple>& anchors) {
  std::sort(anchors.begin(), anchors.end(), [cc](const anchorTuple& lhs, const anchorTuple& rhs) {
    return abs((int)cc - (int)std::get<2>(lhs)) < abs((int)cc - (int)std::get<2>(rhs));
  });
}

void Explorer::anchorLEFTHandSide() {  // Explorer.cpp:413-478
  const Params& P = C.P;
  const unsigned int K = P.K;
  bool goFurther = true;
  unsigned int nbKmers = std::get<1>(m_LEFT_KMpositions) - std::get<0>(m_LEFT_KMpositions) + 1;
  unsigned int pivot = std::get<1>(m_LEFT_KMpositions);
  unsigned int limit = std::get<0>(m_LEFT_KMpositions);
  int degree = 0;
  TSeq anchor;
  std::vector<unsigned int> anchorPos;
  double current_count = (double)m_coverage[pivot].first;
  double next_count = 0;
  unsigned int j = pivot;
  anchorPos.push_back(pivot);
  while (goFurther & (j >= limit + 1)) {
    next_count = m_coverage[j - 1].first;
    if ((next_count >= P.gp_MIN_COUNT) & (next_count < P.p_MAX_IN_COUNT))
      goFurther = isExpectedbyMyLastNode(P, (unsigned int)next_count, (unsigned int)current_count);
    else
      goFurther = false;
    if (!goFurther & (current_count >= P.gp_MIN_COUNT) & (next_count >= P.gp_MIN_COUNT) & (next_count < P.p_MAX_IN_COUNT)) {
      anchorPos.push_back(j - 1);
      goFurther = true;
      current_count = next_count;
    }
    --j;
  }
  for (int anc = 0; anc < (int)anchorPos.size(); anc++) {
    anchor = getKmerAt(m_sequence, anchorPos[anc], K);
    degree = getOutDegree(C, anchor, RIGHT);
    // :454 — the count is read at index `anc`, not at anchorPos[anc]
    if ((anc == 0) || ((anc != 0) & (degree > 1)))
      m_LEFT_anchors.push_back(std::make_tuple(anchor, anchorPos[anc], m_coverage[anc].first));
  }
  if (m_LEFT_anchors.size() < std::min(P.p_MIN_START_ANCHORS, nbKmers)) {
    j = pivot;
    goFurther = true;
    while ((j >= limit + 1) & (m_LEFT_anchors.size() < std::min(P.p_MIN_START_ANCHORS, nbKmers))) {
      // :465 `for(unsigned int i(0); i<size; --i)` inspects element 0 only (i wraps around)
      for (unsigned int i = 0; i < m_LEFT_anchors.size(); --i) goFurther &= (std::get<1>(m_LEFT_anchors[i]) != (j - 1));
      if (goFurther) {
        anchor = getKmerAt(m_sequence, j - 1, K);
        degree = getOutDegree(C, anchor, RIGHT);
        if (degree > 1) m_LEFT_anchors.push_back(std::make_tuple(anchor, j - 1, m_coverage[j - 1].first));
      }
      --j;
    }
  }
  sortAnchorsByNearest(m_priorLambda_noise / P.gp_SR_ERROR_RATE, m_LEFT_anchors);
}

void Explorer::anchorRIGHTHandSide() {  // Explorer.cpp:480-543
  const Params& P = C.P;
  const unsigned int K = P.K;
  bool goFurther = true;
  unsigned int nbKmers = std::get<1>(m_RIGHT_KMpositions) - std::get<0>(m_RIGHT_KMpositions) + 1;
  unsigned int pivot = std::get<0>(m_RIGHT_KMpositions);
  unsigned int limit = std::get<1>(m_RIGHT_KMpositions);
  TSeq anchor;
  std::vector<unsigned int> anchorPos;
  double current_count = (double)m_coverage[pivot].first;
  double next_count = 0;
  unsigned int j = pivot;
  anchorPos.push_back(pivot);
  while (goFurther & ((j + 1) <= limit)) {
    next_count = m_coverage[j + 1].first;
    if ((next_count >= P.gp_MIN_COUNT) & (next_count < P.p_MAX_IN_COUNT))
      goFurther = isExpectedbyMyLastNode(P, (unsigned int)next_count, (unsigned int)current_count);
    else
      goFurther = false;
    if (!goFurther & (current_count >= P.gp_MIN_COUNT) & (next_count >= P.gp_MIN_COUNT) & (next_count < P.p_MAX_IN_COUNT)) {
      anchorPos.push_back(j + 1);
      goFurther = true;
      current_count = next_count;
    }
    ++j;
  }
  for (int anc = 0; anc < (int)anchorPos.size(); anc++) {
    anchor = getKmerAt(m_sequence, anchorPos[anc], K);
    int degree = getOutDegree(C, anchor, LEFT);
    if ((anc == 0) || ((anc != 0) & (degree > 1)))
      m_RIGHT_anchors.push_back(std::make_tuple(anchor, anchorPos[anc], m_coverage[anc].first));
  }
  if (m_RIGHT_anchors.size() < std::min(P.p_MIN_START_ANCHORS, nbKmers)) {
    j = pivot;
    goFurther = true;
    while (((j + 1) <= limit) & (m_RIGHT_anchors.size() < std::min(P.p_MIN_START_ANCHORS, nbKmers))) {
      for (unsigned int i = 0; i < m_RIGHT_anchors.size(); --i) goFurther &= (std::get<1>(m_RIGHT_anchors[i]) != (j + 1));
      if (goFurther) {
        anchor = getKmerAt(m_sequence, j + 1, K);
        int degree = getOutDegree(C, anchor, LEFT);
        if (degree > 1) m_RIGHT_anchors.push_back(std::make_tuple(anchor, j + 1, m_coverage[j + 1].first));
      }
      --j;  // :539 — decrements (sic); unsigned wrap-around ends the loop
    }
  }
  sortAnchorsByNearest(m_priorLambda_noise / P.gp_SR_ERROR_RATE, m_RIGHT_anchors);
}

void Explorer::recordBridge(const Trail& trail) { m_fullPaths.push_back(Trajectory(trail)); }  // :1097

void Explorer::recordEdge(const Trail& trail, const TSeq& reference) {  // :1103-1118
  bool shorter = true;
  Trajectory myTip = Trajectory(trail);
  myTip.trim(C.P.K, C.P.p_CHECK_INTERVAL, trail.getNbFailuresInARow(), m_direction);
  shorter = (myTip.getLength() <= reference.size());
  myTip.reshape(reference, C.P.K, m_direction, shorter);
  if (myTip.cutAnchors(m_location, 0, C.P.K)) {
    if (shorter) m_shortPaths.push_back(myTip);
    else m_longPaths.push_back(myTip);
  }
}

void Explorer::oneMoreStep(const TSeq& reference, std::vector<Trail>& competingPaths, unsigned int& stepCounter,
                           unsigned int PATH_MAXLENGTH) {  // :546-612
  (void)PATH_MAXLENGTH;
  const Params& P = C.P;
  std::vector<Trail> newCompetingPaths;
  std::vector<colouredCount> nextCounts;
  std::vector<std::pair<Status, double>> nodeTags;
  std::vector<unsigned int> indexOfKeptPaths;
  bool cycle = false, aimReached = false, complex = false;
  for (unsigned int t = 0; t < competingPaths.size(); t++) {
    complex = (competingPaths.size() > P.gp_MAX_NB_COMPETING_PATHS);
    nextCounts = competingPaths[t].whatsNext(C, m_direction);
    tagNextNodes(P, nodeTags, nextCounts, competingPaths[t].getLastCount(), complex);
    for (unsigned int i = 0; i < nodeTags.size(); i++) {
      if (nodeTags[i].first != UNEXPECTED) {
        newCompetingPaths.push_back(Trail(competingPaths[t], Dict[i], m_direction, nextCounts[i].first));
        if (nodeTags[i].first == BREAKPOINT) newCompetingPaths.back().recordBreakpoint();
        newCompetingPaths.back().recordDistance(nodeTags[i].second);
        if (m_direction == RIGHT) aimReached = newCompetingPaths.back().checkAims(m_RIGHT_anchors, m_direction);
        else aimReached = newCompetingPaths.back().checkAims(m_LEFT_anchors, m_direction);
        if (aimReached) {
          recordBridge(newCompetingPaths.back());
          if (newCompetingPaths.back().getLength() > reference.size()) newCompetingPaths.pop_back();
        } else {
          cycle = newCompetingPaths.back().ThinkIveAlreadyGotThere(competingPaths[t].getSeq());
          if (cycle) newCompetingPaths.pop_back();
        }
      }
    }
  }
  complex = (newCompetingPaths.size() > P.gp_MAX_NB_COMPETING_PATHS);
  ++stepCounter;
  if (complex & (stepCounter % P.p_CHECK_INTERVAL == 0)) {
    scoreBridges(P, newCompetingPaths, stepCounter, reference, m_direction);
    m_complexRegion |= doABitOfGardening(P, indexOfKeptPaths, newCompetingPaths);
    competingPaths.clear();
    for (unsigned int t = 0; t < indexOfKeptPaths.size(); t++) competingPaths.push_back(newCompetingPaths[indexOfKeptPaths[t]]);
  } else
    competingPaths = newCompetingPaths;
  if (m_trace && m_trace->enabled && m_trace->steps)
    m_trace->ev.push_back({TR_STEP, (long)stepCounter, (long)competingPaths.size(), (long)m_fullPaths.size(), 0, 0.0, ""});
}

void Explorer::oneMoreStepInTheDark(int& xdrop, const TSeq& reference, std::vector<Trail>& competingPaths,
                                    unsigned int& stepCounter, unsigned int PATH_MAXLENGTH) {  // :615-687
  const Params& P = C.P;
  std::vector<Trail> newCompetingPaths;
  std::vector<colouredCount> nextCounts;
  std::vector<std::pair<Status, double>> nodeTags;
  std::vector<unsigned int> indexOfKeptPaths;
  bool complex = false, cycle = false;
  unsigned int counter = 0;
  for (unsigned int t = 0; t < competingPaths.size(); t++) {
    counter = 0;
    complex = (competingPaths.size() > 7);  // :634 hard-coded
    nextCounts = competingPaths[t].whatsNext(C, m_direction);
    tagNextNodes(P, nodeTags, nextCounts, competingPaths[t].getLastCount(), complex);
    for (unsigned int i = 0; i < nodeTags.size(); i++) {
      if (nodeTags[i].first != UNEXPECTED) {
        ++counter;
        newCompetingPaths.push_back(Trail(competingPaths[t], Dict[i], m_direction, nextCounts[i].first));
        if (nodeTags[i].first == BREAKPOINT) newCompetingPaths.back().recordBreakpoint();
        newCompetingPaths.back().recordDistance(nodeTags[i].second);
        cycle = newCompetingPaths.back().ThinkIveAlreadyGotThere(competingPaths[t].getSeq());
        if (cycle || (stepCounter + 1 > PATH_MAXLENGTH)) {
          newCompetingPaths.back().seedAndExtend(C, reference, m_direction, xdrop, (unsigned int)P.p_MAX_NB_BORDER_FAILURES);
          recordEdge(newCompetingPaths.back(), reference);
          newCompetingPaths.pop_back();
        }
      }
    }
    if (counter == 0) {  // dead-end path
      competingPaths[t].seedAndExtend(C, reference, m_direction, xdrop, (unsigned int)P.p_MAX_NB_BORDER_FAILURES);
      recordEdge(competingPaths[t], reference);
    }
  }
  ++stepCounter;
  if ((stepCounter % P.p_CHECK_INTERVAL == 0) || (newCompetingPaths.size() >= P.p_MAX_NB_OF_BORDER_PATHS)) {
    scoreEdges(xdrop, newCompetingPaths, stepCounter, reference, m_direction);
    if (newCompetingPaths.size() > 5) {
      m_complexRegion |= doABitOfGardening(P, indexOfKeptPaths, newCompetingPaths);
      competingPaths.clear();
      for (unsigned int t = 0; t < indexOfKeptPaths.size(); t++) competingPaths.push_back(newCompetingPaths[indexOfKeptPaths[t]]);
    } else
      competingPaths = newCompetingPaths;
  } else
    competingPaths = newCompetingPaths;
  if (m_trace && m_trace->enabled && m_trace->steps)
    m_trace->ev.push_back({TR_STEP, (long)stepCounter, (long)competingPaths.size(),
                           (long)(m_shortPaths.size() + m_longPaths.size()), (long)xdrop, 0.0, ""});
}

void Explorer::scoreEdges(int& xdrop, std::vector<Trail>& newCompetingPaths, unsigned int stepCounter,
                          const TSeq& reference, Direction direction) {  // :709-740
  (void)stepCounter;
  bool boolean = true;
  std::vector<Trail> newSelectedPaths, trashPaths;
  int new_xdrop = 0, current_xdrop = 0;
  if (!newCompetingPaths.empty()) {
    xdrop += 2;
    for (unsigned int t = 0; t < newCompetingPaths.size(); t++) {
      boolean = newCompetingPaths[t].seedAndExtend(C, reference, direction, xdrop, (unsigned int)C.P.p_MAX_NB_BORDER_FAILURES);
      if (!boolean) trashPaths.push_back(newCompetingPaths[t]);
      else {
        newSelectedPaths.push_back(newCompetingPaths[t]);
        current_xdrop = (int)(newCompetingPaths[t].getLastScore() * (-1));
        if ((new_xdrop > current_xdrop) || (new_xdrop == 0)) new_xdrop = current_xdrop;
      }
    }
    xdrop = new_xdrop;
    newCompetingPaths = newSelectedPaths;
    if (newCompetingPaths.empty()) {
      for (unsigned int t = 0; t < trashPaths.size(); t++) recordEdge(trashPaths[t], reference);
    }
  }
}

Trajectory Explorer::sortOutBestBorder() {  // :310-329
  std::pair<bool, unsigned int> result = findBestBORDER(m_longPaths);
  if (result.first) return m_longPaths[result.second];
  result = findBestBORDER(m_shortPaths);
  if (result.first) return m_shortPaths[result.second];
  return Trajectory();
}

bool Explorer::searchBridge() {  // :868-989
  const Params& P = C.P;
  const unsigned int K = P.K;
  bool pathHasBeenFound = false;
  unsigned int PATH_MAXLENGTH = 0, limit, stepCounter = 0;
  double diff = 0;
  std::vector<anchorTuple> aims, anchors;
  colouredCount currentCount;
  TSeq currentGap, currentAnchor, currentTarget, currentRefSeq, bestPath;
  unsigned int whichStart = 0, bestOne;
  std::vector<Trail> competingPaths;

  anchors = (m_direction == LEFT) ? m_RIGHT_anchors : m_LEFT_anchors;
  aims = (m_direction == LEFT) ? m_LEFT_anchors : m_RIGHT_anchors;
  limit = (unsigned int)anchors.size();
  limit = std::min(limit, P.p_MAX_START_ANCHORS);

  for (int s = 0; s < (int)limit; s++) {
    if (!pathHasBeenFound) {
      competingPaths.clear();
      m_fullPaths.clear();
      stepCounter = 0;
      whichStart = std::get<1>(anchors[s]);
      currentTarget = (m_direction == RIGHT)
                          ? extractSolidSequence(m_sequence, std::get<0>(m_RIGHT_KMpositions), std::get<1>(m_RIGHT_KMpositions), K)
                          : extractSolidSequence(m_sequence, std::get<0>(m_LEFT_KMpositions), std::get<1>(m_LEFT_KMpositions), K);
      currentGap.clear();
      if ((m_direction == RIGHT) & (whichStart + K < std::get<0>(m_RIGHT_KMpositions)))
        currentGap = extractWeakSequence(m_sequence, whichStart, std::get<0>(m_RIGHT_KMpositions), K);
      else if ((m_direction == LEFT) & (std::get<1>(m_LEFT_KMpositions) + K < whichStart))
        currentGap = extractWeakSequence(m_sequence, std::get<1>(m_LEFT_KMpositions), whichStart, K);
      PATH_MAXLENGTH = (unsigned int)(int)(1.2 * currentGap.size() + 3 * K);

      currentAnchor = std::get<0>(anchors[s]);
      currentCount = m_coverage[whichStart];
      competingPaths.push_back(Trail(currentAnchor, currentCount));
      if (m_direction == RIGHT) {
        currentRefSeq = currentAnchor;
        currentRefSeq += currentGap;
        currentRefSeq += currentTarget;
        competingPaths[0].setLeftAnchor((int)whichStart);
      } else {
        currentRefSeq = currentTarget;
        currentRefSeq += currentGap;
        currentRefSeq += currentAnchor;
        competingPaths[0].setRightAnchor((int)whichStart);
      }
      while ((!competingPaths.empty()) & (competingPaths.size() <= P.p_MAX_NB_OF_INNER_PATHS) & (stepCounter < PATH_MAXLENGTH))
        oneMoreStep(currentRefSeq, competingPaths, stepCounter, PATH_MAXLENGTH);

      if (!m_fullPaths.empty() & !pathHasBeenFound) {
        std::vector<unsigned int> index;
        unsigned int limit2 = std::get<1>(m_RIGHT_KMpositions);
        for (unsigned int t = 0; t < m_fullPaths.size(); t++) {
          m_fullPaths[t].scoreSequence(currentRefSeq);
          if (m_fullPaths[t].cutAnchors(m_location, limit2, K)) index.push_back(t);
        }
        if (m_fullPaths.size() != index.size()) {
          m_shortPaths = m_fullPaths;
          m_fullPaths.clear();
          // :959 — pushes m_shortPaths[t], not m_shortPaths[index[t]]
          for (unsigned int t = 0; t < index.size(); t++) m_fullPaths.push_back(m_shortPaths[t]);
        }
        if (!m_fullPaths.empty()) {
          bestOne = findBestBridge(m_fullPaths);
          bestPath = m_fullPaths[bestOne].getSeq();
          diff = (double)getWeakLength() - (double)bestPath.size();
          // :970-971 minScore is computed but unused; the test below uses gp_MIN_INNER_SCORE
          if ((diff < getWeakLength() * 0.05 || ((getWeakLength() < 6) & (bestPath.size() < 6))) &
              (m_fullPaths[bestOne].getIDScore() >= P.gp_MIN_INNER_SCORE)) {
            std::get<1>(m_LEFT_KMpositions) = m_fullPaths[bestOne].getLeftAnchor();
            std::get<0>(m_RIGHT_KMpositions) = m_fullPaths[bestOne].getRightAnchor();
            m_weakSequence.first = m_fullPaths[bestOne].getSeq();
            m_weakSequence.second = CORRECTED;
            pathHasBeenFound = true;
          }
        }
      }
    }
  }
  traceResult(pathHasBeenFound);
  return pathHasBeenFound;
}

bool Explorer::searchEdge() {  // :992-1081
  const Params& P = C.P;
  const unsigned int K = P.K;
  bool pathHasBeenFound = false;
  unsigned int PATH_MAXLENGTH = 0, limit, stepCounter = 0;
  double diff = 0, minScore = 0;
  std::vector<anchorTuple> anchors;
  colouredCount currentCount;
  TSeq currentGap, currentAnchor, currentRefSeq;
  unsigned int whichStart = 0;
  Trajectory winner;
  int xdrop;
  std::vector<Trail> competingPaths;

  anchors = (m_direction == LEFT) ? m_RIGHT_anchors : m_LEFT_anchors;
  limit = (unsigned int)anchors.size();
  limit = std::min(limit, P.p_MAX_START_ANCHORS);

  for (int s = 0; s < (int)limit; s++) {
    stepCounter = 0;
    competingPaths.clear();
    xdrop = (int)((int)P.p_CHECK_INTERVAL * P.p_ALLOWED_FAILURE_RATE + 1);  // :1031 -> 2
    whichStart = std::get<1>(anchors[s]);
    currentGap = extractWeakBorderSequence(m_sequence, whichStart, K, m_location);
    PATH_MAXLENGTH = (unsigned int)(int)(1.2 * currentGap.size() + 2 * K);
    currentAnchor = std::get<0>(anchors[s]);
    currentCount = m_coverage[whichStart];
    competingPaths.push_back(Trail(currentAnchor, currentCount));
    if (m_direction == RIGHT) {
      currentRefSeq = currentAnchor;
      currentRefSeq += currentGap;
      competingPaths[0].setLeftAnchor((int)whichStart);
    } else {
      currentRefSeq = currentGap;
      currentRefSeq += currentAnchor;
      competingPaths[0].setRightAnchor((int)whichStart);
    }
    while ((!competingPaths.empty()) & (competingPaths.size() <= P.p_MAX_NB_OF_INNER_PATHS) & (stepCounter < PATH_MAXLENGTH))
      oneMoreStepInTheDark(xdrop, currentRefSeq, competingPaths, stepCounter, PATH_MAXLENGTH);
  }
  if (!m_shortPaths.empty() || !m_longPaths.empty()) {
    winner = sortOutBestBorder();
    diff = (double)getWeakLength() - (double)winner.getLength();
    if ((getWeakLength() >= 300) || m_complexRegion) minScore = std::max(0.75, P.gp_MIN_BORDER_SCORE);
    else minScore = P.gp_MIN_BORDER_SCORE;
    if ((diff < getWeakLength() * 0.05 || ((getWeakLength() < 6) & (winner.getLength() < 6))) &
        (winner.getIDScore() >= minScore)) {
      pathHasBeenFound = true;
      m_weakSequence.first = winner.getSeq();
      m_weakSequence.second = CORRECTED;
      if (m_location == TAIL) std::get<1>(m_LEFT_KMpositions) = winner.getLeftAnchor();
      else std::get<0>(m_RIGHT_KMpositions) = winner.getRightAnchor();
    }
  }
  traceResult(pathHasBeenFound);
  return pathHasBeenFound;
}

// ---------------------------------------------------------------- Read
Read::Read(const Ctx& c, const std::string& id, const TSeq& sequence, Trace* trace)
    : C(c), m_trace(trace), m_id(id), m_sequence(sequence), m_priorLambda_noise(c.P.gp_MIN_COUNT),
      m_head(TSeq(), UNCORRECTED), m_tail(TSeq(), UNCORRECTED), m_nbInKmers(0) {}

bool Read::reCoverage() {  // Read.cpp:174-195
  m_nbInKmers = 0;
  m_coverage = getLRCountsInSR(C, m_sequence);
  if (!m_coverage.empty())
    for (unsigned int p = 0; p < m_coverage.size(); p++)
      if (m_coverage[p].first > C.P.gp_MIN_COUNT) m_nbInKmers++;
  return (m_nbInKmers > 0);
}

bool Read::setInitialStructure() {  // Read.cpp:214-258
  const unsigned int K = C.P.K;
  TSeq temp;
  unsigned int len = 0;
  m_head = std::make_pair(temp, ABSENT);
  m_tail = std::make_pair(temp, ABSENT);
  if (!m_InKmersPositions.empty()) {
    if (std::get<0>(m_InKmersPositions[0]) > 0) {
      temp = extractWeakBorderSequence(m_sequence, std::get<0>(m_InKmersPositions[0]), K, HEAD);
      m_head = std::make_pair(temp, UNCORRECTED);
      len += (unsigned)temp.size();
    }
  }
  if (std::get<1>(m_InKmersPositions.back()) + 1 < m_coverage.size()) {
    temp = extractWeakBorderSequence(m_sequence, std::get<1>(m_InKmersPositions.back()), K, TAIL);
    m_tail = std::make_pair(temp, UNCORRECTED);
    len += (unsigned)temp.size();
  }
  for (unsigned int i = 0; i + 1 < m_InKmersPositions.size(); i++) {
    temp = extractSolidSequence(m_sequence, std::get<0>(m_InKmersPositions[i]), std::get<1>(m_InKmersPositions[i]), K);
    m_newInnerStructure.push_back(std::make_pair(temp, std::get<2>(m_InKmersPositions[i])));
    len += (unsigned)temp.size();
    if (std::get<0>(m_InKmersPositions[i + 1]) > std::get<1>(m_InKmersPositions[i]) + K)
      temp = extractWeakSequence(m_sequence, std::get<1>(m_InKmersPositions[i]), std::get<0>(m_InKmersPositions[i + 1]), K);
    else
      temp.clear();
    m_newInnerStructure.push_back(std::make_pair(temp, UNCORRECTED));
    len += (unsigned)temp.size();
  }
  temp = extractSolidSequence(m_sequence, std::get<0>(m_InKmersPositions.back()), std::get<1>(m_InKmersPositions.back()), K);
  m_newInnerStructure.push_back(std::make_pair(temp, std::get<2>(m_InKmersPositions.back())));
  len += (unsigned)temp.size();
  return (len == m_sequence.size());
}

bool Read::defineStructure2() {  // Read.cpp:260-276
  bool checok = true;
  double thr(C.P.gp_MIN_COUNT);
  checok = findINRegions(C.P, m_InKmersPositions, m_coverage);
  thr = computeSeqErrorThreshold(C.P, m_coverage);
  m_priorLambda_noise = thr;
  analyzeINRegions(C, m_InKmersPositions, m_sequence, m_coverage, thr);
  if (!m_InKmersPositions.empty()) checok &= setInitialStructure();
  if (m_trace && m_trace->enabled) {
    m_trace->ev.push_back({TR_THRESHOLD, 0, 0, 0, 0, thr, ""});
    for (auto& r : m_InKmersPositions) m_trace->ev.push_back({TR_REGION, (long)std::get<0>(r), (long)std::get<1>(r), 0, 0, 0.0, ""});
  }
  return checok;
}

std::pair<TSeq, Status> Read::getSolidRegion(kmerStretch coordinates) {  // :288-292
  TSeq seq = extractSolidSequence(m_sequence, std::get<0>(coordinates), std::get<1>(coordinates), C.P.K);
  return std::make_pair(seq, std::get<2>(coordinates));
}
void Read::updateINNER(Explorer& e, int reg) {  // :294-303
  m_InKmersPositions[reg] = e.getLEFTHandPositions();
  m_InKmersPositions[reg + 1] = e.getRIGHTHandPositions();
  m_newInnerStructure[2 * reg + 1] = e.getWeakSeq();
  m_newInnerStructure[2 * reg] = getSolidRegion(e.getLEFTHandPositions());
  m_newInnerStructure[2 * (reg + 1)] = getSolidRegion(e.getRIGHTHandPositions());
}
void Read::updateHEAD(Explorer& e) {  // :305-311
  m_InKmersPositions[0] = e.getRIGHTHandPositions();
  m_newInnerStructure[0] = getSolidRegion(e.getRIGHTHandPositions());
  m_head = e.getWeakSeq();
}
void Read::updateTAIL(Explorer& e) {  // :313-318
  m_InKmersPositions.back() = e.getLEFTHandPositions();
  m_newInnerStructure.back() = getSolidRegion(e.getLEFTHandPositions());
  m_tail = e.getWeakSeq();
}
void Read::updateCorrSeq() {  // :320-326
  m_correction.clear();
  m_correction += m_head.first;
  for (size_t reg = 0; reg < m_newInnerStructure.size(); reg++) m_correction += m_newInnerStructure[reg].first;
  m_correction += m_tail.first;
}

void Read::correct2() {  // Read.cpp:336-386
  Explorer myExplorer(C, m_sequence, m_coverage, m_priorLambda_noise, m_trace);
  bool success;
  if (m_InKmersPositions.size() > 0) {
    for (int reg = 0; reg < (int)m_InKmersPositions.size() - 1; reg++) {
      myExplorer.initializeINNER(m_InKmersPositions[reg], m_InKmersPositions[reg + 1], RIGHT);
      success = myExplorer.searchBridge();
      if (!success) {
        myExplorer.initializeINNER(m_InKmersPositions[reg], m_InKmersPositions[reg + 1], LEFT);
        myExplorer.searchBridge();
      }
      updateINNER(myExplorer, reg);
    }
    if ((m_head.second != ABSENT) & (m_head.first.size() <= C.P.maxBorderLength)) {
      myExplorer.initializeHEAD(m_InKmersPositions[0]);
      success = myExplorer.searchEdge();
      if (success) updateHEAD(myExplorer);
    }
    if ((m_tail.second != ABSENT) & (m_tail.first.size() <= C.P.maxBorderLength)) {
      myExplorer.initializeTAIL(m_InKmersPositions.back());
      success = myExplorer.searchEdge();
      if (success) updateTAIL(myExplorer);
    }
  }
  updateCorrSeq();
}

// Read.cpp:418-433 (Read::outputBasicReadStats) without the file: the four numbers of a row
void Read::basicReadStats(long& rawLength, unsigned int& nbInKmersBefore, int& nbSReg, long& corrLength) const {
  nbInKmersBefore = 0;
  if (!m_InKmersPositions.empty())
    for (unsigned int i = 0; i < m_InKmersPositions.size(); i++)
      nbInKmersBefore += std::get<1>(m_InKmersPositions[i]) - std::get<0>(m_InKmersPositions[i]) + 1;   // :423
  nbSReg = (int)m_InKmersPositions.size();                                                                // :424
  rawLength = (long)m_sequence.size();                                                                    // :428
  corrLength = (long)m_correction.size();                                                                 // :431
}

// main.cpp:247-308, one iteration
ReadStatus correctOneRead(const Ctx& C, const std::string& id, TSeq& seq, Trace* trace, BasicReadStats* stats) {
  if (C.P.gp_reverse) seq = reverseComplement(seq);  // :253 (before any length test)
  Read myLRead(C, id, seq, trace);
  ReadStatus rs = RS_SKIPPED_SHORT;
  if (myLRead.getLength() > (int)C.P.K) {           // :262
    if (myLRead.reCoverage()) {                     // :267
      if (myLRead.defineStructure2()) {             // :272
        myLRead.correct2();                         // :277
        seq = myLRead.getCorrSeq();                 // :285
        if (C.P.gp_reverse) seq = reverseComplement(seq);  // :286
        rs = RS_CORRECTED;
      } else rs = RS_NO_STRUCTURE;                  // :290
    } else rs = RS_NO_SOLID_KMER;                   // :294
    if (stats) {                                    // :305 (commented out in the reference)
      stats->written = true;
      myLRead.basicReadStats(stats->rawLength, stats->nbInKmersBefore, stats->nbSReg, stats->corrLength);
    }
  }
  return rs;
}

}  // namespace talc_oracle
