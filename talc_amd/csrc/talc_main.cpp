// talc_main.cpp — the drop-in `talc` command line over libtalc_hip.so.
//
// Keeps the reference's CLI / Settings / output-file surface (main.cpp:83-325, Settings.cpp:74-185,
// io.cpp:26-111, Read.cpp:394-415) and replaces its per-read OpenMP loop (main.cpp:247-308) by a
// multi-GPU read sharder: one host thread + one talc_ctx per GPU, reads dealt in contiguous
// blocks balanced by bases, the k-mer table replicated on every GPU; records are merged in
// input order by the single writer.  Pure host code (g++): it only talks to the C ABI.
//
// Options = the reference's table (same names, defaults, ranges), plus:
//   --gpus N         number of GPUs to use (default: all visible)
//   --batch-reads N  reads per device batch (default: two or more batches per worker — two workers per GPU —, 20000..200000)
//   --read-stats     append the per-read rows of Read::outputBasicReadStats (Read.cpp:418-433) to <o>.stats_basics.txt
//                    (the reference has that call commented out, main.cpp:305, and only ever writes the header)
//   -k accepts 18..31 (the reference stops at 30, main.cpp:115-116; 31 still fits 62 bits)
//   -SR / -j accept a Jellyfish 2 count file (.jf, `jellyfish count` output) as well as the text dump, in either mode
//   -qm jellyfish2 works (the reference's is dead code): with -jf2 DIR the counts come from `DIR/jellyfish
//                    dump` (the tool the reference would have queried k-mer by k-mer, Jellyfish.cpp:323-379), without it
//                    from the native reader of the .jf; with neither a -jf2 nor a .jf the reference's behaviour is kept
// -t/--num_threads is accepted and ignored (the parallelism is on the device).
// Differences, all documented in INTEGRATION.md: stdout carries the [TALC] banners but none of
// the reference's always-on debug dumps; log lines are written in input order.
#include <omp.h>

#include <cerrno>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <spawn.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include "talc_hip.h"

namespace {

struct Options {
  std::string seqFile, outPrefix = "out", queryMode = "memory", dump, jdump, jf2;
  talc_params p;
  bool haveK = false, haveSR = false, useJ = false;
  int gpus = -1;
  int nthreads = 1;
  uint32_t batchReads = 200000;
  bool haveBatchReads = false;
  bool readStats = false;
};

void usage(FILE* f) {
  fprintf(f,
          "TALC: Transcriptome-Aware Long Read Correction (MI355X hot path)\n"
          "SYNOPSIS  talc [OPTIONS] <long reads .fa/.fq> -k K -SR <jellyfish dump>\n"
          "  -o, --output TEXT           prefix of the output files (default: out)\n"
          "  -k, --kmerSize INT          k-mer length, 18..31 (required)\n"
          "  -qm, --query-mode TEXT      memory | jellyfish2 (default: memory)\n"
          "  -SR, --SRCounts TEXT        short-read k-mer counts: `jellyfish dump -c` text or the .jf itself (required)\n"
          "  -j, --junctions TEXT        k-mers flanking junctions and their counts\n"
          "  -jf2, --pathToJF2 TEXT      directory of the jellyfish program: -qm jellyfish2 then reads the .jf through\n"
          "                              `jellyfish dump` (without it: the native .jf reader)\n"
          "  --MIN_INNER_SCORE FLOAT     [0.3,0.9] default 0.7\n"
          "  --MIN_BORDER_SCORE FLOAT    [0.5,0.9] default 0.7\n"
          "  --MIN_COUNT INT             >= 2, default 2\n"
          "  --SR_ERROR_RATE DOUBLE      [0.01,0.1] default 0.025\n"
          "  --WINDOW_SIZE INT           >= 6, default 9\n"
          "  --MAX_NB_BRANCHES INT       >= 5, default 7\n"
          "  --ALPHA_FOR_PRED FLOAT      >= 0.67, default 2.57\n"
          "  -t, --num_threads INT       accepted (host threads are not the parallel resource here)\n"
          "  --DEBUG_MODE TEXT           accepted, unused\n"
          "  -rev, --reverse             reverse-complement the long reads before correction\n"
          "  --gpus INT                  GPUs to use (default: all)\n"
          "  --batch-reads INT           reads per device batch (default: at least two batches per worker, 20000..200000)\n"
          "  --read-stats                append per-read rows to <o>.stats_basics.txt (Read.cpp:418-433)\n"
          "  -h, --help / --version\n");
}

[[noreturn]] void parse_error(const std::string& msg) {
  std::cerr << "talc: " << msg << "\n";
  exit(1);  // main.cpp:199: PARSE_ERROR -> return 1
}

double num(const char* s, const char* name) {
  char* end = nullptr;
  double v = strtod(s, &end);
  if (end == s || *end != 0) parse_error(std::string("the given value '") + s + "' cannot be cast for " + name);
  return v;
}
void range(double v, double lo, double hi, const char* name) {
  if (v < lo || v > hi) parse_error(std::string("value out of range for ") + name);
}

Options parse(int argc, const char** argv) {
  Options o;
  talc_params_default(&o.p);
  auto need = [&](int& i) -> const char* {
    if (i + 1 >= argc) parse_error(std::string("option requires an argument: ") + argv[i]);
    return argv[++i];
  };
  auto is = [](const std::string& a, const char* s, const char* l) { return a == std::string("-") + s || a == std::string("--") + l; };
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    if (is(a, "o", "output")) o.outPrefix = need(i);
    else if (is(a, "k", "kmerSize")) { double v = num(need(i), "k"); range(v, 18, 31, "k"); o.p.k = (uint32_t)v; o.haveK = true; }
    else if (is(a, "qm", "query-mode")) { o.queryMode = need(i); if (o.queryMode != "memory" && o.queryMode != "jellyfish2") parse_error("the given value '" + o.queryMode + "' is not in the list of allowed values [memory, jellyfish2]"); }
    else if (is(a, "SR", "SRCounts")) { o.dump = need(i); o.haveSR = true; }
    else if (is(a, "j", "junctions")) { o.jdump = need(i); o.useJ = true; }
    else if (is(a, "jf2", "pathToJF2")) o.jf2 = need(i);
    else if (is(a, "MIN_INNER_SCORE", "MIN_INNER_SCORE")) { o.p.min_inner_score = num(need(i), "MIN_INNER_SCORE"); range(o.p.min_inner_score, 0.3, 0.9, "MIN_INNER_SCORE"); }
    else if (is(a, "MIN_BORDER_SCORE", "MIN_BORDER_SCORE")) { o.p.min_border_score = num(need(i), "MIN_BORDER_SCORE"); range(o.p.min_border_score, 0.5, 0.9, "MIN_BORDER_SCORE"); }
    else if (is(a, "MIN_COUNT", "MIN_COUNT")) { double v = num(need(i), "MIN_COUNT"); range(v, 2, 4e9, "MIN_COUNT"); o.p.min_count = (uint32_t)v; }
    else if (is(a, "SR_ERROR_RATE", "SR_ERROR_RATE")) { o.p.sr_error_rate = num(need(i), "SR_ERROR_RATE"); range(o.p.sr_error_rate, 0.01, 0.1, "SR_ERROR_RATE"); }
    else if (is(a, "WINDOW_SIZE", "WINDOW_SIZE")) { double v = num(need(i), "WINDOW_SIZE"); range(v, 6, 4e9, "WINDOW_SIZE"); o.p.window_size = (uint32_t)v; }
    else if (is(a, "MAX_NB_BRANCHES", "MAX_NB_BRANCHES")) { double v = num(need(i), "MAX_NB_BRANCHES"); range(v, 5, 64, "MAX_NB_BRANCHES"); o.p.max_nb_competing_paths = (uint32_t)v; }
    else if (is(a, "ALPHA_FOR_PRED", "ALPHA_FOR_PRED")) { o.p.alpha = num(need(i), "ALPHA_FOR_PRED"); range(o.p.alpha, 0.67, 1e300, "ALPHA_FOR_PRED"); }
    else if (is(a, "t", "num_threads")) { double v = num(need(i), "num_threads"); range(v, 1, 1e9, "num_threads"); o.nthreads = (int)v; }
    else if (is(a, "DEBUG_MODE", "DEBUG_MODE")) need(i);
    else if (is(a, "rev", "reverse")) o.p.reverse = 1;
    else if (a == "--gpus") { o.gpus = (int)num(need(i), "gpus"); range(o.gpus, 1, 64, "gpus"); }
    else if (a == "--batch-reads") { double v = num(need(i), "batch-reads"); range(v, 1, 4e9, "batch-reads"); o.batchReads = (uint32_t)v; o.haveBatchReads = true; }
    else if (a == "--read-stats") o.readStats = true;
    else if (a == "-h" || a == "--help") { usage(stdout); exit(0); }
    else if (a == "--version") { std::cout << "talc version: 1.01\nLast update: September 2019\n"; exit(0); }
    else if (a.size() > 1 && a[0] == '-') parse_error("unknown option: " + a);
    else {
      if (!o.seqFile.empty()) parse_error("too many arguments");
      o.seqFile = a;
    }
  }
  if (o.seqFile.empty()) parse_error("not enough arguments were provided");
  if (!o.haveK) parse_error("option requires a value: -k, --kmerSize");
  if (!o.haveSR) parse_error("option requires a value: -SR, --SRCounts");
  o.p.use_junctions = o.useJ ? 1 : 0;
  return o;
}

// Settings.cpp:160-185
void outputConfig(const Options& o, const std::string& statFile) {
  std::ofstream f(o.outPrefix + ".config.txt", std::ios_base::trunc);
  f << "TALC: Parameters used for sample: " << o.outPrefix << "\n"
    << "****************************" << "\n"
    << "INPUT=" << o.seqFile << "\n"
    << "OUTPUT=" << o.outPrefix << "\n"
    << "STATS=" << statFile << "\n"
    << "****************************" << "\n"
    << "KmerSize=" << o.p.k << "\n"
    << "Junction mode activated? " << (o.useJ ? 1 : 0) << "\n"
    << "queryMode=" << o.queryMode << "\n"
    << "****************************" << "\n"
    << "MIN_INNER_SCORE=" << o.p.min_inner_score << "\n"
    << "MIN_BORDER_SCORE=" << o.p.min_border_score << "\n"
    << "MAX_NB_BRANCHES=" << o.p.max_nb_competing_paths << "\n"
    << "ALPHA=" << o.p.alpha << "\n"
    << "MIN_SR_COUNT=" << o.p.min_count << "\n"
    << "WINDOW_SIZE=" << o.p.window_size << "\n"
    << "****************************" << std::endl;
}

// Read.cpp:394-415
void setBasicReadStatsHeader(const std::string& statFile) {
  std::ofstream f(statFile, std::ios_base::trunc);
  f << "read_name\traw_length\twhead_length\twtail_length\tnbInKmers\tnbSolidKmers\tnbSolidReg\tnbInWeakReg\tnbInCorrReg\t"
       "CorrHead?\tCorrHeadLen\tCorrTail?\tCorrTailLen\tCorrlength\tnbInKmers2\n";
}

// Streaming FASTA / FASTQ reader (replaces loadSeqData, io.cpp:26-48, which holds the whole file, main.cpp:209-211):
// the format is decided by the first non-empty line ('>' or '@'); id = the whole header line after the marker;
// multi-line sequences are concatenated; FASTQ qualities are skipped by length.  Sequences are kept as raw text:
// the device applies the Dna5 conversion.
// lines of a file through one large buffer (read(2) in 8 MB pieces, memchr for the line ends): the streaming reader's
// std::getline loop was what bounded the whole correction phase once the GPU side had become quick (1 GB/s of FASTA)
class LineReader {
 public:
  explicit LineReader(const std::string& file) : buf_(8u << 20) { fd_ = open(file.c_str(), O_RDONLY); }
  ~LineReader() { if (fd_ >= 0) close(fd_); }
  bool ok() const { return fd_ >= 0; }
  // the next line without its "\n" / "\r\n"; the pointer is valid until the next call
  bool getline(const char*& p, size_t& len) {
    while (true) {
      const char* nl = (pos_ < end_) ? (const char*)memchr(buf_.data() + pos_, '\n', end_ - pos_) : nullptr;
      if (nl) {
        p = buf_.data() + pos_;
        len = (size_t)(nl - p);
        pos_ = (size_t)(nl - buf_.data()) + 1;
        while (len && (p[len - 1] == '\r' || p[len - 1] == '\n')) --len;
        return true;
      }
      if (eof_) {
        if (pos_ >= end_) return false;
        p = buf_.data() + pos_; len = end_ - pos_; pos_ = end_;   // a last line without a newline
        while (len && (p[len - 1] == '\r' || p[len - 1] == '\n')) --len;
        return true;
      }
      // no line end in what is left: move the tail to the front (grow the buffer for a line longer than it) and read on
      if (pos_ > 0) { memmove(&buf_[0], buf_.data() + pos_, end_ - pos_); end_ -= pos_; pos_ = 0; }
      if (end_ == buf_.size()) buf_.resize(buf_.size() * 2);
      const ssize_t r = read(fd_, &buf_[end_], buf_.size() - end_);
      if (r <= 0) eof_ = true; else end_ += (size_t)r;
    }
  }

 private:
  int fd_ = -1;
  std::vector<char> buf_;
  size_t pos_ = 0, end_ = 0;
  bool eof_ = false;
};

// FASTA / FASTQ records one by one (multi-line sequences, blank lines, CRLF), the sequence appended to any sink
class SeqReader {
 public:
  explicit SeqReader(const std::string& file) : in_(file) {
    if (!in_.ok()) { std::cerr << "ERROR: Could not open file " << file << "\n"; ok_ = false; return; }
    while (in_.getline(lp_, ll_)) {   // first non-empty line decides the format
      if (ll_ == 0) continue;
      fastq_ = lp_[0] == '@';
      if (!fastq_ && lp_[0] != '>') ok_ = false;
      pending_ = true;
      break;
    }
  }
  bool ok() const { return ok_; }
  bool fastq() const { return fastq_; }
  // next record; false at the end of the file (or on a malformed FASTQ header: bad() then says so).  sink(p, n) is called
  // with every piece of the record's sequence and returns false to stop (no memory).
  template <class Sink>
  bool next(std::string& id, Sink&& sink) {
    if (!ok_) return false;
    if (!pending_) {
      while (true) {
        if (!in_.getline(lp_, ll_)) return false;
        if (fastq_ ? ll_ != 0 : (ll_ != 0 && lp_[0] == '>')) break;
      }
    }
    pending_ = false;
    if (fastq_) {
      if (lp_[0] != '@') { ok_ = false; return false; }
      id.assign(lp_ + 1, ll_ - 1);
      size_t n = 0;
      while (in_.getline(lp_, ll_)) { if (ll_ != 0 && lp_[0] == '+') break; if (ll_ && !sink(lp_, ll_)) { ok_ = false; return false; } n += ll_; }
      size_t got = 0;
      while (got < n && in_.getline(lp_, ll_)) got += ll_;
      return true;
    }
    id.assign(lp_ + 1, ll_ - 1);
    while (in_.getline(lp_, ll_)) {
      if (ll_ != 0 && lp_[0] == '>') { pending_ = true; break; }
      if (ll_ && !sink(lp_, ll_)) { ok_ = false; return false; }
    }
    return true;
  }
  bool next(std::string& id, std::string& seq) {
    seq.clear();
    return next(id, [&](const char* p, size_t n) { seq.append(p, n); return true; });
  }

 private:
  LineReader in_;
  const char* lp_ = nullptr;
  size_t ll_ = 0;
  bool ok_ = true, fastq_ = false, pending_ = false;
};

// One batch of reads on its way through the pipeline: read -> corrected on a device -> written, in input order.
// a growable page-locked host buffer (talc_pinned_alloc): the reads of a batch are parsed straight into it and the
// corrected records come back into another one, so both directions are DMA transfers that run beside the kernels of
// the GPU's other worker; each worker keeps its two buffers for the whole run
struct PinnedBuf {
  char* p = nullptr;
  size_t cap = 0, len = 0;
  bool pinned = false;
  void release() { if (pinned) talc_pinned_free(p); else free(p); p = nullptr; }
  ~PinnedBuf() { release(); }
  bool reserve(size_t n) {
    if (n <= cap) return true;
    size_t nc = std::max<size_t>(n, std::max<size_t>(cap * 2, 1u << 20));
    bool pin = true;
    char* q = (char*)talc_pinned_alloc(nc);
    if (!q) { q = (char*)malloc(nc); pin = false; }   // (no GPU / no page-locked memory left: pageable works too, only slower)
    if (!q) return false;
    if (len) memcpy(q, p, len);
    release();
    p = q; cap = nc; pinned = pin;
    return true;
  }
  bool append(const char* s, size_t n) {
    if (!reserve(len + n)) return false;
    memcpy(p + len, s, n);
    len += n;
    return true;
  }
  bool append(const std::string& s) { return append(s.data(), s.size()); }
};

struct Chunk {
  uint64_t index = 0;
  std::vector<std::string> ids;
  std::vector<uint64_t> offsets{0};     // into the chunk's input buffer
  int inBuf = -1;                       // which page-locked input buffer of the pool holds the reads
  std::vector<int32_t> status;
  std::vector<int64_t> stats;           // 5 per read (--read-stats)
  std::string text, logText, statsText; // what the writer appends to <o>.fa / <o>.log / <o>.stats_basics.txt
};

// a Jellyfish 2 count file starts with nine digits (the header's length) and the header's opening brace (talc_jf.h)
bool isJfFile(const std::string& path) {
  std::ifstream f(path, std::ios::binary);
  char h[10];
  if (!f.read(h, 10)) return false;
  for (int i = 0; i < 9; ++i)
    if (h[i] < '0' || h[i] > '9') return false;
  return h[9] == '{';
}

// `DIR/jellyfish dump -c [-L minCount] -o out file.jf` as a child process (no shell): the program the reference's
// jellyfish2 mode names (io_pathToJF + "/" + "jellyfish", Jellyfish.cpp:340), asked once for the whole table instead of
// once per look-up.  Runs before anything touches the GPU.
extern "C" char** environ;
bool jellyfishDump(const std::string& dir, const std::string& jf, uint32_t minCount, const std::string& out, std::string& why) {
  const std::string tool = dir + "/jellyfish", lower = std::to_string(minCount);
  std::vector<const char*> av = {tool.c_str(), "dump", "-c"};
  if (minCount > 1) { av.push_back("-L"); av.push_back(lower.c_str()); }
  av.push_back("-o"); av.push_back(out.c_str());
  av.push_back(jf.c_str());
  av.push_back(nullptr);
  pid_t pid = 0;
  const int rc = posix_spawn(&pid, tool.c_str(), nullptr, nullptr, const_cast<char* const*>(av.data()), environ);
  if (rc != 0) { why = "cannot run " + tool + ": " + strerror(rc); return false; }
  int status = 0;
  while (waitpid(pid, &status, 0) < 0)
    if (errno != EINTR) { why = "waitpid failed for " + tool; return false; }
  if (!WIFEXITED(status) || WEXITSTATUS(status) != 0) {
    why = tool + " dump " + jf + " ended with " + (WIFEXITED(status) ? "exit code " + std::to_string(WEXITSTATUS(status)) : std::string("a signal"));
    return false;
  }
  return true;
}

// number of records of a FASTA file = lines that start with '>' (what SeqReader::next would return one by one), counted
// over 4 MB blocks without building a string per line; -1: cannot open
long long countFastaRecords(const std::string& file) {
  FILE* f = fopen(file.c_str(), "rb");
  if (!f) return -1;
  std::vector<char> buf(4u << 20);
  long long n = 0;
  bool atLineStart = true;
  size_t got;
  while ((got = fread(buf.data(), 1, buf.size(), f)) > 0) {
    const char* p = buf.data();
    const char* end = p + got;
    while (p < end) {
      if (atLineStart) { if (*p == '>') ++n; atLineStart = false; }
      const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
      if (!nl) break;
      p = nl + 1;
      atLineStart = true;
    }
  }
  fclose(f);
  return n;
}

double secs(std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
  return std::chrono::duration<double>(b - a).count();
}

}  // namespace

int main(int argc, const char** argv) {
  std::cout << "******************************************************\n"
            << "* TALC : Transcriptome-Aware Long Read Correction    *\n"
            << "*----------------------------------------------------*\n"
            << "*                                                    *\n"
            << "* Kmers are assumed directional                      *\n"
            << "******************************************************" << std::endl;
  std::cout << "[TALC]: Parsing arguments" << std::endl;
  Options o = parse(argc, argv);
  const std::string outFile = o.outPrefix + ".fa", statFile = o.outPrefix + ".stats_basics.txt", logFile = o.outPrefix + ".log";
  outputConfig(o, statFile);          // Settings.cpp:122
  setBasicReadStatsHeader(statFile);  // main.cpp:204

  auto t0 = std::chrono::steady_clock::now();
  std::cout << "[TALC]: Attempting to load sequences." << std::endl;
  // first pass: format check and record count only (the reads themselves stream through in batches below)
  uint64_t nReadsTotal = 0;
  {
    SeqReader probe(o.seqFile);
    std::string id, seq;
    if (probe.ok() && !probe.fastq()) {   // FASTA: the records are the lines that start with '>'
      const long long n = countFastaRecords(o.seqFile);
      if (n >= 0) nReadsTotal = (uint64_t)n;
    } else if (probe.ok()) {
      while (probe.next(id, seq)) ++nReadsTotal;
    }
    if (!probe.ok()) {  // main.cpp:219,323: prints and falls off main
      std::cout << "[TALC]: ISSUE WITH INPUT FILES" << std::endl;
      return 0;
    }
  }
  std::cout << "[TALC]: Hmm...it seems the sequence file is OK." << std::endl;
  std::cout << "[TALC]: " << nReadsTotal << " long read(s) loaded" << std::endl;
  auto t1 = std::chrono::steady_clock::now();

  // -qm jellyfish2 (the reference: no table, one `jellyfish query` child per look-up — and never reached):
  // the same counts as ONE table, from the tool's dump when -jf2 names it, else from the .jf itself; memory mode
  // takes a .jf too.  A jellyfish2 run with neither keeps the reference's behaviour (below).
  bool buildTable = (o.queryMode == "memory");
  std::vector<std::string> tmpFiles;
  if (o.queryMode == "jellyfish2" && !o.jf2.empty()) {
    std::string why;
    const std::string sr = o.outPrefix + ".SRCounts.dump.tmp", jn = o.outPrefix + ".junctions.dump.tmp";
    std::cout << "[TALC]: jellyfish2 mode: " << o.jf2 << "/jellyfish dump of " << o.dump << std::endl;
    tmpFiles.push_back(sr);
    bool ok = jellyfishDump(o.jf2, o.dump, o.p.min_count, sr, why);
    if (ok && o.useJ) { tmpFiles.push_back(jn); ok = jellyfishDump(o.jf2, o.jdump, 0, jn, why); }
    if (!ok) {
      for (const auto& f : tmpFiles) std::remove(f.c_str());
      std::cerr << "talc: " << why << "\n";
      return 2;
    }
    o.dump = sr;
    if (o.useJ) o.jdump = jn;
    buildTable = true;
  } else if (o.queryMode == "jellyfish2" && isJfFile(o.dump)) {
    std::cout << "[TALC]: jellyfish2 mode: reading " << o.dump << " natively" << std::endl;
    buildTable = true;
  }

  talc_table* table = nullptr;
  uint64_t tableSize = 0;
  if (buildTable) {  // main.cpp:224-238
    if (o.useJ) std::cout << "[TALC]: Building the SR-cdBG from count files: " << o.dump << " and " << o.jdump << std::endl;
    else std::cout << "[TALC]: Building the SR-dBG from count file: " << o.dump << std::endl;
    int64_t st[3] = {0, 0, 0};
    // the insert loop of buildCDBG runs on the first GPU when there is one (same table, ~10x faster on a 50 M dump)
    int rc = (talc_device_count() > 0)
                 ? talc_table_build_device(o.dump.c_str(), o.useJ ? o.jdump.c_str() : nullptr, &o.p, 0, &table, st)
                 : talc_table_build(o.dump.c_str(), o.useJ ? o.jdump.c_str() : nullptr, &o.p, &table, st);
    for (const auto& f : tmpFiles) std::remove(f.c_str());
    if (rc != TALC_OK) {
      // an unreadable dump leaves the reference with an empty map (Jellyfish.cpp:249-251); anything else is fatal
      std::cerr << "talc: " << talc_last_error() << "\n";
      if (rc != TALC_ERR_IO) return 2;
    } else {
      tableSize = talc_table_size(table);
      std::cout << "There were " << st[0] << " k-mers retrieved from database." << std::endl;
      std::cout << "In the whole, we have kept " << st[1] << "k-mers, whose counts were over the specified threshold." << std::endl;
    }
    std::cout << "[TALC]: SR-dBG contains " << tableSize << " nodes." << std::endl;
  }
  auto t2 = std::chrono::steady_clock::now();
  // main.cpp:240: with -qm jellyfish2 the reference runs the loop on an EMPTY map (the jellyfish2 code
  // path is dead): every read longer than K logs "No solid kmer could be found."  Kept when the run names
  // neither the jellyfish program nor a .jf file.
  const bool emptyRun = !buildTable;
  if (!emptyRun && tableSize == 0) {
    std::cout << "[TALC]: The de Bruijn Graph is empty...Correction aborted." << std::endl;  // main.cpp:319-320
    return 1;
  }
  std::cout << "[TALC]: Good news, there are nodes in the de Bruijn Graph." << std::endl;
  std::cout << "[TALC]: Maybe we can try and correct some long reads, then?" << std::endl;

  int ndev = 0, nphys = 0;
  if (!emptyRun) {
    nphys = talc_device_count();
    if (nphys <= 0) { std::cerr << "talc: no MI355X / HIP device visible; the correction path has no CPU fallback\n"; return 2; }
    ndev = nphys;
    // TALC_FAKE_GPUS=n (a rehearsal hook for one-GPU boxes): the sharder runs as if n GPUs were present, logical GPU d on
    // physical device d mod the real count — the same worker threads, contexts, dealing and ordered merge
    if (const char* fk = getenv("TALC_FAKE_GPUS")) if (atoi(fk) > 0) ndev = atoi(fk);
    if (o.gpus > 0) ndev = std::min(ndev, o.gpus);
    for (int d = 0; d < std::min(ndev, nphys); ++d)
      if (talc_table_upload(table, d) != TALC_OK) { std::cerr << "talc: device error: " << talc_last_error() << "\n"; talc_table_destroy(table); return 2; }
    std::cout << "[TALC]: correcting on " << ndev << " GPU(s); k-mer table replicated (" << talc_table_device_bytes(table) / 1e9 << " GB each)" << std::endl;
    if (!o.haveBatchReads) {
      // at least two batches per worker (two workers per GPU) so that every worker's transfers find kernels to hide
      // under and the last batches end together; not below 20000 reads (a small batch leaves k_search's 5120 waves a
      // long tail), not above 200000
      const uint64_t per = (nReadsTotal + (uint64_t)ndev * 4 - 1) / ((uint64_t)ndev * 4);
      o.batchReads = (uint32_t)std::min<uint64_t>(200000, std::max<uint64_t>(20000, per));
    }
  }
  auto t2b = std::chrono::steady_clock::now();
  std::cout << "Specified output file name: " << outFile << std::endl;
  std::ofstream of(outFile, std::ios_base::trunc);
  if (!of) { std::cerr << "ERROR: Could not open the file " << outFile << "\n"; return 2; }

  // ---- the pipeline (replaces the load-everything / OpenMP loop / write-everything of main.cpp:209-310):
  //   one READER thread parses batches of --batch-reads reads straight into page-locked buffers of a small pool (input order =
  //   batch index); two WORKERS per GPU (own context and stream each: one batch's transfers run under the other's kernels)
  //   take a batch, correct it, and turn the records into the text of <o>.fa themselves — '>' id, 70-column lines
  //   (io.cpp:50-75) — so that formatting runs in parallel; one WRITER thread appends the finished batches' text strictly
  //   in input order.  At most (workers + 2) batches are in flight.
  SeqReader reader(o.seqFile);
  std::atomic<bool> failed{false};
  uint64_t nextIndex = 0, nextToWrite = 0, basesTotal = 0;
  std::ofstream lf, sf;
  std::mutex failMu;
  std::string failMsg;
  std::atomic<uint64_t> readErrors{0};
  const int nWorkers = emptyRun ? 1 : 2 * ndev;
  const int nInBufs = nWorkers + 2;
  std::vector<PinnedBuf> inBufs(nInBufs);
  std::mutex qMu;                                   // guards everything below
  std::condition_variable cvFree, cvReady, cvDone, cvRoom;
  std::vector<int> freeIn;
  for (int i = 0; i < nInBufs; ++i) freeIn.push_back(i);
  std::deque<std::unique_ptr<Chunk>> ready;
  bool readerDone = false;
  std::map<uint64_t, std::unique_ptr<Chunk>> finished;
  int workersLeft = nWorkers;
  auto setFailed = [&]() {
    { std::lock_guard<std::mutex> g(failMu); if (!failed) failMsg = talc_last_error(); failed = true; }
    std::lock_guard<std::mutex> g(qMu);
    cvFree.notify_all(); cvReady.notify_all(); cvDone.notify_all(); cvRoom.notify_all();
  };
  double readBusy = 0, writeBusy = 0;            // the reader's / the writer's own time
  std::atomic<long long> deviceBusyUs{0};        // summed over the workers
  std::atomic<long long> ctxUs{0}, createUs{0}, correctUs{0}, fetchUs{0}, unpackUs{0}, waitChunkUs{0};   // ... and its parts
  auto usSince = [](std::chrono::steady_clock::time_point t) { return (long long)(1e6 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count()); };
  // what a batch's reads take, from the file's size and the scan's read count: the page-locked buffers are allocated once,
  // at that size, by as many threads as there are buffers (an allocation of a few hundred MB takes tens of milliseconds)
  uint64_t batchBytesEstimate = 0;
  if (!emptyRun && nReadsTotal > 0) {
    struct stat sbq;
    if (stat(o.seqFile.c_str(), &sbq) == 0) batchBytesEstimate = (uint64_t)((double)sbq.st_size / (double)nReadsTotal * (double)std::min<uint64_t>(o.batchReads, nReadsTotal) * (reader.fastq() ? 0.55 : 1.05)) + (1u << 20);
    std::vector<std::thread> th;
    for (int i = 0; i < nInBufs; ++i) th.emplace_back([&, i] { inBufs[i].reserve(batchBytesEstimate); });
    for (auto& t : th) t.join();
  }
  auto readerThread = [&]() {
    std::string id;
    while (!failed) {
      int bi = -1;
      {
        std::unique_lock<std::mutex> g(qMu);
        cvFree.wait(g, [&] { return failed.load() || !freeIn.empty(); });
        if (failed) break;
        bi = freeIn.back(); freeIn.pop_back();
      }
      const auto tr0 = std::chrono::steady_clock::now();
      std::unique_ptr<Chunk> c(new Chunk());
      PinnedBuf& in = inBufs[bi];
      in.len = 0;
      c->inBuf = bi;
      bool bad = false;
      while (c->ids.size() < o.batchReads && reader.next(id, [&](const char* q, size_t m) { if (!in.append(q, m)) { bad = true; return false; } return true; })) {
        c->ids.push_back(id);
        c->offsets.push_back(in.len);
      }
      readBusy += std::chrono::duration<double>(std::chrono::steady_clock::now() - tr0).count();
      if (bad) { setFailed(); break; }
      if (c->ids.empty()) { std::lock_guard<std::mutex> g(qMu); freeIn.push_back(bi); break; }
      c->index = nextIndex++;
      basesTotal += in.len;
      std::lock_guard<std::mutex> g(qMu);
      ready.push_back(std::move(c));
      cvReady.notify_one();
    }
    std::lock_guard<std::mutex> g(qMu);
    readerDone = true;
    cvReady.notify_all();
  };
  // the text of one batch: '>' id, the sequence wrapped at 70 columns; the log and stats lines of its reads
  auto formatChunk = [&](Chunk& k, const char* recs, const uint64_t* oo) {
    const size_t n = k.ids.size();
    size_t need = 0;
    for (size_t r = 0; r < n; ++r) { const size_t L = (size_t)(oo[r + 1] - oo[r]); need += k.ids[r].size() + 2 + L + (L + 69) / 70; }
    k.text.resize(need);
    char* w = &k.text[0];
    for (size_t r = 0; r < n; ++r) {
      // (a read that exhausted the device scratch is written through unchanged, like any read the reference fails on:
      //  it logs and goes on, main.cpp:298-303)
      const char* msg = k.status[r] == TALC_READ_NO_STRUCTURE ? "Unable to define convenient structure."       // main.cpp:290
                        : k.status[r] == TALC_READ_NO_SOLID_KMER ? "No solid kmer could be found."             // main.cpp:294
                        : k.status[r] == TALC_READ_ERROR ? "Device scratch exhausted; read left uncorrected." : nullptr;
      if (msg) { k.logText += "[Read: "; k.logText += k.ids[r]; k.logText += " ]: "; k.logText += msg; k.logText += '\n'; }
      if (o.readStats && k.stats.size() == 5 * n && k.stats[5 * r]) {   // Read.cpp:425-431
        k.statsText += "\n" + k.ids[r] + "\t" + std::to_string(k.stats[5 * r + 1]) + "\t" + std::to_string(k.stats[5 * r + 2]) + "\t" +
                       std::to_string(k.stats[5 * r + 3]) + "\t" + std::to_string(k.stats[5 * r + 4]);
      }
      *w++ = '>';
      memcpy(w, k.ids[r].data(), k.ids[r].size()); w += k.ids[r].size();
      *w++ = '\n';
      const char* q = recs + oo[r];
      const size_t L = (size_t)(oo[r + 1] - oo[r]);
      for (size_t p = 0; p < L; p += 70) { const size_t m = std::min<size_t>(70, L - p); memcpy(w, q + p, m); w += m; *w++ = '\n'; }
    }
    k.text.resize((size_t)(w - k.text.data()));
  };
  auto passThrough = [&](Chunk& c, const PinnedBuf& in) {
    // no table at all: pass-through with the reference's statuses (Dna5 conversion / -rev still apply)
    const size_t n = c.ids.size();
    c.status.assign(n, TALC_READ_SKIPPED_SHORT);
    std::string all;
    std::vector<uint64_t> oo(n + 1, 0);
    for (size_t r = 0; r < n; ++r) {
      std::string q(in.p + c.offsets[r], c.offsets[r + 1] - c.offsets[r]);
      for (auto& ch : q) { ch = (ch == 'a' || ch == 'A') ? 'A' : (ch == 'c' || ch == 'C') ? 'C' : (ch == 'g' || ch == 'G') ? 'G' : (ch == 't' || ch == 'T') ? 'T' : 'N'; }
      if (o.p.reverse) {
        std::string rcs(q.size(), 'N');
        for (size_t i = 0; i < q.size(); ++i) { char ch = q[q.size() - 1 - i]; rcs[i] = ch == 'A' ? 'T' : ch == 'C' ? 'G' : ch == 'G' ? 'C' : ch == 'T' ? 'A' : 'N'; }
        q = rcs;
      }
      c.status[r] = q.size() > o.p.k ? TALC_READ_NO_SOLID_KMER : TALC_READ_SKIPPED_SHORT;
      if (o.readStats) { c.stats.resize(5 * n, 0); if (q.size() > o.p.k) { c.stats[5 * r] = 1; c.stats[5 * r + 1] = (int64_t)q.size(); } }
      all += q;
      oo[r + 1] = all.size();
    }
    formatChunk(c, all.data(), oo.data());
  };
  auto worker = [&](int device) {
    talc_ctx* ctx = nullptr;
    { const auto tc0 = std::chrono::steady_clock::now();
      if (device >= 0 && talc_ctx_create(table, &o.p, device, &ctx) != TALC_OK) { setFailed(); }
      ctxUs += usSince(tc0); }
    PinnedBuf outb;
    if (device >= 0 && batchBytesEstimate) outb.reserve(batchBytesEstimate + batchBytesEstimate / 16);   // (one page-locked allocation, not a dozen doublings)
    while (!failed) {
      const auto tw0 = std::chrono::steady_clock::now();
      std::unique_ptr<Chunk> c;
      {
        std::unique_lock<std::mutex> g(qMu);
        cvReady.wait(g, [&] { return failed.load() || !ready.empty() || readerDone; });
        if (failed || ready.empty()) break;
        c = std::move(ready.front()); ready.pop_front();
      }
      waitChunkUs += usSince(tw0);
      PinnedBuf& in = inBufs[c->inBuf];
      if (device < 0) {
        passThrough(*c, in);
      } else {
        const uint32_t n = (uint32_t)c->ids.size();
        c->status.assign(n, TALC_READ_SKIPPED_SHORT);
        talc_batch* b = nullptr;
        const auto td0 = std::chrono::steady_clock::now();
        struct Acc { std::atomic<long long>& a; std::chrono::steady_clock::time_point t; ~Acc() { a += (long long)(1e6 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count()); } } acc{deviceBusyUs, td0};
        if (talc_batch_create(ctx, in.p, c->offsets.data(), n, &b) != TALC_OK) { setFailed(); break; }
        createUs += usSince(td0);
        const auto tk0 = std::chrono::steady_clock::now();
        // < 0: a real HIP / argument error stops the run; TALC_WARN_READ_ERRORS (> 0) is a complete batch in which some
        // reads kept their input sequence (status TALC_READ_ERROR -> a .log line), and the run goes on
        const int crc = talc_batch_correct(ctx, b);
        if (crc < 0) { setFailed(); talc_batch_destroy(b); break; }
        correctUs += usSince(tk0);
        const auto tf0 = std::chrono::steady_clock::now();
        const uint64_t total = talc_batch_corrected_bytes(b);
        std::vector<uint64_t> oo(n + 1);
        if (!outb.reserve(std::max<uint64_t>(total, 1))) { setFailed(); talc_batch_destroy(b); break; }
        if (talc_batch_fetch_corrected(ctx, b, outb.p, total, oo.data(), c->status.data()) != TALC_OK) { setFailed(); talc_batch_destroy(b); break; }
        if (o.readStats) { c->stats.resize(5ull * n); if (talc_batch_fetch_read_stats(ctx, b, c->stats.data()) != TALC_OK) { setFailed(); talc_batch_destroy(b); break; } }
        fetchUs += usSince(tf0);
        const auto tu0 = std::chrono::steady_clock::now();
        formatChunk(*c, outb.p, oo.data());
        unpackUs += usSince(tu0);
        if (const char* tv = getenv("TALC_TIMING")) if (tv[0] == '2') {   // per batch: where this worker's time went
          talc_timing tm; talc_ctx_get_timing(ctx, &tm);
          fprintf(stderr, "[talc-batch] reads %u: create+H2D %.3f s, correct %.3f s (kernels: coverage %.1f structure %.1f search %.1f retry %.1f ms), fetch %.3f s, text %.3f s\n",
                  n, std::chrono::duration<double>(tk0 - td0).count(), std::chrono::duration<double>(tf0 - tk0).count(), tm.coverage_ms, tm.structure_ms, tm.search_ms, tm.retry_ms,
                  std::chrono::duration<double>(tu0 - tf0).count(), usSince(tu0) / 1e6);
        }
        if (crc > 0) for (uint32_t i = 0; i < n; ++i) readErrors += c->status[i] == TALC_READ_ERROR ? 1 : 0;
        talc_batch_destroy(b);
      }
      std::unique_lock<std::mutex> g(qMu);
      freeIn.push_back(c->inBuf); c->inBuf = -1;
      cvFree.notify_one();
      // (the batch the writer waits for always gets in; the others wait while the writer is more than a few batches behind)
      cvRoom.wait(g, [&] { return failed.load() || c->index == nextToWrite || finished.size() < (size_t)nWorkers + 2; });
      finished[c->index] = std::move(c);
      cvDone.notify_one();
    }
    if (ctx) talc_ctx_destroy(ctx);
    std::lock_guard<std::mutex> g(qMu);
    --workersLeft;
    cvDone.notify_all();
  };
  auto writerThread = [&]() {   // io.cpp:50-75 + SeqFileOut FASTA writer, io.cpp:105-111 log lines
    while (true) {
      std::unique_ptr<Chunk> c;
      {
        std::unique_lock<std::mutex> g(qMu);
        cvDone.wait(g, [&] { return (!finished.empty() && finished.begin()->first == nextToWrite) || workersLeft == 0 || failed.load(); });
        if (!finished.empty() && finished.begin()->first == nextToWrite) { c = std::move(finished.begin()->second); finished.erase(finished.begin()); }
        else if (workersLeft == 0 || failed) break;
      }
      if (!c) continue;
      const auto tw0 = std::chrono::steady_clock::now();
      if (!c->logText.empty()) { if (!lf.is_open()) lf.open(logFile, std::ios_base::app); lf << c->logText; lf.flush(); }
      if (!c->statsText.empty()) { if (!sf.is_open()) sf.open(statFile, std::ios_base::app); sf << c->statsText; }
      of.write(c->text.data(), (std::streamsize)c->text.size());
      writeBusy += std::chrono::duration<double>(std::chrono::steady_clock::now() - tw0).count();
      std::lock_guard<std::mutex> g(qMu);
      ++nextToWrite;
      cvRoom.notify_all();
    }
  };
  {
    std::thread rd(readerThread), wr(writerThread);
    std::vector<std::thread> workers;
    if (emptyRun) workers.emplace_back(worker, -1);
    else for (int d = 0; d < ndev; ++d) for (int w = 0; w < 2; ++w) workers.emplace_back(worker, d % nphys);
    for (auto& w : workers) w.join();
    rd.join();
    wr.join();
  }
  of.close();
  auto t3 = std::chrono::steady_clock::now();
  if (failed) {
    std::cerr << "talc: device error: " << failMsg << "\n";
    if (table) talc_table_destroy(table);
    return 2;
  }
  if (table) talc_table_destroy(table);
  if (readErrors) std::cerr << "talc: " << readErrors << " read(s) exhausted the device scratch and were written uncorrected (see " << logFile << ")\n";
  std::cout << "[TALC]: Looks like we are done now." << std::endl;
  // the reference's own split (main.cpp:213-236: loading the reads, building the graph; :311-313: the correction), in
  // wall-clock seconds instead of CPU minutes.  The correction phase is a pipeline: its three busy times overlap.
  fprintf(stderr, "[talc] scan=%.3fs table=%.3fs upload=%.3fs read+correct+write=%.3fs (%.3g bases/s, %llu batches of <= %u reads) total=%.3fs\n",
          secs(t0, t1), secs(t1, t2), secs(t2, t2b), secs(t2b, t3), secs(t2b, t3) > 0 ? (double)basesTotal / secs(t2b, t3) : 0.0,
          (unsigned long long)nextIndex, o.batchReads, secs(t0, t3));
  fprintf(stderr, "[talc-timing] {\"scan_s\": %.4f, \"table_parse_build_s\": %.4f, \"upload_s\": %.4f, \"correct_phase_s\": %.4f, "
                  "\"reader_busy_s\": %.4f, \"device_busy_s_sum_over_workers\": %.4f, \"writer_busy_s\": %.4f, \"total_s\": %.4f, "
                  "\"device_parts_s\": {\"ctx_create\": %.4f, \"batch_create_h2d\": %.4f, \"correct\": %.4f, \"fetch_d2h\": %.4f, \"records_to_text\": %.4f, \"waiting_for_reader\": %.4f}, "
                  "\"reads\": %llu, \"bases\": %llu, \"batches\": %llu, \"batch_reads\": %u, \"gpus\": %d, \"workers\": %d}\n",
          secs(t0, t1), secs(t1, t2), secs(t2, t2b), secs(t2b, t3), readBusy, (double)deviceBusyUs.load() / 1e6, writeBusy, secs(t0, t3),
          ctxUs.load() / 1e6, createUs.load() / 1e6, correctUs.load() / 1e6, fetchUs.load() / 1e6, unpackUs.load() / 1e6, waitChunkUs.load() / 1e6,
          (unsigned long long)nReadsTotal, (unsigned long long)basesTotal, (unsigned long long)nextIndex, o.batchReads, ndev, emptyRun ? 1 : 2 * ndev);
  return 0;
}
