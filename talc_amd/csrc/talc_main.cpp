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
//   --batch-reads N  reads per device batch (default 200000)
//   -k accepts 18..31 (the reference stops at 30, main.cpp:115-116; 31 still fits 62 bits)
// -t/--num_threads is accepted and ignored (the parallelism is on the device).
// Differences, all documented in INTEGRATION.md: stdout carries the [TALC] banners but none of
// the reference's always-on debug dumps; log lines are written in input order.
#include <omp.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <thread>
#include <vector>

#include "talc_hip.h"

namespace {

struct Options {
  std::string seqFile, outPrefix = "out", queryMode = "memory", dump, jdump;
  talc_params p;
  bool haveK = false, haveSR = false, useJ = false;
  int gpus = -1;
  int nthreads = 1;
  uint32_t batchReads = 200000;
};

void usage(FILE* f) {
  fprintf(f,
          "TALC: Transcriptome-Aware Long Read Correction (MI355X hot path)\n"
          "SYNOPSIS  talc [OPTIONS] <long reads .fa/.fq> -k K -SR <jellyfish dump>\n"
          "  -o, --output TEXT           prefix of the output files (default: out)\n"
          "  -k, --kmerSize INT          k-mer length, 18..31 (required)\n"
          "  -qm, --query-mode TEXT      memory | jellyfish2 (default: memory)\n"
          "  -SR, --SRCounts TEXT        short-read k-mer counts, `jellyfish dump -c` text (required)\n"
          "  -j, --junctions TEXT        k-mers flanking junctions and their counts\n"
          "  -jf2, --pathToJF2 TEXT      accepted, unused\n"
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
          "  --batch-reads INT           reads per device batch (default 200000)\n"
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
    else if (is(a, "jf2", "pathToJF2")) need(i);
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
    else if (a == "--batch-reads") { double v = num(need(i), "batch-reads"); range(v, 1, 4e9, "batch-reads"); o.batchReads = (uint32_t)v; }
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

// io.cpp:26-48 with SeqFileIn::readRecords semantics: FASTA ('>') or FASTQ ('@') decided by the first
// record; id = the whole header line after the marker; multi-line sequences concatenated; qualities dropped.
// Sequences are kept as raw text: the device applies the Dna5 conversion.
bool loadSeqData(const std::string& file, std::vector<std::string>& ids, std::string& bases, std::vector<uint64_t>& offsets) {
  std::ifstream in(file);
  if (!in) { std::cerr << "ERROR: Could not open file " << file << "\n"; return false; }
  auto chomp = [](std::string& l) { while (!l.empty() && (l.back() == '\r' || l.back() == '\n')) l.pop_back(); };
  std::string line;
  bool started = false, fastq = false, have = false;
  offsets.assign(1, 0);
  while (std::getline(in, line)) {
    chomp(line);
    if (!started) {
      if (line.empty()) continue;
      started = true;
      fastq = line[0] == '@';
      if (!fastq && line[0] != '>') return false;
    }
    if (fastq) {
      if (line.empty()) continue;
      if (line[0] != '@') return false;
      ids.push_back(line.substr(1));
      const size_t start = bases.size();
      while (std::getline(in, line)) { chomp(line); if (!line.empty() && line[0] == '+') break; bases += line; }
      const size_t want = bases.size() - start;
      size_t got = 0;
      while (got < want && std::getline(in, line)) { chomp(line); got += line.size(); }
      offsets.push_back(bases.size());
    } else if (!line.empty() && line[0] == '>') {
      if (have) offsets.push_back(bases.size());
      ids.push_back(line.substr(1));
      have = true;
    } else if (have) {
      bases += line;
    }
  }
  if (!fastq && have) offsets.push_back(bases.size());
  return true;
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
  std::vector<std::string> ids;
  std::string bases;
  std::vector<uint64_t> offsets;
  if (!loadSeqData(o.seqFile, ids, bases, offsets)) {  // main.cpp:219,323: prints and falls off main
    std::cout << "[TALC]: ISSUE WITH INPUT FILES" << std::endl;
    return 0;
  }
  std::cout << "[TALC]: Hmm...it seems the sequence file is OK." << std::endl;
  std::cout << "[TALC]: " << ids.size() << " long read(s) loaded" << std::endl;
  auto t1 = std::chrono::steady_clock::now();

  talc_table* table = nullptr;
  uint64_t tableSize = 0;
  if (o.queryMode == "memory") {  // main.cpp:224-238
    if (o.useJ) std::cout << "[TALC]: Building the SR-cdBG from count files: " << o.dump << " and " << o.jdump << std::endl;
    else std::cout << "[TALC]: Building the SR-dBG from count file: " << o.dump << std::endl;
    int64_t st[3] = {0, 0, 0};
    // the insert loop of buildCDBG runs on the first GPU when there is one (same table, ~10x faster on a 50 M dump)
    int rc = (talc_device_count() > 0)
                 ? talc_table_build_device(o.dump.c_str(), o.useJ ? o.jdump.c_str() : nullptr, &o.p, 0, &table, st)
                 : talc_table_build(o.dump.c_str(), o.useJ ? o.jdump.c_str() : nullptr, &o.p, &table, st);
    if (rc != TALC_OK) {
      if (rc == TALC_ERR_IO) {
        // an unreadable dump leaves the reference with an empty map (Jellyfish.cpp:249-251)
        std::cerr << "talc: " << talc_last_error() << "\n";
      } else {
        std::cerr << "talc: " << talc_last_error() << "\n";
        return 2;
      }
    } else {
      tableSize = talc_table_size(table);
      std::cout << "There were " << st[0] << " k-mers retrieved from database." << std::endl;
      std::cout << "In the whole, we have kept " << st[1] << "k-mers, whose counts were over the specified threshold." << std::endl;
    }
    std::cout << "[TALC]: SR-dBG contains " << tableSize << " nodes." << std::endl;
  }
  auto t2 = std::chrono::steady_clock::now();
  // main.cpp:240: with -qm jellyfish2 the reference runs the loop on an EMPTY map (the jellyfish2 code
  // path is dead): every read longer than K logs "No solid kmer could be found."
  const bool emptyRun = (o.queryMode == "jellyfish2");
  if (!emptyRun && tableSize == 0) {
    std::cout << "[TALC]: The de Bruijn Graph is empty...Correction aborted." << std::endl;  // main.cpp:319-320
    return 1;
  }
  std::cout << "[TALC]: Good news, there are nodes in the de Bruijn Graph." << std::endl;
  std::cout << "[TALC]: Maybe we can try and correct some long reads, then?" << std::endl;

  const uint32_t nReads = (uint32_t)ids.size();
  std::vector<std::string> outSeqs(nReads);
  std::vector<int32_t> status(nReads, TALC_READ_SKIPPED_SHORT);
  bool failed = false;
  std::string failMsg;
  if (emptyRun) {
    // no table at all: pass-through with the reference's statuses (Dna5 conversion / -rev still apply)
    for (uint32_t r = 0; r < nReads; ++r) {
      std::string s = bases.substr(offsets[r], offsets[r + 1] - offsets[r]);
      for (auto& c : s) { c = (c == 'a' || c == 'A') ? 'A' : (c == 'c' || c == 'C') ? 'C' : (c == 'g' || c == 'G') ? 'G' : (c == 't' || c == 'T') ? 'T' : 'N'; }
      if (o.p.reverse) {
        std::string rcs(s.size(), 'N');
        for (size_t i = 0; i < s.size(); ++i) { char c = s[s.size() - 1 - i]; rcs[i] = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : 'N'; }
        s = rcs;
      }
      status[r] = s.size() > o.p.k ? TALC_READ_NO_SOLID_KMER : TALC_READ_SKIPPED_SHORT;
      outSeqs[r] = s;
    }
  } else {
    int ndev = talc_device_count();
    if (ndev <= 0) { std::cerr << "talc: no MI355X / HIP device visible; the correction path has no CPU fallback\n"; return 2; }
    if (o.gpus > 0) ndev = std::min(ndev, o.gpus);
    ndev = (int)std::max<uint32_t>(1, std::min<uint32_t>((uint32_t)ndev, std::max<uint32_t>(nReads, 1)));
    // contiguous blocks of reads balanced by bases
    std::vector<uint32_t> bounds(ndev + 1, nReads);
    bounds[0] = 0;
    {
      const uint64_t total = offsets[nReads];
      uint32_t r = 0;
      for (int d = 1; d < ndev; ++d) {
        const uint64_t target = total / ndev * d;
        while (r < nReads && offsets[r] < target) ++r;
        bounds[d] = r;
      }
    }
    std::cout << "[TALC]: correcting on " << ndev << " GPU(s); k-mer table replicated (" << talc_table_device_bytes(table) / 1e9 << " GB each)" << std::endl;
    for (int d = 0; d < ndev && !failed; ++d)
      if (talc_table_upload(table, d) != TALC_OK) { failed = true; failMsg = talc_last_error(); }
    std::vector<std::thread> workers;
    std::vector<std::string> errs(ndev);
    for (int d = 0; d < ndev && !failed; ++d) {
      workers.emplace_back([&, d]() {
        talc_ctx* ctx = nullptr;
        if (talc_ctx_create(table, &o.p, d, &ctx) != TALC_OK) { errs[d] = talc_last_error(); return; }
        for (uint32_t lo = bounds[d]; lo < bounds[d + 1]; lo += o.batchReads) {
          const uint32_t hi = std::min<uint32_t>(bounds[d + 1], lo + o.batchReads);
          const uint32_t n = hi - lo;
          std::vector<uint64_t> boffs(n + 1);
          for (uint32_t i = 0; i <= n; ++i) boffs[i] = offsets[lo + i] - offsets[lo];
          talc_batch* b = nullptr;
          if (talc_batch_create(ctx, bases.data() + offsets[lo], boffs.data(), n, &b) != TALC_OK) { errs[d] = talc_last_error(); break; }
          int rc = talc_batch_correct(ctx, b);
          if (rc != TALC_OK) errs[d] = talc_last_error();
          const uint64_t total = talc_batch_corrected_bytes(b);
          std::vector<char> buf(std::max<uint64_t>(total, 1));
          std::vector<uint64_t> oo(n + 1);
          if (talc_batch_fetch_corrected(ctx, b, buf.data(), total, oo.data(), status.data() + lo) != TALC_OK) { errs[d] = talc_last_error(); talc_batch_destroy(b); break; }
          for (uint32_t i = 0; i < n; ++i) outSeqs[lo + i].assign(buf.data() + oo[i], oo[i + 1] - oo[i]);
          talc_batch_destroy(b);
        }
        talc_ctx_destroy(ctx);
      });
    }
    for (auto& w : workers) w.join();
    for (auto& e : errs) if (!e.empty()) { failed = true; failMsg = e; }
  }
  auto t3 = std::chrono::steady_clock::now();
  if (failed) {
    std::cerr << "talc: device error: " << failMsg << "\n";
    if (table) talc_table_destroy(table);
    return 2;
  }
  // log lines (io.cpp:105-111, appended), in input order
  {
    std::ofstream lf;
    for (uint32_t r = 0; r < nReads; ++r) {
      const char* msg = status[r] == TALC_READ_NO_STRUCTURE ? "Unable to define convenient structure."    // main.cpp:290
                        : status[r] == TALC_READ_NO_SOLID_KMER ? "No solid kmer could be found." : nullptr;  // main.cpp:294
      if (!msg) continue;
      if (!lf.is_open()) lf.open(logFile, std::ios_base::app);
      lf << "[Read: " << ids[r] << " ]: " << msg << std::endl;
    }
  }
  // io.cpp:50-75 + SeqFileOut FASTA writer: '>' id, sequence wrapped at 70 columns
  std::cout << "Specified output file name: " << outFile << std::endl;
  {
    std::ofstream of(outFile, std::ios_base::trunc);
    if (!of) { std::cerr << "ERROR: Could not open the file " << outFile << "\n"; return 2; }
    std::string chunk;
    for (uint32_t r = 0; r < nReads; ++r) {
      of << '>' << ids[r] << '\n';
      const std::string& s = outSeqs[r];
      for (size_t p = 0; p < s.size(); p += 70) { of.write(s.data() + p, (std::streamsize)std::min<size_t>(70, s.size() - p)); of.put('\n'); }
    }
  }
  auto t4 = std::chrono::steady_clock::now();
  if (table) talc_table_destroy(table);
  std::cout << "[TALC]: Looks like we are done now." << std::endl;
  double nb = (double)offsets[nReads];
  fprintf(stderr, "[talc] load=%.3fs table=%.3fs correct=%.3fs (%.3g bases/s) write=%.3fs total=%.3fs\n", secs(t0, t1), secs(t1, t2),
          secs(t2, t3), secs(t2, t3) > 0 ? nb / secs(t2, t3) : 0.0, secs(t3, t4), secs(t0, t4));
  return 0;
}
