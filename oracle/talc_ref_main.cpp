// talc_ref_main.cpp — CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).  parity unpinned.
// Restatement of the reference's driver: main.cpp:83-325 + Settings.cpp:74-185 (option table,
// output files, per-read OpenMP loop).  stdout banners are reduced; the four output files
// (<o>.fa, <o>.log, <o>.config.txt, <o>.stats_basics.txt) follow the reference text exactly.
// Extensions (documented): -k accepts 18..31 (reference 18..30, main.cpp:115-116);
// --table-backend map|flat selects the oracle table implementation; log lines are emitted in
// input order (the reference's order is nondeterministic for -t > 1).
#include <omp.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <string>
#include <vector>

#include "talc_oracle.hpp"

using namespace talc_oracle;

static void usage() {
  std::cerr << "talc_ref <reads.fa|fq> -k K -SR dump [-j junctions] [-o prefix] [-t threads] [-rev]\n"
               "  [--MIN_INNER_SCORE f] [--MIN_BORDER_SCORE f] [--MIN_COUNT n] [--SR_ERROR_RATE f]\n"
               "  [--WINDOW_SIZE n] [--MAX_NB_BRANCHES n] [--ALPHA_FOR_PRED f] [-qm memory|jellyfish2]\n"
               "  [--table-backend map|flat] [--read-stats]\n";
}

template <typename T>
static void printNum(std::ostream& os, T v) { os << v; }

int main(int argc, const char* argv[]) {
  Params P;
  std::string seqFile, outPrefix = "out", queryMode = "memory", dump, jdump, backend = "map";
  bool readStats = false;
  bool haveK = false, haveSR = false, useJ = false;
  int nthreads = 1;
  auto need = [&](int& i) -> const char* {
    if (i + 1 >= argc) { usage(); exit(1); }
    return argv[++i];
  };
  auto range = [&](double v, double lo, double hi, const char* name) {
    if (v < lo || v > hi) { std::cerr << "talc_ref: value out of range for " << name << "\n"; exit(1); }
  };
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    if (a == "-o" || a == "--output") outPrefix = need(i);
    else if (a == "-k" || a == "--kmerSize") { P.K = (unsigned)atoi(need(i)); haveK = true; range(P.K, 18, 31, "k"); }
    else if (a == "-qm" || a == "--query-mode") { queryMode = need(i); if (queryMode != "memory" && queryMode != "jellyfish2") { usage(); return 1; } }
    else if (a == "-SR" || a == "--SRCounts") { dump = need(i); haveSR = true; }
    else if (a == "-j" || a == "--junctions") { jdump = need(i); useJ = true; }
    else if (a == "-jf2" || a == "--pathToJF2") need(i);
    else if (a == "--MIN_INNER_SCORE" || a == "-MIN_INNER_SCORE") { P.gp_MIN_INNER_SCORE = atof(need(i)); range(P.gp_MIN_INNER_SCORE, 0.3, 0.9, "MIN_INNER_SCORE"); }
    else if (a == "--MIN_BORDER_SCORE" || a == "-MIN_BORDER_SCORE") { P.gp_MIN_BORDER_SCORE = atof(need(i)); range(P.gp_MIN_BORDER_SCORE, 0.5, 0.9, "MIN_BORDER_SCORE"); }
    else if (a == "--MIN_COUNT" || a == "-MIN_COUNT") { P.gp_MIN_COUNT = (unsigned)atoi(need(i)); range(P.gp_MIN_COUNT, 2, 1e18, "MIN_COUNT"); }
    else if (a == "--SR_ERROR_RATE" || a == "-SR_ERROR_RATE") { P.gp_SR_ERROR_RATE = atof(need(i)); range(P.gp_SR_ERROR_RATE, 0.01, 0.1, "SR_ERROR_RATE"); }
    else if (a == "--WINDOW_SIZE" || a == "-WINDOW_SIZE") { P.gp_WINDOW_SIZE = (unsigned)atoi(need(i)); range(P.gp_WINDOW_SIZE, 6, 1e18, "WINDOW_SIZE"); }
    else if (a == "--MAX_NB_BRANCHES" || a == "-MAX_NB_BRANCHES") { P.gp_MAX_NB_COMPETING_PATHS = (unsigned)atoi(need(i)); range(P.gp_MAX_NB_COMPETING_PATHS, 5, 1e18, "MAX_NB_BRANCHES"); }
    else if (a == "--ALPHA_FOR_PRED" || a == "-ALPHA_FOR_PRED") { P.gp_ALPHA = atof(need(i)); range(P.gp_ALPHA, 0.67, 1e300, "ALPHA_FOR_PRED"); }
    else if (a == "-t" || a == "--num_threads") { nthreads = atoi(need(i)); range(nthreads, 1, 1e9, "num_threads"); }
    else if (a == "--DEBUG_MODE" || a == "-DEBUG_MODE") need(i);
    else if (a == "-rev" || a == "--reverse") P.gp_reverse = true;
    else if (a == "--table-backend") backend = need(i);
    else if (a == "--read-stats") readStats = true;   // the call the reference has commented out at main.cpp:305
    else if (a == "-h" || a == "--help") { usage(); return 0; }
    else if (a == "--version") { std::cout << "TALC version: 1.01\nLast update: September 2019\n"; return 0; }
    else if (!a.empty() && a[0] == '-') { std::cerr << "talc_ref: unknown option " << a << "\n"; return 1; }
    else seqFile = a;
  }
  if (!haveK || !haveSR || seqFile.empty()) { usage(); return 1; }  // main.cpp:114,127,199
  P.gp_useJunctions = useJ;

  const std::string outFile = outPrefix + ".fa", statFile = outPrefix + ".stats_basics.txt",
                    logFile = outPrefix + ".log";
  {  // Settings.cpp:160-185 outputConfig
    std::ofstream o(outPrefix + ".config.txt", std::ios_base::trunc);
    o << "TALC: Parameters used for sample: " << outPrefix << "\n"
      << "****************************" << "\n"
      << "INPUT=" << seqFile << "\n"
      << "OUTPUT=" << outPrefix << "\n"
      << "STATS=" << statFile << "\n"
      << "****************************" << "\n"
      << "KmerSize=" << P.K << "\n"
      << "Junction mode activated? " << P.gp_useJunctions << "\n"
      << "queryMode=" << queryMode << "\n"
      << "****************************" << "\n"
      << "MIN_INNER_SCORE=" << P.gp_MIN_INNER_SCORE << "\n"
      << "MIN_BORDER_SCORE=" << P.gp_MIN_BORDER_SCORE << "\n"
      << "MAX_NB_BRANCHES=" << P.gp_MAX_NB_COMPETING_PATHS << "\n"
      << "ALPHA=" << P.gp_ALPHA << "\n"
      << "MIN_SR_COUNT=" << P.gp_MIN_COUNT << "\n"
      << "WINDOW_SIZE=" << P.gp_WINDOW_SIZE << "\n"
      << "****************************" << std::endl;
  }
  {  // Read.cpp:394-415 setBasicReadStatsHeader
    std::ofstream o(statFile, std::ios_base::trunc);
    o << "read_name\traw_length\twhead_length\twtail_length\tnbInKmers\tnbSolidKmers\tnbSolidReg\tnbInWeakReg\t"
         "nbInCorrReg\tCorrHead?\tCorrHeadLen\tCorrTail?\tCorrTailLen\tCorrlength\tnbInKmers2\n";
  }

  std::vector<std::string> ids;
  std::vector<TSeq> seqs;
  if (loadSeqData(ids, seqs, seqFile) != 0) {  // main.cpp:219,323
    std::cout << "[TALC]: ISSUE WITH INPUT FILES" << std::endl;
    return 0;
  }
  std::cout << "[TALC]: " << ids.size() << " long read(s) loaded" << std::endl;

  Table table(backend == "flat" ? Table::FLAT : Table::MAP);
  auto t0 = std::chrono::steady_clock::now();
  if (queryMode == "memory") {  // main.cpp:224-238
    BuildStats st = buildCDBG(table, dump, jdump, P);
    decolourRepeatsFromDBG(table, P);
    std::cout << "There were " << st.onlineCounter << " k-mers retrieved from database." << std::endl;
    std::cout << "[TALC]: SR-dBG contains " << table.size() << " nodes." << std::endl;
  }
  auto t1 = std::chrono::steady_clock::now();
  if (!(queryMode == "jellyfish2" || table.size() > 0)) {  // main.cpp:240,317-321
    std::cout << "[TALC]: The de Bruijn Graph is empty...Correction aborted." << std::endl;
    return 1;
  }
  Ctx C; C.P = P; C.dBG = &table;
  std::vector<int> status(ids.size(), 0);
  std::vector<BasicReadStats> bstats(ids.size());
  omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(dynamic)
  for (long r = 0; r < (long)ids.size(); r++) status[r] = (int)correctOneRead(C, ids[r], seqs[r], nullptr, &bstats[r]);
  auto t2 = std::chrono::steady_clock::now();
  if (readStats) {   // Read.cpp:425-431, in input order
    std::ofstream o(statFile, std::ios_base::app);
    for (size_t r = 0; r < ids.size(); ++r)
      if (bstats[r].written)
        o << "\n" << ids[r] << "\t" << bstats[r].rawLength << "\t" << bstats[r].nbInKmersBefore << "\t" << bstats[r].nbSReg << "\t" << bstats[r].corrLength;
  }
  for (size_t r = 0; r < ids.size(); ++r) {  // main.cpp:290,294 (input order)
    if (status[r] == RS_NO_STRUCTURE) throwToLog(ids[r], "Unable to define convenient structure.", logFile);
    else if (status[r] == RS_NO_SOLID_KMER) throwToLog(ids[r], "No solid kmer could be found.", logFile);
  }
  outputSeqData(ids, seqs, outFile);  // main.cpp:310
  auto t3 = std::chrono::steady_clock::now();
  double bases = 0;
  for (auto& s : seqs) bases += s.size();
  auto sec = [](auto a, auto b) { return std::chrono::duration<double>(b - a).count(); };
  fprintf(stderr, "[talc_ref] threads=%d build=%.3fs correct=%.3fs write=%.3fs\n", nthreads, sec(t0, t1), sec(t1, t2),
          sec(t2, t3));
  std::cout << "[TALC]: Looks like we are done now." << std::endl;
  return 0;
}
