// talc_oracle.hpp — CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE)
//
// A plain C++17 / std-only restatement of the reference's per-long-read
// correction hot path (TALC 1.01), written from the reference sources as a
// specification.  Every function cites the reference file:line it follows
// (paths relative to /root/reference/src).
//
// PARITY STATUS: *parity unpinned*.  The reference ships no tests, golden
// vectors or fixtures and cannot be compiled in this image
// because every translation unit needs SeqAn2, which is not vendored
// (Makefile:3, README.md:26).  The SeqAn2 primitives used on the path
// (score-only global/local alignment, gapped X-drop seed extension, Horspool
// find, Dna5 conversion, FASTA I/O) are restated from their published
// behaviour in seqan_shim.hpp and pinned only by hand-derived known-answer
// tests (tests/test_oracle_primitives.py).
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
// use anything in this directory.  The product (talc_amd/) never includes,
// links or calls it.
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

namespace talc_oracle {

typedef std::string TSeq;                                     // Dna5String: chars in "ACGTN"
typedef std::pair<unsigned int, unsigned int> colouredCount;  // utils.hpp:39
typedef std::tuple<TSeq, unsigned int, unsigned int> anchorTuple;  // utils.hpp:43 (kmer, pos, count)
typedef std::pair<TSeq, colouredCount> tipNode;               // utils.hpp:44

// utils.hpp:51-57 (same enumerator order as the reference)
enum Outcome { NO, DEADEND, SUCCESS, ABORTION, BRANCHING, CYCLE };
enum Status { EXPECTED, UNEXPECTED, LOWCOUNT, SUPPORTED, CORRECTED, UNCORRECTED, ABSENT, BREAKPOINT };
enum Direction { LEFT, RIGHT };
enum Location { HEAD, INNER, TAIL, UNKNOWN };

typedef std::tuple<unsigned int, unsigned int, Status> kmerStretch;  // Explorer.hpp:44

// The reference's process globals (Settings.cpp:33-69) and file-static tunables
// (Explorer.cpp:85-104, Jellyfish.cpp:64, Read.cpp:361,368) gathered in one struct.
struct Params {
  unsigned int K = 21;
  unsigned int gp_MIN_COUNT = 2;                // main.cpp:157
  double gp_ALPHA = 2.57;                       // main.cpp:182
  unsigned int gp_WINDOW_SIZE = 9;              // main.cpp:168
  double gp_SR_ERROR_RATE = 0.025;              // main.cpp:162
  double gp_MIN_INNER_SCORE = 0.7;              // main.cpp:142
  double gp_MIN_BORDER_SCORE = 0.7;             // main.cpp:148
  unsigned int gp_MAX_NB_COMPETING_PATHS = 7;   // main.cpp:173
  bool gp_useJunctions = false;
  bool gp_reverse = false;
  // Explorer.cpp:85-102
  unsigned int p_MIN_START_ANCHORS = 3;
  unsigned int p_MAX_START_ANCHORS = 5;
  unsigned int p_MAX_IN_COUNT = 100000;
  unsigned int p_MAX_NB_OF_BORDER_PATHS = 75;
  unsigned int p_MAX_NB_OF_INNER_PATHS = 50;
  unsigned int p_CHECK_INTERVAL = 6;
  double p_ALLOWED_FAILURE_RATE = 0.3;
  int p_MAX_NB_BORDER_FAILURES = 3;
  unsigned int colouredCountThr = 10000;        // Jellyfish.cpp:64
  unsigned int maxBorderLength = 500;           // Read.cpp:361,368
};

// ---- the SR k-mer table (Jellyfish.hpp:32-34): an ordered map keyed by k-mer text ----
// Two back-ends behind one interface: MAP is the reference's own data structure
// (std::map<Dna5String, pair<uint,uint>>) and is the one timed as cpu_baseline;
// FLAT is an unordered flat hash over 2-bit packed keys used to keep big parity
// runs fast.  Both return identical values (tests/test_oracle_table.py).
class Table {
 public:
  enum Backend { MAP = 0, FLAT = 1 };
  explicit Table(Backend b = MAP) : backend_(b) {}
  // std::map::insert semantics (Jellyfish.cpp:262): the first occurrence wins.
  bool insert(const TSeq& kmer, colouredCount v);
  bool contains(const TSeq& kmer) const;
  colouredCount at(const TSeq& kmer) const;       // (0,0) if absent (Jellyfish.cpp:410-411)
  void setColour(const TSeq& kmer, unsigned int c);  // only if present
  size_t size() const;
  Backend backend() const { return backend_; }
  // bulk build helpers (not in the reference; arrays of packed 2-bit k-mers, A=0,C=1,G=2,T=3,
  // first base in the most significant position).  Keys must be pre-deduplicated or
  // are inserted first-wins in array order.
  void insertPacked(const uint64_t* keys, const uint32_t* counts, const uint32_t* jcounts,
                    uint64_t n, unsigned int K, bool sortedHint);

 private:
  Backend backend_;
  std::map<TSeq, colouredCount> map_;
  // FLAT: open addressing on packed keys (+ a side map for k-mers that contain N or have odd length)
  std::vector<uint64_t> fkeys_;
  std::vector<colouredCount> fvals_;
  std::map<TSeq, colouredCount> fother_;
  uint64_t fcount_ = 0;
  unsigned int fK_ = 0;
  void flatGrow();
  bool flatFind(uint64_t key, uint64_t& slot) const;
};

// Jellyfish.cpp:236-295 buildCDBG + utils.cpp:658-669 decolourRepeatsFromDBG
struct BuildStats { long onlineCounter = 0; long actualCounter = 0; long badLines = 0; };
BuildStats buildCDBG(Table& dBG, const std::string& countsTable, const std::string& junctionCountsTable,
                     const Params& P);
void decolourRepeatsFromDBG(Table& dBG, const Params& P);
// junction colouring on arrays (same rule as Jellyfish.cpp:278-289)
void colourJunctions(Table& dBG, const std::vector<std::pair<TSeq, long>>& junctions, const Params& P);

struct Ctx {
  Params P;
  const Table* dBG = nullptr;
};

// ---- queries (Jellyfish.cpp) ----
std::vector<TSeq> getKmers(const TSeq& Seq, unsigned int kmerSize);                          // :69-82
std::vector<TSeq> getSuccessors(const TSeq& kmer, Direction direction);                      // :116-126
std::vector<colouredCount> getNextCounts(const Ctx& C, const TSeq& kmer, Direction direction);  // :299-321
int getOutDegree(const Ctx& C, const TSeq& kmer, Direction direction);                       // :383-393
colouredCount getCount(const Ctx& C, const TSeq& kmer);                                      // :397-413
std::vector<colouredCount> getLRCountsInSR(const Ctx& C, const TSeq& Seq);                   // :471-496

// ---- utils.cpp live helpers ----
TSeq formNextKmer(const TSeq& kmer, char new_base, Direction direction);                     // :370-387
TSeq getKmerAt(const TSeq& ref, unsigned int position, unsigned int kmerSize);               // :625-629
TSeq extractSolidSequence(const TSeq& ref, unsigned int s, unsigned int e, unsigned int k);  // :632-636
TSeq extractWeakSequence(const TSeq& ref, unsigned int e, unsigned int s, unsigned int k);   // :638-642
TSeq extractWeakBorderSequence(const TSeq& ref, unsigned int kmPos, unsigned int k, Location loc);  // :644-650

// ---- count model (Explorer.cpp:1185-1298) ----
bool isExpectedbyMyModel(const Params& P, unsigned int nextc, unsigned int cc, Status classe);
bool isExpectedbyMyLastNode(const Params& P, unsigned int nextc, unsigned int cc);
void tagNextNodes(const Params& P, std::vector<std::pair<Status, double>>& nodeTags,
                  std::vector<colouredCount>& nextCounts, unsigned int count, bool complex);

// ---- Read.cpp free functions ----
bool findINRegions(const Params& P, std::vector<kmerStretch>& StartEndKmers,
                   const std::vector<colouredCount>& counts);                                // :440-489
double computeSeqErrorThreshold(const Params& P, const std::vector<colouredCount>& counts);  // :493-518
void analyzeINRegions(const Ctx& C, std::vector<kmerStretch>& StartEndKmers, const TSeq& refSequence,
                      const std::vector<colouredCount>& counts, double solidityThr);         // :524-600

// ---- Trail (Trail.hpp / Trail.cpp) ----
class Trail {
 public:
  Trail();
  Trail(const TSeq& kmer, const colouredCount& ccount);                                      // :57
  Trail(const Trail& path, char newBase, Direction direction, unsigned int count);           // :76
  const TSeq& getSeq() const { return m_sequence; }
  unsigned int getLength() const { return (unsigned int)m_sequence.size(); }
  unsigned int getLastCount() const { return m_lastStep.second.first; }
  const TSeq& getLastKmer() const { return m_lastStep.first; }
  std::vector<colouredCount> whatsNext(const Ctx& C, Direction direction) const;             // :305
  double getLastScore() const { return m_lastScore; }
  void setLastScore(double s) { m_lastScore = s; }
  void Overlapscore(const TSeq& reference, Direction direction);                             // :145
  unsigned int getNbFailuresInARow() const { return m_nbFailuresInARow; }
  bool seedAndExtend(const Ctx& C, const TSeq& reference, Direction direction, int xdrop,
                     unsigned int MAX_FAILURES);                                             // :193
  int getNbBreakpoints() const { return m_nbBreakpoints; }
  void recordBreakpoint() { m_nbBreakpoints++; }
  double getDistance() const { return m_distance; }
  void recordDistance(double d) { m_distance += d; }
  int getLeftAnchor() const { return m_leftAnchor; }
  void setLeftAnchor(int p) { m_leftAnchor = p; }
  int getRightAnchor() const { return m_rightAnchor; }
  void setRightAnchor(int p) { m_rightAnchor = p; }
  bool ThinkIveAlreadyGotThere(const TSeq& history) const;                                   // :289
  bool checkAims(const std::vector<anchorTuple>& aims, Direction direction);                 // :273

 private:
  TSeq m_sequence;
  tipNode m_lastStep;
  double m_lastScore;
  unsigned int m_nbFailuresInARow;
  int m_nbBreakpoints;
  double m_distance;
  int m_leftAnchor;
  int m_rightAnchor;
};

// Trail.cpp:341-437  -> (refExtension, histExtension, posOnRef, score, stopThere)
std::tuple<TSeq, TSeq, int, double, bool> getSeedAndExtension(const TSeq& reference, const TSeq& candidate,
                                                              int xdrop, Direction direction,
                                                              unsigned int seedSize);

// ---- Trajectory (Trajectory.hpp / Trajectory.cpp) ----
class Trajectory {
 public:
  Trajectory();                                                                              // :37
  explicit Trajectory(const Trail& trail);                                                   // :43
  const TSeq& getSeq() const { return m_sequence; }
  void setSeq(const TSeq& s) { m_sequence = s; }
  unsigned int getLength() const { return (unsigned int)m_sequence.size(); }
  unsigned int getLeftAnchor() const { return m_leftAnchor; }
  unsigned int getRightAnchor() const { return m_rightAnchor; }
  void setRightAnchor(unsigned int r) { m_rightAnchor = r; }
  void trim(unsigned int minSize, unsigned int intervalLength, unsigned int nbFailuresInARow,
            Direction direction);                                                            // :89
  void reshape(const TSeq& reference, unsigned int kmerSize, Direction direction, bool shorter);  // :114
  bool cutAnchors(Location location, unsigned int limit, unsigned int kmerSize);             // :157
  double getLastScore() const { return m_lastScore; }
  double getScore() const { return m_score; }
  double getIDScore() const { return m_IDscore; }
  void scoreSequence(const TSeq& reference);                                                 // :239
  double getMeanDistance() const { return m_distance; }

 private:
  TSeq m_sequence;
  unsigned int m_leftAnchor;
  unsigned int m_rightAnchor;
  double m_score;
  double m_IDscore;
  int m_nbBreakpoints;
  double m_lastScore;
  double m_distance;
};

unsigned int findBestBridge(const std::vector<Trajectory>& trajectories);                    // :282
std::pair<bool, unsigned int> findBestBORDER(const std::vector<Trajectory>& trajectories);   // :306
double computeIDScore(const TSeq& gap, const TSeq& history);                                 // :337
double computeEditDistance(const TSeq& reference, const TSeq& history);                      // :386
std::tuple<TSeq, TSeq, int, double> findStopPosition(const TSeq& reference, const TSeq& shorterPath,
                                                     int xdrop, Direction direction,
                                                     unsigned int kmerSize);                 // :482
double computePercentID(const TSeq& seq1, const TSeq& seq2);                                 // :505

// Explorer.cpp:689-706, 773-865 (free functions)
void scoreBridges(const Params& P, std::vector<Trail>& newCompetingPaths, unsigned int stepCounter,
                  const TSeq& reference, Direction direction);
bool doABitOfGardening(const Params& P, std::vector<unsigned int>& indexOfKeptPaths,
                       std::vector<Trail>& newCompetingPaths);

// ---- trace (not in the reference): a flat event log used to localise GPU/oracle divergences ----
struct TraceEvent {
  int kind;              // see TraceKind
  long a, b, c, d;
  double x;
  std::string s;
};
enum TraceKind {
  TR_REGION = 1,        // a=start b=end (after analyzeINRegions)
  TR_THRESHOLD = 2,     // x=priorLambda_noise
  TR_SEARCH = 3,        // a=location b=direction c=#LEFT anchors d=#RIGHT anchors
  TR_ANCHOR = 4,        // a=side(0 LEFT,1 RIGHT) b=pos c=count s=kmer
  TR_RESULT = 5,        // a=location b=success c=LEFT.end d=RIGHT.start s=weak sequence
  TR_STEP = 6,          // a=step b=#paths after the step c=#fullPaths/#edges so far
};
struct Trace {
  bool enabled = false;
  bool steps = false;
  std::vector<TraceEvent> ev;
};

// ---- Explorer (Explorer.hpp / Explorer.cpp) ----
class Explorer {
 public:
  Explorer(const Ctx& C, const TSeq& refSequence, const std::vector<colouredCount>& coverage,
           double lambda, Trace* trace);                                                     // :132
  void reset();                                                                              // :155
  std::pair<TSeq, Status> getWeakSeq() const { return m_weakSequence; }
  void setWeakSequence();                                                                    // :218
  unsigned int getWeakLength() const { return (unsigned int)m_weakSequence.first.size(); }
  kmerStretch getLEFTHandPositions() const { return m_LEFT_KMpositions; }
  kmerStretch getRIGHTHandPositions() const { return m_RIGHT_KMpositions; }
  void initializeINNER(kmerStretch startKmPos, kmerStretch endKmPos, Direction direction);   // :228
  void initializeHEAD(kmerStretch kmPos);                                                    // :245
  void initializeTAIL(kmerStretch kmPos);                                                    // :259
  void anchorLEFTHandSide();                                                                 // :413
  void anchorRIGHTHandSide();                                                                // :480
  void oneMoreStep(const TSeq& reference, std::vector<Trail>& competingPaths,
                   unsigned int& stepCounter, unsigned int PATH_MAXLENGTH);                  // :546
  void oneMoreStepInTheDark(int& xdrop, const TSeq& reference, std::vector<Trail>& competingPaths,
                            unsigned int& stepCounter, unsigned int PATH_MAXLENGTH);         // :615
  bool searchBridge();                                                                       // :868
  bool searchEdge();                                                                         // :992
  void scoreEdges(int& xdrop, std::vector<Trail>& newCompetingPaths, unsigned int stepCounter,
                  const TSeq& reference, Direction direction);                               // :709
  Trajectory sortOutBestBorder();                                                            // :310
  void recordBridge(const Trail& trail);                                                     // :1097
  void recordEdge(const Trail& trail, const TSeq& reference);                                // :1103
  bool complexRegion() const { return m_complexRegion; }

 private:
  const Ctx& C;
  Trace* m_trace;
  TSeq m_sequence;
  std::pair<TSeq, Status> m_weakSequence;
  std::vector<colouredCount> m_coverage;
  double m_priorLambda_noise;
  kmerStretch m_LEFT_KMpositions;
  kmerStretch m_RIGHT_KMpositions;
  std::vector<anchorTuple> m_LEFT_anchors;
  std::vector<anchorTuple> m_RIGHT_anchors;
  Location m_location;
  Direction m_direction;
  bool m_complexRegion;
  std::vector<Trajectory> m_fullPaths;
  std::vector<Trajectory> m_longPaths;
  std::vector<Trajectory> m_shortPaths;
  void traceSearch();
  void traceResult(bool success);
};

// ---- Read (Read.hpp / Read.cpp) ----
// status codes of one read after the main.cpp:247-308 loop body
enum ReadStatus {
  RS_CORRECTED = 0,       // main.cpp:277-286
  RS_SKIPPED_SHORT = 1,   // main.cpp:262 (length <= K): passes through, no log line
  RS_NO_SOLID_KMER = 2,   // main.cpp:294  "No solid kmer could be found."
  RS_NO_STRUCTURE = 3,    // main.cpp:290  "Unable to define convenient structure."
};

class Read {
 public:
  Read(const Ctx& C, const std::string& id, const TSeq& sequence, Trace* trace = nullptr);   // :104
  int getLength() const { return (int)m_sequence.size(); }
  bool reCoverage();                                                                         // :174
  bool defineStructure2();                                                                   // :260
  void correct2();                                                                           // :336
  const TSeq& getCorrSeq() const { return m_correction; }
  const std::vector<colouredCount>& getCoverage() const { return m_coverage; }
  const std::vector<kmerStretch>& getRegions() const { return m_InKmersPositions; }
  double getPriorNoise() const { return m_priorLambda_noise; }
  int nbInKmers() const { return m_nbInKmers; }
  // the row Read::outputBasicReadStats appends to <o>.stats_basics.txt (Read.cpp:418-433): raw length, span of the
  // IN regions as they stand, their number, length of the correction (the name is the caller's)
  void basicReadStats(long& rawLength, unsigned int& nbInKmersBefore, int& nbSReg, long& corrLength) const;

 private:
  bool setInitialStructure();                                                                // :214
  std::pair<TSeq, Status> getSolidRegion(kmerStretch coordinates);                           // :288
  void updateINNER(Explorer& e, int reg);                                                    // :294
  void updateHEAD(Explorer& e);                                                              // :305
  void updateTAIL(Explorer& e);                                                              // :313
  void updateCorrSeq();                                                                      // :320
  const Ctx& C;
  Trace* m_trace;
  std::string m_id;
  TSeq m_sequence;
  TSeq m_correction;
  std::vector<colouredCount> m_coverage;
  double m_priorLambda_noise;
  std::vector<kmerStretch> m_InKmersPositions;
  std::vector<std::pair<TSeq, Status>> m_newInnerStructure;
  std::pair<TSeq, Status> m_head;
  std::pair<TSeq, Status> m_tail;
  int m_nbInKmers;
};

// main.cpp:247-308 loop body for one read.  `seq` is modified in place exactly like
// mySeqs[r] in the reference (including the -rev quirk: reads that are not corrected
// stay reverse-complemented, main.cpp:253 vs :286).
struct BasicReadStats { bool written = false; long rawLength = 0; unsigned int nbInKmersBefore = 0; int nbSReg = 0; long corrLength = 0; };
// stats (optional): what the commented-out call at main.cpp:305 would have appended for this read — it sits inside
// `if (getLength() > K)`, after the try block, so every read longer than K gets a row, corrected or not.
ReadStatus correctOneRead(const Ctx& C, const std::string& id, TSeq& seq, Trace* trace = nullptr, BasicReadStats* stats = nullptr);

// ---- I/O restatements (io.cpp + SeqAn semantics) ----
TSeq toDna5(const std::string& raw);                       // Dna5 conversion: acgtn -> upper, others -> N
TSeq reverseComplement(const TSeq& s);
int loadSeqData(std::vector<std::string>& ids, std::vector<TSeq>& seqs, const std::string& file);  // io.cpp:26
int outputSeqData(const std::vector<std::string>& ids, const std::vector<TSeq>& seqs,
                  const std::string& file);                                                  // io.cpp:50
void throwToLog(const std::string& seqName, const std::string& chaine, const std::string& outFile);  // io.cpp:105

// counters of "cannot happen" situations met (oracle-undefined behaviour in the reference)
struct UBCounters {
  long infixClamped = 0;      // infix/prefix/suffix with out-of-range or inverted bounds
  long seedTooShort = 0;      // getSeedAndExtension on a sequence shorter than the seed
  long gardeningOOB = 0;      // Explorer.cpp:852 out-of-range read (guarded)
};
UBCounters& ubCounters();

}  // namespace talc_oracle
