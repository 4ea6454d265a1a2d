/* talc_hip.h — C ABI of libtalc_hip.so, the MI355X-native replacement of TALC's per-long-read
 * correction hot path (reference: lbroseus/TALC 1.01, paths below are under /root/reference/src).
 *
 * TALC has no plugin/FFI API; the seam this library sits behind is made of two C++ call
 * surfaces of the reference:
 *   (1) the k-mer table surface   Jellyfish.hpp:42-71  + utils.hpp:145
 *   (2) the per-read surface      Read.hpp:43-77, driven by main.cpp:247-308
 * Each entry point below cites the reference interface it replaces.  Plain pointers and sizes
 * only; no C++ or torch types cross the boundary; no exceptions cross the boundary.
 *
 * Error convention (replaces the reference's bool returns / throw std::string,
 * Read.cpp:194,275, main.cpp:298-303): every function returns 0 on success or a negative
 * talc_error; talc_last_error() returns a human-readable message for the calling thread.
 * Per-read outcomes are reported in a status array so the caller can emit the reference's log
 * lines verbatim (main.cpp:290,294).
 *
 * Threading (reference: any number of OpenMP threads on one shared read-only map,
 * main.cpp:247): a talc_table is immutable after talc_table_upload and may be shared by any
 * number of contexts; a talc_ctx is bound to one device + one HIP stream and must be used by
 * one host thread at a time.
 */
#ifndef TALC_HIP_H
#define TALC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TALC_ABI_VERSION 1

typedef enum talc_error {
  TALC_OK = 0,
  TALC_ERR_INVALID = -1,     /* bad argument / parameter out of the supported range */
  TALC_ERR_IO = -2,          /* cannot open / parse an input file */
  TALC_ERR_NOMEM = -3,       /* host or device allocation failed */
  TALC_ERR_DEVICE = -4,      /* HIP runtime error (no GPU, launch failure, ...) */
  TALC_ERR_CAPACITY = -5,    /* caller-provided output buffer too small (needed size reported) */
  TALC_ERR_STATE = -6,       /* call sequence error (e.g. table not uploaded) */
  /* not an error (positive): the batch is complete and valid, but some reads carry TALC_READ_ERROR — they
   * exhausted the device scratch even in the retry pass and are returned unchanged, like any read the
   * reference fails on (main.cpp:298-303 logs and goes on); talc_ctx_get_timing().n_failed says how many */
  TALC_WARN_READ_ERRORS = 1
} talc_error;

/* Per-read status after the main.cpp:247-308 loop body. */
typedef enum talc_read_status {
  TALC_READ_CORRECTED = 0,      /* main.cpp:277-286: corrected sequence returned */
  TALC_READ_SKIPPED_SHORT = 1,  /* main.cpp:262: length <= K; passed through, no log line */
  TALC_READ_NO_SOLID_KMER = 2,  /* main.cpp:294: log "No solid kmer could be found." */
  TALC_READ_NO_STRUCTURE = 3,   /* main.cpp:290: log "Unable to define convenient structure." */
  TALC_READ_ERROR = 4           /* device scratch exhausted even after the retry pass; the read is
                                   passed through unchanged and talc_batch_correct returns
                                   TALC_WARN_READ_ERRORS (> 0: the other records are valid) */
} talc_read_status;

/* The reference's process globals (Settings.cpp:33-63) plus its hard-coded tunables
 * (Explorer.cpp:85-102, Jellyfish.cpp:64, Read.cpp:361,368) as one POD.  Field meaning and
 * defaults are the reference's; talc_params_default() fills them. */
typedef struct talc_params {
  uint32_t k;                       /* K, -k; 18..31 (reference CLI stops at 30, main.cpp:115) */
  uint32_t min_count;               /* gp_MIN_COUNT, --MIN_COUNT (2) */
  double alpha;                     /* gp_ALPHA, --ALPHA_FOR_PRED (2.57) */
  uint32_t window_size;             /* gp_WINDOW_SIZE, --WINDOW_SIZE (9) */
  double sr_error_rate;             /* gp_SR_ERROR_RATE, --SR_ERROR_RATE (0.025) */
  double min_inner_score;           /* gp_MIN_INNER_SCORE (0.7) */
  double min_border_score;          /* gp_MIN_BORDER_SCORE (0.7) */
  uint32_t max_nb_competing_paths;  /* gp_MAX_NB_COMPETING_PATHS, --MAX_NB_BRANCHES (7) */
  int32_t use_junctions;            /* gp_useJunctions: informational; colouring is explicit */
  int32_t reverse;                  /* gp_reverse, -rev */
  uint32_t min_start_anchors;       /* p_MIN_START_ANCHORS (3)        Explorer.cpp:85 */
  uint32_t max_start_anchors;       /* p_MAX_START_ANCHORS (5)        Explorer.cpp:86 */
  uint32_t max_in_count;            /* p_MAX_IN_COUNT (100000)        Explorer.cpp:88 */
  uint32_t max_nb_border_paths;     /* p_MAX_NB_OF_BORDER_PATHS (75)  Explorer.cpp:90 */
  uint32_t max_nb_inner_paths;      /* p_MAX_NB_OF_INNER_PATHS (50)   Explorer.cpp:91 */
  uint32_t check_interval;          /* p_CHECK_INTERVAL (6)           Explorer.cpp:93 */
  double allowed_failure_rate;      /* p_ALLOWED_FAILURE_RATE (0.3)   Explorer.cpp:94 */
  int32_t max_nb_border_failures;   /* p_MAX_NB_BORDER_FAILURES (3)   Explorer.cpp:97 */
  uint32_t coloured_count_thr;      /* colouredCountThr (10000)       Jellyfish.cpp:64; <= 65535 */
  uint32_t max_border_length;       /* head/tail limit (500)          Read.cpp:361,368 */
} talc_params;

typedef struct talc_table talc_table;   /* the SR k-mer table ("SR-dBG", Settings.cpp:50) */
typedef struct talc_ctx talc_ctx;       /* one device + stream + scratch */
typedef struct talc_batch talc_batch;   /* a batch of reads resident in HBM */

int talc_abi_version(void);
const char* talc_last_error(void);
int talc_params_default(talc_params* p);
/* number of visible HIP devices (0 when there is no GPU); never initialises a device */
int talc_device_count(void);
/* Page-locked host memory for the buffers handed to talc_batch_create / talc_batch_fetch_corrected (replaces the
 * StringSets of loadSeqData / outputSeqData, io.cpp:26-75): copies to and from it are DMA transfers that overlap the
 * kernels of another context's batch.  Ordinary (pageable) buffers work everywhere too, only slower.  NULL on failure. */
void* talc_pinned_alloc(uint64_t bytes);
void talc_pinned_free(void* p);

/* ---------------------------------------------------------------- (1) table surface ------
 * Replaces  colouredDBG buildCDBG(int, string& dump, string& junctionDump)   Jellyfish.cpp:236-295
 *           void decolourRepeatsFromDBG(colouredDBG&, K)                     utils.cpp:658-669
 * k-mers are directional (non-canonical, main.cpp:89).  Packed k-mers are 2 bits per base
 * (A=0,C=1,G=2,T=3), first base in the most significant position of the 2K-bit value. */

/* Parse a `jellyfish dump -c` text file ("KMER count" per line, whitespace separated,
 * Jellyfish.cpp:251-269): keeps lines with count >= min_count, first duplicate wins; then,
 * if junction_path is not NULL, colours k-mers from the junction dump (both strands,
 * jcount < coloured_count_thr, Jellyfish.cpp:273-290); then un-colours the 4 homopolymer
 * k-mers (utils.cpp:658-669).  Lines whose k-mer is not K letters of ACGT can never match a
 * query and are counted in stats but not stored.  A count token that does not start with a number
 * (std::stoi throws in the reference, Jellyfish.cpp:259) reads as 0 here.  stats (may be NULL) receives
 * {lines read, lines kept, malformed lines}.
 * Either path may also name a Jellyfish 2 count file (the `.jf` of `jellyfish count`, format "binary/sorted": the file
 * the reference's jellyfish2 query mode would have asked `jellyfish query` about once per look-up,
 * Jellyfish.cpp:323-379,415-467,498-552).  It is recognised by its header and read under the same contract (every
 * record is a "line"; csrc/talc_jf.h); a file that carries that header but does not verify — another format, k-mers
 * of another length than p->k, a body that is not whole records, padding bits set, a zero count — fails with
 * TALC_ERR_INVALID and a message, it is never guessed at. */
int talc_table_build(const char* dump_path, const char* junction_path, const talc_params* p,
                     talc_table** out, int64_t stats[3]);

/* Same build from arrays (dump order = array order; the count >= min_count filter and the
 * first-duplicate-wins rule are applied here).  No colouring, no de-colouring. */
int talc_table_from_arrays(const uint64_t* kmers, const uint32_t* counts, uint64_t n,
                           const talc_params* p, talc_table** out);
/* The same two builders with the insert loop of buildCDBG (Jellyfish.cpp:251-269) run on GPU `device`
 * (parallel text parse on the host, CAS insertion with the first-duplicate-wins rule on the device); the
 * result is an ordinary talc_table, identical in content: every lookup returns what it returns on a
 * host-built table.  The image stays on that GPU: talc_table_colour / talc_table_decolour_repeats run there as
 * kernels (Jellyfish.cpp:273-290: last line wins, both strands; utils.cpp:658-669), talc_table_upload to the
 * same GPU adopts it without a copy, and a host image is only made when something needs one (host lookups, an
 * upload to another GPU).  Fails with TALC_ERR_DEVICE when the GPU cannot be used. */
int talc_table_build_device(const char* dump_path, const char* junction_path, const talc_params* p,
                            int device, talc_table** out, int64_t stats[3]);
int talc_table_from_arrays_device(const uint64_t* kmers, const uint32_t* counts, uint64_t n,
                                  const talc_params* p, int device, talc_table** out);
/* Junction colouring on arrays, in order, both strands (Jellyfish.cpp:278-289). */
int talc_table_colour(talc_table* t, const uint64_t* jkmers, const int64_t* jcounts, uint64_t n);
/* decolourRepeatsFromDBG (utils.cpp:658-669). */
int talc_table_decolour_repeats(talc_table* t);

uint64_t talc_table_size(const talc_table* t);        /* SR_DBG.size() (main.cpp:237) */
/* Device memory of one uploaded copy: the two bucket tables and the presence filter, plus the walk tables
 * (2 * capacity * 32 bytes, as much as the bucket tables: the fast-forward's lookahead records) once an upload has
 * built them; before any upload, the size an upload without walk tables would have.  An upload builds them when they
 * leave a reserve (64 GB, or a quarter of the device if that is less) to the correction batches
 * (547 M k-mers: 70 GB of buckets + 70 GB of walk records + 1.4 GB of filter on a 288 GB device);
 * the environment variable TALC_WALK=0 turns them off, TALC_WALK=1 makes their allocation mandatory. */
uint64_t talc_table_device_bytes(const talc_table* t);

/* Copy the table to `device` (HBM resident, replicated per GPU).  The table becomes
 * immutable.  May be called once per device. */
int talc_table_upload(talc_table* t, int device);

/* The table image of one GPU as plain bytes, for replication across the GPUs of a node (built once,
 * sent to the peers over xGMI): two arrays of talc_table_image_bytes() bytes each (the RIGHT and the LEFT bucket
 * table, talc_table_capacity() buckets of 32 bytes).  export copies them device-to-device into caller-owned DEVICE
 * buffers on `device` (e.g. tensors that a RCCL broadcast then sends); import builds a table on `device` from such
 * buffers (it copies them; the buffers stay the caller's) with the given parameters — the exporter's — and
 * size(); the imported table is then uploaded / used like any other. */
uint64_t talc_table_capacity(const talc_table* t);
uint64_t talc_table_image_bytes(const talc_table* t);
int talc_table_export_device(talc_table* t, int device, void* dst_right, void* dst_left);
int talc_table_import_device(const talc_params* p, uint64_t capacity, uint64_t n_kmers, const void* src_right,
                             const void* src_left, int device, talc_table** out);

/* Test hooks on the uploaded table — the reference's point queries:
 *   getCount(kmer)               Jellyfish.cpp:397-413  -> (count, junction colour) or (0,0)
 *   getNextCounts(kmer, dir)     Jellyfish.cpp:299-321  -> 4 x (count, colour), order A,C,G,T
 * direction: 0 = LEFT, 1 = RIGHT (utils.hpp:56).  Host pointers. */
int talc_table_lookup_batch(talc_table* t, int device, const uint64_t* kmers, uint64_t n,
                            uint32_t* counts, uint32_t* jcounts);
int talc_table_next_counts_batch(talc_table* t, int device, const uint64_t* kmers, uint64_t n,
                                 int direction, uint32_t* counts4, uint32_t* jcounts4);
/* Host-side point query on the host image of the table (verification hook for the table
 * builder; works without a GPU, before talc_table_upload or after it). */
int talc_table_lookup_host_batch(const talc_table* t, const uint64_t* kmers, uint64_t n, uint32_t* counts,
                                 uint32_t* jcounts);
void talc_table_destroy(talc_table* t);

/* ---------------------------------------------------------------- (2) per-read surface ---
 * Replaces, for a whole batch, the loop body of main.cpp:247-308:
 *   Read(id, seq); getLength()>K; reCoverage(); defineStructure2(); correct2(); getCorrSeq()
 * (Read.hpp:43-77) including the -rev handling of main.cpp:253,286. */

/* p->k and p->min_count must be the table's: the fast-forward reads "exactly one successor >= MIN_COUNT" as
 * "the other three are absent" (tagNextNodes, Explorer.cpp:1281-1297), which only holds when the table was
 * filtered with the same MIN_COUNT (Jellyfish.cpp:260). */
int talc_ctx_create(talc_table* t, const talc_params* p, int device, talc_ctx** out);
void talc_ctx_destroy(talc_ctx* c);

/* Upload a batch: `bases` are raw characters (any case; anything but ACGT becomes N exactly
 * like SeqAn's Dna5 conversion), concatenated; offsets[n_reads+1].  Caller-owned, not retained. */
int talc_batch_create(talc_ctx* c, const char* bases, const uint64_t* offsets, uint32_t n_reads,
                      talc_batch** out);
void talc_batch_destroy(talc_batch* b);

/* Read::reCoverage (Read.cpp:174-195) for every read of the batch: the k-mer probe kernel.
 * Results stay on the device — as the hits only: a bitmap word per 64 positions plus the {count, colour} pairs of the
 * k-mers that are in the table; talc_batch_fetch_coverage expands them into the reference's dense vector. */
int talc_batch_coverage(talc_ctx* c, talc_batch* b);
/* counts/jcounts: one entry per k-mer, reads concatenated (read r contributes max(0,L_r-K+1)
 * entries); kmer_offsets[n_reads+1] (may be NULL); n_in_kmers[n_reads] = #{count > min_count}
 * (Read.cpp:190, may be NULL). */
int talc_batch_fetch_coverage(talc_ctx* c, talc_batch* b, uint32_t* counts, uint32_t* jcounts,
                              uint64_t* kmer_offsets, int32_t* n_in_kmers);
uint64_t talc_batch_num_kmers(const talc_batch* b);
uint64_t talc_batch_num_bases(const talc_batch* b);

/* The whole hot path on the device: coverage -> structure (defineStructure2) -> path search
 * (correct2) -> reassembly.  Synchronous: returns when the corrected records are in HBM. */
int talc_batch_correct(talc_ctx* c, talc_batch* b);
/* Total corrected size (bytes) so the caller can allocate; valid after talc_batch_correct. */
uint64_t talc_batch_corrected_bytes(const talc_batch* b);
/* out: corrected (or passed-through) sequences as upper-case ACGTN text, concatenated in input
 * order; out_offsets[n_reads+1]; status[n_reads] (talc_read_status).  Reads that were not
 * corrected are returned exactly as the reference leaves mySeqs[r] (Dna5-converted; still
 * reverse-complemented under -rev, main.cpp:253 vs :286). */
int talc_batch_fetch_corrected(talc_ctx* c, talc_batch* b, char* out, uint64_t out_capacity,
                               uint64_t* out_offsets, int32_t* status);

/* The rows Read::outputBasicReadStats (Read.cpp:418-433) appends to <o>.stats_basics.txt — the reference has the call
 * commented out (main.cpp:305), so its file only ever holds the header; the numbers exist on the device anyway.
 * stats5[5 r ..] = {row written (length > K, main.cpp:262), raw length, sum over the IN regions of end - start + 1 as
 * they stand after the read's last step (Read.cpp:423), number of IN regions, length of the correction (0 unless the
 * read was corrected)}.  Valid after talc_batch_correct. */
int talc_batch_fetch_read_stats(talc_ctx* c, talc_batch* b, int64_t* stats5);

/* Same records, copied device-to-device into a caller-owned DEVICE buffer (e.g. a tensor that
 * is then gathered over RCCL); out_offsets/status are host arrays and may be NULL. */
int talc_batch_copy_corrected_device(talc_ctx* c, talc_batch* b, void* device_out, uint64_t out_capacity,
                                     uint64_t* out_offsets, int32_t* status);

/* Convenience: create + correct + fetch + destroy. */
int talc_correct_batch(talc_ctx* c, const char* bases, const uint64_t* offsets, uint32_t n_reads,
                       char* out, uint64_t out_capacity, uint64_t* out_offsets, int32_t* status);

/* ---------------------------------------------------------------- measurement -----------
 * HIP-event timings (ms) of the kernels launched by the last talc_batch_coverage /
 * talc_batch_correct call on this context's stream, and work counters. */
typedef struct talc_timing {
  float encode_ms;        /* ASCII -> Dna5 codes (+ reverse complement) */
  float coverage_ms;      /* k-mer probe kernel (a4) */
  float structure_ms;     /* defineStructure2 kernel (a5-a9) */
  float search_ms;        /* path-search kernel (a10-a22) */
  float emit_ms;          /* reassembly / output kernel */
  float retry_ms;         /* second pass for reads whose scratch overflowed */
  uint64_t n_kmers;       /* k-mers probed by the coverage kernel */
  uint64_t n_bases;       /* raw bases in the batch */
  uint64_t n_trail_steps; /* successor probes issued by the search kernel (Trail-steps) */
  uint64_t n_dp_cells;    /* DP cells evaluated by the search kernel */
  uint32_t n_retried;     /* reads that needed the big-scratch retry pass */
  uint32_t n_failed;      /* reads with TALC_READ_ERROR */
} talc_timing;
int talc_ctx_get_timing(const talc_ctx* c, talc_timing* out);

/* Debug hook: textual trace of one read of a batch (regions, anchors, per-gap results) in the
 * same line format as the test oracle's trace; used to localise divergences.  Returns the
 * number of bytes needed (including the NUL). */
int64_t talc_batch_trace_read(talc_ctx* c, talc_batch* b, uint32_t read_index, char* buf, uint64_t cap);

/* Test hook: one wave-cooperative DP primitive on the device (mode 0: alignment score,
 * 1: seed-and-extend, 2: k-mer window search, 3: successor tagging); out: 12 ints; not part of the
 * reference surface. */
int talc_test_dp(talc_ctx* c, int mode, const char* a, int la, const char* b, int lb, int p0, int p1, int p2, int p3,
                 int32_t* out);

#ifdef __cplusplus
}
#endif
#endif /* TALC_HIP_H */
