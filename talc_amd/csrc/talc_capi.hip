// talc_capi.hip — the C ABI of libtalc_hip.so (include/talc_hip.h): host orchestration of the
// GPU k-mer table and of the per-read correction kernels.  gfx950 only.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <numeric>
#include <sstream>
#include <string>
#include <vector>

#include "talc_common.h"
#include "talc_hip.h"
#include "talc_kernels_build.h"
#include "talc_kernels_probe.h"
#include "talc_kernels_search.h"
#include "talc_table_host.h"

using namespace talc;

// ------------------------------------------------------------------ error plumbing
static thread_local std::string g_err;
static int fail(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
#define HIPCHK(x)                                                                          \
  do {                                                                                     \
    hipError_t _e = (x);                                                                   \
    if (_e != hipSuccess) return fail(TALC_ERR_DEVICE, "%s failed: %s (%s:%d)", #x, hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)

// The table lives where it was built.  A host-built table has a host image (h.right / h.left) that uploads copy; a
// table built (or imported) on a GPU has a *staged* device image there — colouring and de-colouring run on it as
// kernels, talc_table_upload to that same GPU adopts it without any copy, and the host image is only materialised
// when something asks for it (host lookups, an upload to another GPU).
struct talc_table {
  HostTable h;
  bool hostValid = true;       // h.right / h.left hold the current table
  int stagedDev = -1;          // GPU holding the built, not yet uploaded image (-1: none)
  Bucket* stR = nullptr;
  Bucket* stL = nullptr;
};

// make the host image current (device-built tables: copy it back from the GPU that holds it)
static int ensure_host(talc_table* t) {
  if (t->hostValid) return TALC_OK;
  const Bucket *srcR = nullptr, *srcL = nullptr;
  int dev = -1;
  if (t->stagedDev >= 0) { srcR = t->stR; srcL = t->stL; dev = t->stagedDev; }
  else if (!t->h.dev.empty()) { srcR = t->h.dev.begin()->second.right; srcL = t->h.dev.begin()->second.left; dev = t->h.dev.begin()->first; }
  if (!srcR) return fail(TALC_ERR_STATE, "the table has neither a host image nor a device image");
  const uint64_t bytes = t->h.capacity * sizeof(Bucket);
  if (!t->h.right) t->h.right = (Bucket*)malloc(bytes);
  if (!t->h.left) t->h.left = (Bucket*)malloc(bytes);
  if (!t->h.right || !t->h.left) return fail(TALC_ERR_NOMEM, "cannot allocate the host image (%llu bytes)", (unsigned long long)(2 * bytes));
  hipError_t e = hipSetDevice(dev);
  if (e == hipSuccess) e = hipMemcpy(t->h.right, srcR, bytes, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(t->h.left, srcL, bytes, hipMemcpyDeviceToHost);
  if (e != hipSuccess) return fail(TALC_ERR_DEVICE, "copying the table image to the host: %s", hipGetErrorString(e));
  t->hostValid = true;
  return TALC_OK;
}

struct Stage {
  // per-wave scratch for the search kernel
  uint8_t* scratch = nullptr;
  uint64_t scratch_bytes = 0;
  uint32_t n_slots = 0;
  SearchCaps caps;
  // edge tasks (first pass only): one box per slot + the slots' claim counters (talc_kernels_search.h, "edge tasks")
  uint8_t* boxes = nullptr;
  uint64_t boxes_bytes = 0;
  uint32_t box_seq_cap = 0;
};

struct talc_ctx {
  talc_table* table = nullptr;
  talc_params p;
  DevParams dp;
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev[8];
  TableView view;
  talc_timing timing;
  Stage stage;          // default scratch
  uint32_t* d_queue = nullptr;   // work-queue counters
  uint32_t* d_hist = nullptr;    // 1024 buckets of the work-queue ordering
  uint64_t* d_counters = nullptr;  // [0]=trail steps [1]=dp cells
  uint32_t* d_thr = nullptr;       // the count model's thresholds by count (DevParams.thr)
  // device buffers of finished batches, kept for the next batch of this context (a streaming run creates and destroys
  // a batch per chunk of reads: ~20 hipMalloc / hipFree pairs each time otherwise)
  std::vector<std::pair<uint64_t, void*>> pool;   // (bytes, pointer), free
  std::map<void*, uint64_t> live;                 // pointer -> bytes, handed out
  uint64_t pool_bytes = 0;        // bytes cached (free)
  uint64_t live_bytes = 0;        // bytes handed out
  uint64_t peak_live_bytes = 0;   // the largest footprint the batches of this context have had together
};

// drop cached buffers, oldest first, until the cache holds at most `keep_bytes`
static void ctx_pool_trim(talc_ctx* c, uint64_t keep_bytes) {
  while (!c->pool.empty() && c->pool_bytes > keep_bytes) {
    c->pool_bytes -= c->pool.front().first;
    hipFree(c->pool.front().second);
    c->pool.erase(c->pool.begin());
  }
}

// a device buffer of at least `bytes` from the context's cache (smallest cached one that fits and is not more than
// twice as large), or a fresh one
static int ctx_alloc(talc_ctx* c, void** out, uint64_t bytes) {
  bytes = std::max<uint64_t>(bytes, 256);
  int best = -1;
  for (int i = 0; i < (int)c->pool.size(); ++i)
    if (c->pool[i].first >= bytes && c->pool[i].first <= 2 * bytes + 4096 && (best < 0 || c->pool[i].first < c->pool[best].first)) best = i;
  if (best >= 0) {
    *out = c->pool[best].second;
    c->live[*out] = c->pool[best].first;
    c->live_bytes += c->pool[best].first;
    c->peak_live_bytes = std::max(c->peak_live_bytes, c->live_bytes);
    c->pool_bytes -= c->pool[best].first;
    c->pool.erase(c->pool.begin() + best);
    return TALC_OK;
  }
  if (hipMalloc(out, bytes) != hipSuccess) {
    // out of memory: drop the cache and try once more
    (void)hipGetLastError();
    ctx_pool_trim(c, 0);
    HIPCHK(hipMalloc(out, bytes));
  }
  c->live[*out] = bytes;
  c->live_bytes += bytes;
  c->peak_live_bytes = std::max(c->peak_live_bytes, c->live_bytes);
  return TALC_OK;
}
static void ctx_release(talc_ctx* c, void* p) {
  if (!p) return;
  auto it = c->live.find(p);
  if (it == c->live.end()) { hipFree(p); return; }
  const uint64_t bytes = it->second;
  c->live.erase(it);
  c->live_bytes -= bytes;
  c->pool.push_back({bytes, p});
  c->pool_bytes += bytes;
  // a bounded cache, by count and by bytes: what one batch hands back is what the next one of the same shape asks for, so
  // cache + live buffers never need to exceed the largest footprint the batches of this context have had (everybody
  // else who sizes something from hipMemGetInfo — the search scratch, the retry stage, another context on the same GPU,
  // the walk-table decision of an upload — sees cached bytes as used)
  while (c->pool.size() > 64) {
    c->pool_bytes -= c->pool.front().first;
    hipFree(c->pool.front().second);
    c->pool.erase(c->pool.begin());
  }
  if (c->pool_bytes + c->live_bytes > c->peak_live_bytes)
    ctx_pool_trim(c, c->peak_live_bytes > c->live_bytes ? c->peak_live_bytes - c->live_bytes : 0);
}

struct talc_batch {
  talc_ctx* ctx = nullptr;
  uint32_t n_reads = 0;
  uint64_t n_bases = 0, n_kmers = 0;
  uint32_t max_len = 0;
  std::vector<uint64_t> h_offsets, h_koff;
  std::vector<uint32_t> h_tile_read, h_tile_start, h_chunk_read, h_chunk_start, h_order;
  uint8_t* d_raw = nullptr;
  uint8_t* d_codes = nullptr;
  uint64_t* d_offsets = nullptr;
  uint64_t* d_koff = nullptr;
  uint32_t *d_tile_read = nullptr, *d_tile_start = nullptr, *d_chunk_read = nullptr, *d_chunk_start = nullptr;
  uint32_t* d_order = nullptr;
  uint2* d_cov = nullptr;            // hit pairs, packed per tile inside each read's dense slot (talc_common.h: CovWord)
  CovWord* d_covw = nullptr;         // one word per 64 k-mer positions
  int32_t* d_nin = nullptr;
  // structure + results
  ReadState* d_state = nullptr;
  uint32_t* d_headcov = nullptr;     // 16 x u32 per read: dense counts of its first positions (k_structure -> k_search)
  uint32_t* d_regions = nullptr;     // 3 x u32 per region slot (start, end, hit index of the start)
  uint64_t* d_regoff = nullptr;      // per-read offset (in regions) into d_regions
  std::vector<uint64_t> h_regoff;
  uint8_t* d_out = nullptr;          // corrected codes, per-read capacity slots
  uint64_t* d_outoff = nullptr;      // per-read offset into d_out
  std::vector<uint64_t> h_outoff;
  uint64_t out_capacity = 0;
  bool encoded = false, covered = false, corrected = false;
  std::vector<ReadState> h_state;
  std::vector<uint64_t> h_dense_off;
  uint8_t* d_dense = nullptr;
  uint64_t dense_cap = 0;
  uint64_t* d_dense_off = nullptr;
};

static int ctx_alloc(talc_ctx* c, void** out, uint64_t bytes);
template <typename T>
static int up(talc_ctx* c, T** d, const std::vector<T>& h, hipStream_t s) {
  size_t bytes = std::max<size_t>(h.size(), 1) * sizeof(T);
  int rc_ = ctx_alloc(c, (void**)d, bytes);
  if (rc_) return rc_;
  if (!h.empty()) HIPCHK(hipMemcpyAsync(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, s));
  return TALC_OK;
}

extern "C" {

int talc_abi_version(void) { return TALC_ABI_VERSION; }
const char* talc_last_error(void) { return g_err.c_str(); }

int talc_params_default(talc_params* p) {
  if (!p) return fail(TALC_ERR_INVALID, "null params");
  p->k = 21; p->min_count = 2; p->alpha = 2.57; p->window_size = 9; p->sr_error_rate = 0.025;
  p->min_inner_score = 0.7; p->min_border_score = 0.7; p->max_nb_competing_paths = 7; p->use_junctions = 0;
  p->reverse = 0; p->min_start_anchors = 3; p->max_start_anchors = 5; p->max_in_count = 100000;
  p->max_nb_border_paths = 75; p->max_nb_inner_paths = 50; p->check_interval = 6; p->allowed_failure_rate = 0.3;
  p->max_nb_border_failures = 3; p->coloured_count_thr = 10000; p->max_border_length = 500;
  return TALC_OK;
}

int talc_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return n;
}

// page-locked host memory for read / record buffers: copies to and from it are DMA transfers that run beside kernels
void* talc_pinned_alloc(uint64_t bytes) {
  void* p = nullptr;
  if (hipHostMalloc(&p, std::max<uint64_t>(bytes, 1), hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); fail(TALC_ERR_NOMEM, "cannot allocate %llu bytes of pinned host memory", (unsigned long long)bytes); return nullptr; }
  return p;
}
void talc_pinned_free(void* p) { if (p) (void)hipHostFree(p); }

static int check_params(const talc_params* p) {
  if (!p) return fail(TALC_ERR_INVALID, "null params");
  if (p->k < 18 || p->k > 31) return fail(TALC_ERR_INVALID, "k=%u outside the supported range 18..31", p->k);
  if (p->min_count < 1) return fail(TALC_ERR_INVALID, "min_count must be >= 1 (reference CLI: >= 2)");
  if (p->coloured_count_thr > 65535) return fail(TALC_ERR_INVALID, "coloured_count_thr must be <= 65535");
  if (p->max_nb_competing_paths < 1 || p->max_nb_competing_paths > 64) return fail(TALC_ERR_INVALID, "max_nb_competing_paths must be in 1..64");
  if (p->max_nb_inner_paths < 1 || p->max_nb_inner_paths > 60) return fail(TALC_ERR_INVALID, "max_nb_inner_paths must be in 1..60");
  if (p->max_start_anchors < 1 || p->max_start_anchors > 64) return fail(TALC_ERR_INVALID, "max_start_anchors must be in 1..64");
  if (p->check_interval < 1) return fail(TALC_ERR_INVALID, "check_interval must be >= 1");
  if (!(p->sr_error_rate > 0)) return fail(TALC_ERR_INVALID, "sr_error_rate must be > 0");
  return TALC_OK;
}

// ------------------------------------------------------------------ table
int talc_table_from_arrays(const uint64_t* kmers, const uint32_t* counts, uint64_t n, const talc_params* p,
                           talc_table** out) {
  int rc = check_params(p);
  if (rc) return rc;
  if (!out || (n && (!kmers || !counts))) return fail(TALC_ERR_INVALID, "null argument");
  uint64_t kept = 0;
#pragma omp parallel for reduction(+ : kept)
  for (long i = 0; i < (long)n; ++i) kept += counts[i] >= p->min_count ? 1 : 0;
  talc_table* t = new talc_table();
  t->h.p = *p;
  if (!t->h.allocate(kept)) { delete t; return fail(TALC_ERR_NOMEM, "cannot allocate host table for %llu k-mers", (unsigned long long)kept); }
  t->h.insertAll(kmers, counts, n);
  *out = t;
  return TALC_OK;
}

// the table built on `device` (talc_kernels_build.h); the image stays there (staged) until talc_table_upload adopts it
// the device builder's core: n dump lines as device arrays dK / dC (line i of the dump at index i; the kernels drop the
// lines below MIN_COUNT themselves), `kept` = how many reach MIN_COUNT (sizes the tables).  Takes ownership of dK / dC.
static int build_table_from_device_arrays(uint64_t* dK, uint32_t* dC, uint64_t n, uint64_t kept, const talc_params* p, int device,
                                          talc_table** out, double h2d_seconds, double h2d_megabytes) {
  talc_table* t = new talc_table();
  t->h.p = *p;
  {   // (sparser than load 0.5 when the device has the room: HostTable::capacity_for)
    hipDeviceProp_t prop;
    uint64_t devBytes = 0;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) devBytes = (uint64_t)prop.totalGlobalMem;
    t->h.capacity = HostTable::capacity_for(kept, devBytes);
  }
  uint32_t *dSR = nullptr, *dSL = nullptr;
  unsigned long long* dStats = nullptr;
  Bucket *dR = nullptr, *dL = nullptr;
  auto cleanup = [&]() { hipFree(dK); hipFree(dC); hipFree(dSR); hipFree(dSL); hipFree(dStats); };
  if (t->h.capacity >= (1ULL << 32)) { cleanup(); delete t; return fail(TALC_ERR_NOMEM, "table of %llu k-mers exceeds 2^32 buckets", (unsigned long long)kept); }
  t->hostValid = false;
  const uint64_t cap = t->h.capacity, bytes = cap * sizeof(Bucket);
#define BCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { cleanup(); hipFree(dR); hipFree(dL); delete t; return fail(TALC_ERR_DEVICE, "%s: %s", #x, hipGetErrorString(e_)); } } while (0)
  const bool timing = getenv("TALC_TIMING") != nullptr;
  auto tnow = []() { return std::chrono::steady_clock::now(); };
  auto tsec = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
  const auto td1 = tnow();
  BCHK(hipSetDevice(device));
  BCHK(hipMalloc((void**)&dR, bytes)); BCHK(hipMalloc((void**)&dL, bytes));
  BCHK(hipMalloc((void**)&dSR, std::max<uint64_t>(n, 1) * 4)); BCHK(hipMalloc((void**)&dSL, std::max<uint64_t>(n, 1) * 4));
  BCHK(hipMalloc((void**)&dStats, 3 * 8));
  BCHK(hipMemset(dR, 0xFF, bytes)); BCHK(hipMemset(dL, 0xFF, bytes)); BCHK(hipMemset(dStats, 0, 3 * 8));
  BCHK(hipDeviceSynchronize());
  const auto td2 = tnow();
  if (n) {
    const unsigned nb = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_build_claim, dim3(nb), dim3(256), 0, 0, dR, dL, cap, p->k, dK, dC, n, p->min_count, dSR, dSL);
    hipLaunchKernelGGL(k_build_resolve, dim3(nb), dim3(256), 0, 0, dR, dL, p->k, dK, n, dSR, dSL);
    hipLaunchKernelGGL(k_build_write, dim3(nb), dim3(256), 0, 0, dR, dL, p->k, dK, dC, n, dSR, dSL);
  }
  hipLaunchKernelGGL(k_build_finalize, dim3((unsigned)((cap + 255) / 256)), dim3(256), 0, 0, dR, dL, cap, dStats);
  BCHK(hipGetLastError());
  BCHK(hipDeviceSynchronize());
  unsigned long long st[3];
  BCHK(hipMemcpy(st, dStats, 3 * 8, hipMemcpyDeviceToHost));
#undef BCHK
  const auto td3 = tnow();
  cleanup();
  if (timing) fprintf(stderr, "[talc-lib] device build: %.0f MB to the device %.3f s, table allocations + clears %.3f s, kernels %.3f s, frees %.3f s\n",
                      h2d_megabytes, h2d_seconds, tsec(td1, td2), tsec(td2, td3), tsec(td3, tnow()));
  t->stagedDev = device; t->stR = dR; t->stL = dL;
  t->h.nkmers = st[0]; t->h.nbuckets_right = st[1]; t->h.nbuckets_left = st[2];
  *out = t;
  return TALC_OK;
}

int talc_table_from_arrays_device(const uint64_t* kmers, const uint32_t* counts, uint64_t n, const talc_params* p, int device,
                                  talc_table** out) {
  int rc = check_params(p);
  if (rc) return rc;
  if (!out || (n && (!kmers || !counts))) return fail(TALC_ERR_INVALID, "null argument");
  if (n >= 0xFFFFFFFEull) return fail(TALC_ERR_INVALID, "the device builder takes fewer than 2^32-2 entries");
  uint64_t kept = 0;
#pragma omp parallel for reduction(+ : kept)
  for (long i = 0; i < (long)n; ++i) kept += counts[i] >= p->min_count ? 1 : 0;
  uint64_t* dK = nullptr; uint32_t* dC = nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  hipError_t e = hipSetDevice(device);
  if (e == hipSuccess) e = hipFree(nullptr);   // (the runtime's own start-up)
  if (e == hipSuccess) e = hipMalloc((void**)&dK, std::max<uint64_t>(n, 1) * 8);
  if (e == hipSuccess) e = hipMalloc((void**)&dC, std::max<uint64_t>(n, 1) * 4);
  if (e == hipSuccess && n) e = hipMemcpy(dK, kmers, n * 8, hipMemcpyHostToDevice);
  if (e == hipSuccess && n) e = hipMemcpy(dC, counts, n * 4, hipMemcpyHostToDevice);
  if (e != hipSuccess) { hipFree(dK); hipFree(dC); return fail(TALC_ERR_DEVICE, "copying the dump's arrays to the device: %s", hipGetErrorString(e)); }
  const double h2d = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return build_table_from_device_arrays(dK, dC, n, kept, p, device, out, h2d, (double)n * 12 / 1e6);
}

// The text dump parsed ON the device (talc_kernels_build.h): the file's bytes are read by a few host threads into
// page-locked buffers and copied as they are; two kernels make the builder's arrays.  Returns TALC_OK with *out set, or
// a positive value when the file is not for this route (too small to matter, a line that is not canonical, no memory):
// the caller then parses on the host, as before.
static int table_from_text_on_device(const char* path, const talc_params* p, int device, talc_table** out, DumpStats& ds) {
  struct stat sb;
  if (stat(path, &sb) != 0) return fail(TALC_ERR_IO, "cannot open %s", path);
  const uint64_t size = (uint64_t)sb.st_size;
  if (size < (8u << 20) || getenv("TALC_HOST_PARSE")) return 1;
  {   // a Jellyfish 2 count file goes the host's way (talc_jf.h)
    char head[64] = {0};
    FILE* f = fopen(path, "rb");
    if (!f) return fail(TALC_ERR_IO, "cannot open %s", path);
    const size_t got = fread(head, 1, sizeof head, f);
    fclose(f);
    if (jfLooksLike(head, got)) return 1;
  }
  const bool timing = getenv("TALC_TIMING") != nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  auto secs = [&](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count(); };
  if (hipSetDevice(device) != hipSuccess || hipFree(nullptr) != hipSuccess) return fail(TALC_ERR_DEVICE, "device %d cannot be used", device);
  uint8_t* dText = nullptr;
  if (hipMalloc((void**)&dText, size + 64) != hipSuccess) { (void)hipGetLastError(); return 1; }
  // ---- the file's bytes to the device: reader threads, each with its own descriptor, page-locked buffer and stream
  const uint64_t CH = 32ull << 20;
  const uint64_t nch = (size + CH - 1) / CH;
  const int T = (int)std::min<uint64_t>(nch, 8);
  std::atomic<uint64_t> next{0};
  std::atomic<int> err{0};
#pragma omp parallel num_threads(T)
  {
    int fd = open(path, O_RDONLY);
    void* pin = nullptr;
    hipStream_t st = nullptr;
    bool ok = fd >= 0 && hipSetDevice(device) == hipSuccess && hipHostMalloc(&pin, CH) == hipSuccess && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess;
    while (ok && !err.load()) {
      const uint64_t i = next.fetch_add(1);
      if (i >= nch) break;
      const uint64_t off = i * CH, len = std::min<uint64_t>(CH, size - off);
      uint64_t got = 0;
      while (got < len) { const ssize_t r = pread(fd, (char*)pin + got, len - got, (off_t)(off + got)); if (r <= 0) { ok = false; break; } got += (uint64_t)r; }
      if (!ok) break;
      if (hipMemcpyAsync(dText + off, pin, len, hipMemcpyHostToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) ok = false;
    }
    if (!ok) err.store(1);
    if (st) hipStreamDestroy(st);
    if (pin) hipHostFree(pin);
    if (fd >= 0) close(fd);
  }
  if (err.load()) { hipFree(dText); (void)hipGetLastError(); return 1; }
  const double tUp = secs(t0);
  // ---- lines per tile, the tiles' first line numbers, the lines themselves
  const auto t1 = std::chrono::steady_clock::now();
  const uint64_t ntiles = (size + kParseTile - 1) / kParseTile;
  uint32_t* dCount = nullptr; uint64_t* dFirst = nullptr; ParseStats* dPS = nullptr;
  uint64_t* dK = nullptr; uint32_t* dC = nullptr;
  auto drop = [&]() { hipFree(dText); hipFree(dCount); hipFree(dFirst); hipFree(dPS); hipFree(dK); hipFree(dC); (void)hipGetLastError(); };
  if (ntiles >= (1ull << 31) || hipMalloc((void**)&dCount, ntiles * 4) != hipSuccess || hipMalloc((void**)&dFirst, ntiles * 8) != hipSuccess ||
      hipMalloc((void**)&dPS, sizeof(ParseStats)) != hipSuccess || hipMemset(dPS, 0, sizeof(ParseStats)) != hipSuccess) { drop(); return 1; }
  hipLaunchKernelGGL(k_parse_count, dim3((unsigned)ntiles), dim3(kParseThreads), 0, 0, dText, size, dCount);
  std::vector<uint32_t> hCount(ntiles);
  if (hipMemcpy(hCount.data(), dCount, ntiles * 4, hipMemcpyDeviceToHost) != hipSuccess) { drop(); return 1; }
  std::vector<uint64_t> hFirst(ntiles);
  uint64_t nlines = 0;
  for (uint64_t i = 0; i < ntiles; ++i) { hFirst[i] = nlines; nlines += hCount[i]; }
  if (nlines == 0 || nlines >= 0xFFFFFFFEull) { drop(); return 1; }
  if (hipMemcpy(dFirst, hFirst.data(), ntiles * 8, hipMemcpyHostToDevice) != hipSuccess ||
      hipMalloc((void**)&dK, nlines * 8) != hipSuccess || hipMalloc((void**)&dC, nlines * 4) != hipSuccess) { drop(); return 1; }
  hipLaunchKernelGGL(k_parse_lines, dim3((unsigned)ntiles), dim3(kParseThreads), 0, 0, dText, size, dFirst, p->k, p->min_count, dK, dC, dPS);
  ParseStats ps;
  if (hipGetLastError() != hipSuccess || hipMemcpy(&ps, dPS, sizeof ps, hipMemcpyDeviceToHost) != hipSuccess) { drop(); return 1; }
  hipFree(dText); hipFree(dCount); hipFree(dFirst); hipFree(dPS);
  dText = nullptr; dCount = nullptr; dFirst = nullptr; dPS = nullptr;
  if (ps.flags != 0) {   // a line the device parser does not take: the host's tokeniser decides what every line means
    hipFree(dK); hipFree(dC);
    if (timing) fprintf(stderr, "[talc-lib] the dump has lines that are not 'KMER count': parsing on the host\n");
    return 1;
  }
  if (timing) fprintf(stderr, "[talc-lib] dump parsed on the device: %.0f MB of text to the device in %.3f s (%d reader threads), %llu lines parsed in %.3f s\n",
                      (double)size / 1e6, tUp, T, (unsigned long long)nlines, secs(t1));
  ds.nread += (int64_t)nlines; ds.nkept += (int64_t)ps.kept;
  return build_table_from_device_arrays(dK, dC, nlines, ps.kept, p, device, out, 0.0, 0.0);
}

// Junction colouring (Jellyfish.cpp:273-290) on the staged device image: last line wins, both strands.
static int colour_on_device(talc_table* t, const uint64_t* jkmers, const int64_t* jcounts, uint64_t n) {
  if (n == 0) return TALC_OK;
  if (n >= (1ull << 31)) return fail(TALC_ERR_INVALID, "the device colouring takes fewer than 2^31 junction lines");
  HIPCHK(hipSetDevice(t->stagedDev));
  uint64_t hsize = 1024;
  while (hsize < 4 * n) hsize *= 2;   // two bids per line at most: load <= 0.5
  uint64_t *dJ = nullptr, *dIds = nullptr; int64_t* dC = nullptr; unsigned long long* dHK = nullptr; uint32_t* dHS = nullptr;
  auto cleanup = [&]() { hipFree(dJ); hipFree(dC); hipFree(dIds); hipFree(dHK); hipFree(dHS); };
#define CCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { cleanup(); return fail(TALC_ERR_DEVICE, "%s: %s", #x, hipGetErrorString(e_)); } } while (0)
  CCHK(hipMalloc((void**)&dJ, n * 8)); CCHK(hipMalloc((void**)&dC, n * 8)); CCHK(hipMalloc((void**)&dIds, 2 * n * 8));
  CCHK(hipMalloc((void**)&dHK, hsize * 8)); CCHK(hipMalloc((void**)&dHS, hsize * 4));
  CCHK(hipMemcpy(dJ, jkmers, n * 8, hipMemcpyHostToDevice)); CCHK(hipMemcpy(dC, jcounts, n * 8, hipMemcpyHostToDevice));
  CCHK(hipMemset(dHK, 0xFF, hsize * 8)); CCHK(hipMemset(dHS, 0, hsize * 4));
  const unsigned nb = (unsigned)((2 * n + 255) / 256);
  hipLaunchKernelGGL(k_colour_claim, dim3(nb), dim3(256), 0, 0, t->stR, t->h.capacity, t->h.p.k, dJ, dC, n, t->h.p.coloured_count_thr,
                     dIds, dHK, dHS, hsize - 1);
  hipLaunchKernelGGL(k_colour_write, dim3(nb), dim3(256), 0, 0, t->stR, t->stL, t->h.capacity, t->h.p.k, dJ, dC, n, dIds, dHK, dHS,
                     hsize - 1);
  CCHK(hipGetLastError());
  CCHK(hipDeviceSynchronize());
#undef CCHK
  cleanup();
  t->hostValid = false;
  return TALC_OK;
}
static int decolour_on_device(talc_table* t) {
  HIPCHK(hipSetDevice(t->stagedDev));
  hipLaunchKernelGGL(k_decolour_repeats, dim3(1), dim3(64), 0, 0, t->stR, t->stL, t->h.capacity, t->h.p.k);
  HIPCHK(hipGetLastError());
  HIPCHK(hipDeviceSynchronize());
  t->hostValid = false;
  return TALC_OK;
}

static int table_build_impl(const char* dump_path, const char* junction_path, const talc_params* p, int device, talc_table** out,
                            int64_t stats[3]) {
  int rc = check_params(p);
  if (rc) return rc;
  if (!dump_path || !out) return fail(TALC_ERR_INVALID, "null argument");
  // Jellyfish.cpp:251-269: whitespace-separated "kmer count" per line
  std::vector<uint64_t> kmers;
  std::vector<uint32_t> counts;
  DumpStats ds;
  std::string why;
  const bool timing = getenv("TALC_TIMING") != nullptr;   // (diagnostic: where a table build's wall time goes, on stderr)
  const auto tb0 = std::chrono::steady_clock::now();
  talc_table* t = nullptr;
  int viaDevice = 1;   // > 0: not taken
  if (device >= 0) {
    viaDevice = table_from_text_on_device(dump_path, p, device, &t, ds);
    if (viaDevice < 0) return viaDevice;
    if (viaDevice == 0 && timing)
      fprintf(stderr, "[talc-lib] dump to table on the device %.3f s (%llu lines)\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - tb0).count(), (unsigned long long)ds.nread);
  }
  if (viaDevice > 0) {
    if (!parseDumpFile(dump_path, p->k, p->min_count, true, kmers, &counts, nullptr, ds, &why))
      return why.empty() ? fail(TALC_ERR_IO, "cannot open %s", dump_path) : fail(TALC_ERR_INVALID, "%s", why.c_str());
    const auto tb1 = std::chrono::steady_clock::now();
    rc = (device >= 0) ? talc_table_from_arrays_device(kmers.data(), counts.data(), kmers.size(), p, device, &t)
                       : talc_table_from_arrays(kmers.data(), counts.data(), kmers.size(), p, &t);
    if (timing) {
      const auto tb2 = std::chrono::steady_clock::now();
      fprintf(stderr, "[talc-lib] dump parse %.3f s (%llu lines), table build on %s %.3f s\n", std::chrono::duration<double>(tb1 - tb0).count(),
              (unsigned long long)ds.nread, device >= 0 ? "the device (incl. its first HIP call)" : "the host", std::chrono::duration<double>(tb2 - tb1).count());
    }
    if (rc) return rc;
  }
  std::vector<uint64_t>().swap(kmers);
  std::vector<uint32_t>().swap(counts);
  if (junction_path && junction_path[0]) {  // Jellyfish.cpp:273-290
    std::vector<uint64_t> jk;
    std::vector<int64_t> jc;
    DumpStats js;
    if (!parseDumpFile(junction_path, p->k, 0, false, jk, nullptr, &jc, js, &why)) {
      talc_table_destroy(t);
      return why.empty() ? fail(TALC_ERR_IO, "cannot open %s", junction_path) : fail(TALC_ERR_INVALID, "%s", why.c_str());
    }
    ds.nbad += js.nbad;
    if ((rc = talc_table_colour(t, jk.data(), jc.data(), jk.size()))) { talc_table_destroy(t); return rc; }
  }
  if ((rc = talc_table_decolour_repeats(t))) { talc_table_destroy(t); return rc; }  // main.cpp:232
  if (stats) { stats[0] = ds.nread; stats[1] = ds.nkept; stats[2] = ds.nbad; }
  *out = t;
  return TALC_OK;
}

int talc_table_build(const char* dump_path, const char* junction_path, const talc_params* p, talc_table** out,
                     int64_t stats[3]) {
  return table_build_impl(dump_path, junction_path, p, -1, out, stats);
}
int talc_table_build_device(const char* dump_path, const char* junction_path, const talc_params* p, int device,
                            talc_table** out, int64_t stats[3]) {
  if (device < 0) return fail(TALC_ERR_INVALID, "device must be >= 0");
  return table_build_impl(dump_path, junction_path, p, device, out, stats);
}

int talc_table_colour(talc_table* t, const uint64_t* jkmers, const int64_t* jcounts, uint64_t n) {
  if (!t || (n && (!jkmers || !jcounts))) return fail(TALC_ERR_INVALID, "null argument");
  if (t->h.frozen) return fail(TALC_ERR_STATE, "table already uploaded (immutable)");
  if (t->stagedDev >= 0) return colour_on_device(t, jkmers, jcounts, n);
  t->h.colour(jkmers, jcounts, n);
  return TALC_OK;
}
int talc_table_decolour_repeats(talc_table* t) {
  if (!t) return fail(TALC_ERR_INVALID, "null argument");
  if (t->h.frozen) return fail(TALC_ERR_STATE, "table already uploaded (immutable)");
  if (t->stagedDev >= 0) return decolour_on_device(t);
  t->h.decolourRepeats();
  return TALC_OK;
}
uint64_t talc_table_size(const talc_table* t) { return t ? t->h.nkmers : 0; }
// bits per k-mer of the presence filter (config 2: 10 bits 2.64 ms, 14 2.49, 20 2.39, 28 2.34 for k_coverage; TALC_FILTER_BITS: experiments)
static uint64_t filter_words_for(uint64_t nkmers) {
  uint64_t bitsPerKmer = 20;
  if (const char* e = getenv("TALC_FILTER_BITS")) bitsPerKmer = std::min<uint64_t>(64, std::max<uint64_t>(4, strtoull(e, nullptr, 10)));
  return (std::max<uint64_t>(64, (nkmers * bitsPerKmer + 63) / 64) + 7) & ~7ull;   // whole 64-byte blocks
}
// device bytes of one uploaded copy (what talc_table_upload allocated), or — before any upload — of the copy an upload
// would make without the walk tables (whether those are built is decided then, from the free memory)
uint64_t talc_table_device_bytes(const talc_table* t) {
  if (!t) return 0;
  if (!t->h.dev.empty()) {
    const DeviceCopy& dc = t->h.dev.begin()->second;
    return 2 * t->h.capacity * sizeof(Bucket) + dc.filterWords * 8 + (dc.walkRight ? 2 * t->h.capacity * sizeof(WalkEntry) : 0);
  }
  return 2 * t->h.capacity * sizeof(Bucket) + filter_words_for(t->h.nkmers) * 8;
}

int talc_table_upload(talc_table* t, int device) {
  if (!t) return fail(TALC_ERR_INVALID, "null table");
  if (t->h.dev.count(device)) return TALC_OK;
  const uint64_t bytes = t->h.capacity * sizeof(Bucket);
  const bool adopt = (t->stagedDev == device);
  // everything this call allocates is freed again when a later step fails; an adopted image stays the table's staged one
  // until the copy is complete (a failed upload leaves the table as it was)
  struct Guard {
    DeviceCopy dc; bool adopted = false, done = false;
    ~Guard() {
      if (done) return;
      if (!adopted) { hipFree(dc.right); hipFree(dc.left); }
      hipFree(dc.filter); hipFree(dc.walkRight); hipFree(dc.walkLeft);
    }
  } g;
  DeviceCopy& dc = g.dc;
  g.adopted = adopt;
  if (adopt) {   // built (or imported) on this GPU: the image is adopted as it stands
    HIPCHK(hipSetDevice(device));
    dc.right = t->stR; dc.left = t->stL;
  } else {
    int rc = ensure_host(t);
    if (rc) return rc;
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipMalloc((void**)&dc.right, bytes));
    HIPCHK(hipMalloc((void**)&dc.left, bytes));
    HIPCHK(hipMemcpy(dc.right, t->h.right, bytes, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dc.left, t->h.left, bytes, hipMemcpyHostToDevice));
  }
  {   // presence filter, from the RIGHT table
    dc.filterWords = filter_words_for(t->h.nkmers);
    HIPCHK(hipMalloc((void**)&dc.filter, dc.filterWords * 8));
    HIPCHK(hipMemset(dc.filter, 0, dc.filterWords * 8));
    if (t->h.capacity)
      hipLaunchKernelGGL(k_build_filter, dim3((unsigned)((t->h.capacity + 255) / 256)), dim3(256), 0, 0, dc.right, t->h.capacity,
                         t->h.p.k, (unsigned long long*)dc.filter, dc.filterWords);
    HIPCHK(hipGetLastError());
    // ... and every RIGHT bucket's in-degree into its key word (talc_common.h: the coverage kernel's left degrees)
    if (t->h.capacity)
      hipLaunchKernelGGL(k_build_indegree, dim3((unsigned)((t->h.capacity + 255) / 256)), dim3(256), 0, 0, dc.right, dc.left, t->h.capacity,
                         (uint32_t)t->h.p.min_count);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
  }
  // walk tables (WalkEntry, talc_common.h).  Built when they leave the correction batches and their scratch a reserve
  // (64 GB, or a quarter of the device if that is less).  TALC_WALK=0 turns them off, TALC_WALK=1 insists.
  {
    const char* env = getenv("TALC_WALK");
    const uint64_t wbytes = t->h.capacity * sizeof(WalkEntry);
    size_t freeB = 0, totalB = 0;
    HIPCHK(hipMemGetInfo(&freeB, &totalB));
    const uint64_t reserve = std::min<uint64_t>(64ull << 30, (uint64_t)totalB / 4);
    const bool want = env ? atoi(env) != 0 : ((uint64_t)freeB >= 2 * wbytes + reserve);
    if (want && t->h.capacity) {
      if (hipMalloc((void**)&dc.walkRight, wbytes) != hipSuccess || hipMalloc((void**)&dc.walkLeft, wbytes) != hipSuccess) {
        (void)hipGetLastError();
        hipFree(dc.walkRight); dc.walkRight = dc.walkLeft = nullptr;
        if (env) return fail(TALC_ERR_NOMEM, "TALC_WALK=1 but the walk tables (%llu bytes) do not fit the device", (unsigned long long)(2 * wbytes));
      } else {
        const uint64_t nthr = 2 * t->h.capacity;
        hipLaunchKernelGGL(k_build_walk, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, 0, dc.right, dc.left, t->h.capacity,
                           t->h.p.k, (uint32_t)t->h.p.min_count, dc.walkRight, dc.walkLeft);
        HIPCHK(hipGetLastError());
        HIPCHK(hipDeviceSynchronize());
      }
    }
  }
  t->h.dev[device] = dc;
  g.done = true;
  if (adopt) { t->stR = t->stL = nullptr; t->stagedDev = -1; }
  t->h.frozen = true;
  return TALC_OK;
}

// ---- the device image as plain bytes (replication across the GPUs of a node: rank 0 builds, the image travels over
// RCCL / xGMI in caller-owned device buffers, every other rank imports it)
uint64_t talc_table_capacity(const talc_table* t) { return t ? t->h.capacity : 0; }
uint64_t talc_table_image_bytes(const talc_table* t) { return t ? t->h.capacity * sizeof(Bucket) : 0; }

int talc_table_export_device(talc_table* t, int device, void* dst_right, void* dst_left) {
  if (!t || !dst_right || !dst_left) return fail(TALC_ERR_INVALID, "null argument");
  const Bucket *srcR = nullptr, *srcL = nullptr;
  auto it = t->h.dev.find(device);
  if (it != t->h.dev.end()) { srcR = it->second.right; srcL = it->second.left; }
  else if (t->stagedDev == device) { srcR = t->stR; srcL = t->stL; }
  else return fail(TALC_ERR_STATE, "the table has no image on device %d", device);
  HIPCHK(hipSetDevice(device));
  const uint64_t bytes = t->h.capacity * sizeof(Bucket);
  HIPCHK(hipMemcpy(dst_right, srcR, bytes, hipMemcpyDeviceToDevice));
  HIPCHK(hipMemcpy(dst_left, srcL, bytes, hipMemcpyDeviceToDevice));
  HIPCHK(hipDeviceSynchronize());
  return TALC_OK;
}

int talc_table_import_device(const talc_params* p, uint64_t capacity, uint64_t n_kmers, const void* src_right, const void* src_left,
                             int device, talc_table** out) {
  int rc = check_params(p);
  if (rc) return rc;
  if (!out || !src_right || !src_left || capacity == 0 || capacity >= (1ULL << 32)) return fail(TALC_ERR_INVALID, "bad argument");
  talc_table* t = new talc_table();
  t->h.p = *p; t->h.capacity = capacity; t->h.nkmers = n_kmers; t->hostValid = false;
  const uint64_t bytes = capacity * sizeof(Bucket);
  hipError_t e = hipSetDevice(device);
  if (e == hipSuccess) e = hipMalloc((void**)&t->stR, bytes);
  if (e == hipSuccess) e = hipMalloc((void**)&t->stL, bytes);
  if (e == hipSuccess) e = hipMemcpy(t->stR, src_right, bytes, hipMemcpyDeviceToDevice);
  if (e == hipSuccess) e = hipMemcpy(t->stL, src_left, bytes, hipMemcpyDeviceToDevice);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  // an image carries no parameters of its own: what the kernels rely on — every stored count >= MIN_COUNT, keys of K - 1
  // bases — is checked against the parameters given (an image filtered with a lower MIN_COUNT would give wrong regions)
  unsigned long long chk[2] = {~0ull, 0ull};
  unsigned long long* dChk = nullptr;
  if (e == hipSuccess) e = hipMalloc((void**)&dChk, sizeof chk);
  if (e == hipSuccess) e = hipMemcpy(dChk, chk, sizeof chk, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_image_check, dim3((unsigned)((capacity + 255) / 256)), dim3(256), 0, 0, t->stR, capacity, dChk);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpy(chk, dChk, sizeof chk, hipMemcpyDeviceToHost);
  hipFree(dChk);
  if (e != hipSuccess) { hipFree(t->stR); hipFree(t->stL); delete t; return fail(TALC_ERR_DEVICE, "importing the table image: %s", hipGetErrorString(e)); }
  const unsigned long long keyBits = 2ull * (p->k - 1);
  if ((chk[0] != ~0ull && chk[0] < p->min_count) || (keyBits < 64 && (chk[1] >> keyBits) != 0ull)) {
    hipFree(t->stR); hipFree(t->stL); delete t;
    return fail(TALC_ERR_INVALID, "the image does not belong to these parameters: smallest stored count %llu (MIN_COUNT %u), keys wider than %llu bits: %s",
                chk[0] == ~0ull ? 0ull : chk[0], p->min_count, keyBits, (keyBits < 64 && (chk[1] >> keyBits) != 0ull) ? "yes" : "no");
  }
  t->stagedDev = device;
  *out = t;
  return TALC_OK;
}

static int table_view(talc_table* t, int device, TableView& v) {
  auto it = t->h.dev.find(device);
  if (it == t->h.dev.end()) return fail(TALC_ERR_STATE, "table not uploaded to device %d", device);
  v.right = it->second.right; v.left = it->second.left; v.capacity = t->h.capacity; v.k = t->h.p.k;
  v.filter = it->second.filter; v.filterWords = it->second.filterWords;
  v.walkRight = it->second.walkRight; v.walkLeft = it->second.walkLeft;
  return TALC_OK;
}

int talc_table_lookup_batch(talc_table* t, int device, const uint64_t* kmers, uint64_t n, uint32_t* counts,
                            uint32_t* jcounts) {
  if (!t || !kmers || !counts || !jcounts) return fail(TALC_ERR_INVALID, "null argument");
  TableView v;
  int rc = table_view(t, device, v);
  if (rc) return rc;
  if (n == 0) return TALC_OK;
  HIPCHK(hipSetDevice(device));
  uint64_t* dk; uint32_t *dc, *dj;
  HIPCHK(hipMalloc((void**)&dk, n * 8)); HIPCHK(hipMalloc((void**)&dc, n * 4)); HIPCHK(hipMalloc((void**)&dj, n * 4));
  HIPCHK(hipMemcpy(dk, kmers, n * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_lookup, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, v, dk, n, dc, dj);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpy(counts, dc, n * 4, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(jcounts, dj, n * 4, hipMemcpyDeviceToHost));
  hipFree(dk); hipFree(dc); hipFree(dj);
  return TALC_OK;
}

int talc_table_next_counts_batch(talc_table* t, int device, const uint64_t* kmers, uint64_t n, int direction,
                                 uint32_t* counts4, uint32_t* jcounts4) {
  if (!t || !kmers || !counts4 || !jcounts4) return fail(TALC_ERR_INVALID, "null argument");
  TableView v;
  int rc = table_view(t, device, v);
  if (rc) return rc;
  if (n == 0) return TALC_OK;
  HIPCHK(hipSetDevice(device));
  uint64_t* dk; uint32_t *dc, *dj;
  HIPCHK(hipMalloc((void**)&dk, n * 8)); HIPCHK(hipMalloc((void**)&dc, n * 16)); HIPCHK(hipMalloc((void**)&dj, n * 16));
  HIPCHK(hipMemcpy(dk, kmers, n * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_next_counts, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, v, dk, n, direction ? 1 : 0, dc, dj);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpy(counts4, dc, n * 16, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(jcounts4, dj, n * 16, hipMemcpyDeviceToHost));
  hipFree(dk); hipFree(dc); hipFree(dj);
  return TALC_OK;
}

int talc_table_lookup_host_batch(const talc_table* t, const uint64_t* kmers, uint64_t n, uint32_t* counts,
                                 uint32_t* jcounts) {
  if (!t || (n && (!kmers || !counts || !jcounts))) return fail(TALC_ERR_INVALID, "null argument");
  int rc = ensure_host(const_cast<talc_table*>(t));   // (a device-built table: the image is copied back on first use)
  if (rc) return rc;
#pragma omp parallel for schedule(static) if (n > 100000)
  for (long i = 0; i < (long)n; ++i) t->h.lookup(kmers[i], counts[i], jcounts[i]);
  return TALC_OK;
}

void talc_table_destroy(talc_table* t) {
  if (!t) return;
  for (auto& kv : t->h.dev) {
    if (hipSetDevice(kv.first) == hipSuccess) { hipFree(kv.second.right); hipFree(kv.second.left); hipFree(kv.second.filter);
      hipFree(kv.second.walkRight); hipFree(kv.second.walkLeft); }
  }
  if (t->stagedDev >= 0 && hipSetDevice(t->stagedDev) == hipSuccess) { hipFree(t->stR); hipFree(t->stL); }
  delete t;
}

// ------------------------------------------------------------------ context
static void free_stage(Stage& s) { if (s.scratch) hipFree(s.scratch); if (s.boxes) hipFree(s.boxes); s = Stage(); }

int talc_ctx_create(talc_table* t, const talc_params* p, int device, talc_ctx** out) {
  int rc = check_params(p);
  if (rc) return rc;
  if (!t || !out) return fail(TALC_ERR_INVALID, "null argument");
  if (p->k != t->h.p.k) return fail(TALC_ERR_INVALID, "k mismatch between params (%u) and table (%u)", p->k, t->h.p.k);
  if (p->min_count != t->h.p.min_count)
    return fail(TALC_ERR_INVALID, "min_count mismatch between params (%u) and the table it was filtered with (%u)", p->min_count, t->h.p.min_count);
  TableView v;
  rc = table_view(t, device, v);
  if (rc) return rc;
  HIPCHK(hipSetDevice(device));
  talc_ctx* c = new talc_ctx();
  c->table = t; c->p = *p; c->device = device; c->view = v;
  memset(&c->timing, 0, sizeof c->timing);
  DevParams& d = c->dp;
  d.K = p->k; d.MIN_COUNT = p->min_count; d.ALPHA = p->alpha; d.WINDOW = p->window_size; d.ERR = p->sr_error_rate;
  d.MIN_INNER = p->min_inner_score; d.MIN_BORDER = p->min_border_score; d.MAXB = p->max_nb_competing_paths;
  d.reverse = p->reverse; d.MIN_START_ANCHORS = p->min_start_anchors; d.MAX_START_ANCHORS = p->max_start_anchors;
  d.MAX_IN_COUNT = p->max_in_count; d.MAX_BORDER_PATHS = p->max_nb_border_paths; d.MAX_INNER_PATHS = p->max_nb_inner_paths;
  d.CHECK_INTERVAL = p->check_interval; d.FAILURE_RATE = p->allowed_failure_rate;
  d.MAX_BORDER_FAILURES = p->max_nb_border_failures; d.MAX_BORDER_LEN = p->max_border_length;
  d.costEdgeLin = 2800; d.costEdgeQuad = 135;
  if (const char* e = getenv("TALC_COST_LIN")) d.costEdgeLin = (uint32_t)strtoul(e, nullptr, 10);     // (tuning runs)
  if (const char* e = getenv("TALC_COST_QUAD")) d.costEdgeQuad = (uint32_t)strtoul(e, nullptr, 10);
  d.costGapQuad = 10; d.costGapFork = 200; d.costGapCap = 900; d.pad_ = 0;   // (profiles/r03/cost_sweep.txt)
  if (const char* e = getenv("TALC_COST_GAPQ")) d.costGapQuad = (uint32_t)strtoul(e, nullptr, 10);
  if (const char* e = getenv("TALC_COST_GAPFORK")) d.costGapFork = (uint32_t)strtoul(e, nullptr, 10);
  if (const char* e = getenv("TALC_COST_GAPCAP")) d.costGapCap = (uint32_t)strtoul(e, nullptr, 10);
  HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  for (auto& e : c->ev) HIPCHK(hipEventCreate(&e));
  HIPCHK(hipMalloc((void**)&c->d_queue, kQueueWords * sizeof(uint32_t)));
  HIPCHK(hipMalloc((void**)&c->d_hist, (1024 + 256) * sizeof(uint32_t)));   // (+ the batch's fork statistics, k_order_scale)
  HIPCHK(hipMalloc((void**)&c->d_counters, (128 + 2 * 8192) * sizeof(uint64_t)));   // 128 counters + the profile build's record of every wave's last read
  {   // isExpectedbyMyModel as two thresholds per count (Explorer.cpp:1185-1201), from the formula itself, for this ALPHA
    const uint32_t n = 4096;
    HIPCHK(hipMalloc((void**)&c->d_thr, 2ull * n * sizeof(uint32_t)));
    hipLaunchKernelGGL(k_build_thresholds, dim3((n + 255) / 256), dim3(256), 0, c->stream, d.ALPHA, n, c->d_thr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    d.thr = c->d_thr; d.thrN = n; d.pad2_ = 0;
  }
  *out = c;
  return TALC_OK;
}

void talc_ctx_destroy(talc_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  free_stage(c->stage);
  for (auto& e : c->pool) hipFree(e.second);
  for (auto& e : c->live) hipFree(e.first);   // (batches that outlived their context: their memory goes with it)
  if (c->d_queue) hipFree(c->d_queue);
  if (c->d_hist) hipFree(c->d_hist);
  if (c->d_counters) hipFree(c->d_counters);
  if (c->d_thr) hipFree(c->d_thr);
  for (auto& e : c->ev) if (e) hipEventDestroy(e);
  if (c->stream) hipStreamDestroy(c->stream);
  delete c;
}

int talc_ctx_get_timing(const talc_ctx* c, talc_timing* out) {
  if (!c || !out) return fail(TALC_ERR_INVALID, "null argument");
  *out = c->timing;
  return TALC_OK;
}

// ------------------------------------------------------------------ batch
void talc_batch_destroy(talc_batch* b) {
  if (!b) return;
  (void)hipSetDevice(b->ctx->device);
  void* ptrs[] = {b->d_raw, b->d_codes, b->d_offsets, b->d_koff, b->d_tile_read, b->d_tile_start, b->d_chunk_read,
                  b->d_chunk_start, b->d_order, b->d_cov, b->d_covw, b->d_nin, b->d_state, b->d_regions, b->d_regoff, b->d_out, b->d_outoff,
                  b->d_dense, b->d_dense_off, b->d_headcov};
  for (void* p : ptrs) if (p) ctx_release(b->ctx, p);
  delete b;
}

int talc_batch_create(talc_ctx* c, const char* bases, const uint64_t* offsets, uint32_t n_reads, talc_batch** out) {
  if (!c || !offsets || !out || (!bases && n_reads && offsets[n_reads] > 0)) return fail(TALC_ERR_INVALID, "null argument");
  HIPCHK(hipSetDevice(c->device));
  talc_batch* b = new talc_batch();
  b->ctx = c; b->n_reads = n_reads;
  b->h_offsets.assign(offsets, offsets + n_reads + 1);
  if (b->h_offsets[0] != 0) { delete b; return fail(TALC_ERR_INVALID, "offsets[0] must be 0"); }
  b->n_bases = b->h_offsets[n_reads];
  const uint32_t K = c->p.k;
  b->h_koff.resize(n_reads + 1);
  b->h_regoff.resize(n_reads + 1);
  b->h_outoff.resize(n_reads + 1);
  uint64_t ko = 0, ro = 0, oo = 0;
  for (uint32_t r = 0; r < n_reads; ++r) {
    if (offsets[r + 1] < offsets[r]) { delete b; return fail(TALC_ERR_INVALID, "offsets must be non-decreasing"); }
    const uint64_t L = offsets[r + 1] - offsets[r];
    if (L > 0x7fffff00ull) { delete b; return fail(TALC_ERR_INVALID, "read %u too long", r); }
    b->max_len = std::max<uint32_t>(b->max_len, (uint32_t)L);
    const uint64_t nk = L >= K ? L - K + 1 : 0;
    b->h_koff[r] = ko; ko += nk;
    b->h_regoff[r] = ro; ro += nk / 2 + 2;
    b->h_outoff[r] = oo; oo += out_capacity_for(L);
    for (uint64_t p = 0; p < nk; p += COV_TILE) { b->h_tile_read.push_back(r); b->h_tile_start.push_back((uint32_t)p); }
    for (uint64_t p = 0; p < L; p += 4096) { b->h_chunk_read.push_back(r); b->h_chunk_start.push_back((uint32_t)p); }
  }
  b->h_koff[n_reads] = ko; b->h_regoff[n_reads] = ro; b->h_outoff[n_reads] = oo;
  b->n_kmers = ko; b->out_capacity = oo;
  // longest reads first: the path-search queue is consumed in this order
  b->h_order.resize(n_reads);
  std::iota(b->h_order.begin(), b->h_order.end(), 0u);
  std::stable_sort(b->h_order.begin(), b->h_order.end(), [&](uint32_t x, uint32_t y) {
    return (offsets[x + 1] - offsets[x]) > (offsets[y + 1] - offsets[y]);
  });
  hipStream_t s = c->stream;
  int rc;
  if ((rc = ctx_alloc(c, (void**)&b->d_raw, std::max<uint64_t>(b->n_bases, 1)))) return rc;
  // (+ 64: a search may read a stretch of a read in place, and the wave routines fetch whole 8- and 16-byte words)
  if ((rc = ctx_alloc(c, (void**)&b->d_codes, std::max<uint64_t>(b->n_bases, 1) + 64))) return rc;
  if (b->n_bases) HIPCHK(hipMemcpyAsync(b->d_raw, bases, b->n_bases, hipMemcpyHostToDevice, s));
  if ((rc = up(c, &b->d_offsets, b->h_offsets, s))) return rc;
  if ((rc = up(c, &b->d_koff, b->h_koff, s))) return rc;
  if ((rc = up(c, &b->d_regoff, b->h_regoff, s))) return rc;
  if ((rc = up(c, &b->d_outoff, b->h_outoff, s))) return rc;
  if ((rc = up(c, &b->d_tile_read, b->h_tile_read, s))) return rc;
  if ((rc = up(c, &b->d_tile_start, b->h_tile_start, s))) return rc;
  if ((rc = up(c, &b->d_chunk_read, b->h_chunk_read, s))) return rc;
  if ((rc = up(c, &b->d_chunk_start, b->h_chunk_start, s))) return rc;
  if ((rc = up(c, &b->d_order, b->h_order, s))) return rc;
  if ((rc = ctx_alloc(c, (void**)&b->d_cov, std::max<uint64_t>(b->n_kmers, 1) * sizeof(uint2)))) return rc;
  if ((rc = ctx_alloc(c, (void**)&b->d_covw, cov_words_total(b->n_kmers, n_reads) * sizeof(CovWord)))) return rc;
  if ((rc = ctx_alloc(c, (void**)&b->d_nin, std::max<uint32_t>(n_reads, 1) * sizeof(int32_t)))) return rc;
  if ((rc = ctx_alloc(c, (void**)&b->d_state, std::max<uint32_t>(n_reads, 1) * sizeof(ReadState)))) return rc;
  if ((rc = ctx_alloc(c, (void**)&b->d_headcov, std::max<uint32_t>(n_reads, 1) * (uint64_t)kHeadCov * sizeof(uint32_t)))) return rc;
  if ((rc = ctx_alloc(c, (void**)&b->d_regions, std::max<uint64_t>(ro, 1) * 3 * sizeof(uint32_t)))) return rc;
  if ((rc = ctx_alloc(c, (void**)&b->d_out, std::max<uint64_t>(oo, 1)))) return rc;
  HIPCHK(hipStreamSynchronize(s));
  *out = b;
  return TALC_OK;
}

uint64_t talc_batch_num_kmers(const talc_batch* b) { return b ? b->n_kmers : 0; }
uint64_t talc_batch_num_bases(const talc_batch* b) { return b ? b->n_bases : 0; }

static int launch_encode(talc_ctx* c, talc_batch* b) {
  if (!b->h_chunk_read.empty())
    hipLaunchKernelGGL(k_encode, dim3((unsigned)b->h_chunk_read.size()), dim3(256), 0, c->stream, b->d_raw, b->d_codes,
                       b->d_offsets, b->d_chunk_read, b->d_chunk_start, c->p.reverse ? 1 : 0);
  HIPCHK(hipGetLastError());
  b->encoded = true;
  return TALC_OK;
}
static int launch_coverage(talc_ctx* c, talc_batch* b) {
  HIPCHK(hipMemsetAsync(b->d_nin, 0, std::max<uint32_t>(b->n_reads, 1) * sizeof(int32_t), c->stream));
  if (!b->h_tile_read.empty())
    hipLaunchKernelGGL(k_coverage, dim3((unsigned)b->h_tile_read.size()), dim3(COV_THREADS), 0, c->stream, c->view,
                       b->d_codes, b->d_offsets, b->d_koff, b->d_tile_read, b->d_tile_start, b->d_cov, b->d_covw, b->d_nin,
                       c->p.min_count);
  HIPCHK(hipGetLastError());
  b->covered = true;
  return TALC_OK;
}

int talc_batch_coverage(talc_ctx* c, talc_batch* b) {
  if (!c || !b || b->ctx != c) return fail(TALC_ERR_INVALID, "bad context/batch");
  HIPCHK(hipSetDevice(c->device));
  int rc;
  HIPCHK(hipEventRecord(c->ev[0], c->stream));
  if (!b->encoded && (rc = launch_encode(c, b))) return rc;
  HIPCHK(hipEventRecord(c->ev[1], c->stream));
  if ((rc = launch_coverage(c, b))) return rc;
  HIPCHK(hipEventRecord(c->ev[2], c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  HIPCHK(hipEventElapsedTime(&c->timing.encode_ms, c->ev[0], c->ev[1]));
  HIPCHK(hipEventElapsedTime(&c->timing.coverage_ms, c->ev[1], c->ev[2]));
  c->timing.n_kmers = b->n_kmers; c->timing.n_bases = b->n_bases;
  return TALC_OK;
}

int talc_batch_fetch_coverage(talc_ctx* c, talc_batch* b, uint32_t* counts, uint32_t* jcounts, uint64_t* kmer_offsets,
                              int32_t* n_in_kmers) {
  if (!c || !b || b->ctx != c) return fail(TALC_ERR_INVALID, "bad context/batch");
  if (!b->covered) return fail(TALC_ERR_STATE, "coverage has not been computed for this batch");
  HIPCHK(hipSetDevice(c->device));
  if (b->n_kmers && (counts || jcounts)) {
    // the dense vector<colouredCount> of Read.cpp:174-195 exists only here: the device keeps the hits and a bitmap
    // (talc_common.h: CovWord); hipMemcpy on the null stream would not be ordered with the context's stream
    HIPCHK(hipStreamSynchronize(c->stream));
    const uint64_t nw = cov_words_total(b->n_kmers, b->n_reads);
    std::vector<uint2> h(b->n_kmers);
    std::vector<CovWord> w(nw);
    HIPCHK(hipMemcpy(h.data(), b->d_cov, b->n_kmers * sizeof(uint2), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(w.data(), b->d_covw, nw * sizeof(CovWord), hipMemcpyDeviceToHost));
    for (uint32_t r = 0; r < b->n_reads; ++r) {
      const uint64_t k0 = b->h_koff[r], nk = b->h_koff[r + 1] - k0;
      const CovWord* rw = w.data() + cov_word_base(k0, r);
      for (uint64_t p = 0; p < nk; ++p) {
        const CovWord& cw = rw[p >> 6];
        uint32_t cx = 0, cy = 0;
        if ((cw.bits >> (p & 63)) & 1ull) {
          const uint64_t idx = (p & ~(uint64_t)(TALC_COV_TILE - 1)) + cw.rank + (uint64_t)__builtin_popcountll(cw.bits & ((1ull << (p & 63)) - 1ull));
          cx = h[k0 + idx].x; cy = h[k0 + idx].y;
        }
        if (counts) counts[k0 + p] = cx;
        if (jcounts) jcounts[k0 + p] = cy & kCovColourMask;
      }
    }
  }
  if (kmer_offsets) memcpy(kmer_offsets, b->h_koff.data(), (b->n_reads + 1) * 8);
  if (n_in_kmers && b->n_reads) HIPCHK(hipMemcpy(n_in_kmers, b->d_nin, b->n_reads * 4, hipMemcpyDeviceToHost));
  return TALC_OK;
}

#include "talc_capi_correct.inc"

}  // extern "C"
