// talc_jf.h — native reader of a Jellyfish 2 count file (`jellyfish count` output, the `.jf` that `-SR` names in the
// reference's jellyfish2 query mode, main.cpp:125-127), so that the table is built straight from it instead of through
// one `jellyfish query` child process per look-up (Jellyfish.cpp:323-379, 415-467, 498-552).  SURVEY §8f.4.
//
// PARITY UNPINNED: neither a `jellyfish` binary nor a `.jf` file nor the format's description exists in the build
// container; the layout below is Jellyfish 2.x's `generic_file_header` + `binary_dumper` as recollected, and the tests
// check the reader against a writer of the same layout (tests/jf_writer.py).  To make a wrong recollection fail loudly
// instead of yielding wrong counts, everything that can be verified is: the format string, key_len == 2 K, a body that
// is a whole number of records, zero padding bits in every key, no zero count.  The CLI's `-jf2 DIR` route
// (`DIR/jellyfish dump`, talc_main.cpp) is the one that depends on the real tool only.
//
//   file   := header body
//   header := 9 ASCII digits (decimal length L, zero-padded) , L bytes: a JSON object, then NUL padding so that 9 + L is
//             a multiple of the object's "alignment"
//             fields used: "format" == "binary/sorted", "key_len" (bits = 2 K), "counter_len" (bytes per count)
//   body   := records of ceil(key_len / 8) + counter_len bytes: the k-mer as a little-endian integer, two bits per base,
//             A C G T = 0 1 2 3, first base in the highest bits (the packing of talc_common.h), then the count,
//             little-endian; the order is the hash's, not the alphabet's
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <omp.h>

namespace talc {

static constexpr size_t kJfHeaderDigits = 9;

// a text dump starts with a base or '>' — never with nine digits and a brace
static inline bool jfLooksLike(const char* base, size_t size) {
  if (size < kJfHeaderDigits + 2) return false;
  for (size_t i = 0; i < kJfHeaderDigits; ++i)
    if (base[i] < '0' || base[i] > '9') return false;
  return base[kJfHeaderDigits] == '{';
}

// value of a top-level-looking  "name" : value  pair of the header (strings without their quotes)
static inline bool jfField(const std::string& js, const char* name, std::string& val) {
  const std::string key = std::string("\"") + name + "\"";
  size_t p = js.find(key);
  if (p == std::string::npos) return false;
  p += key.size();
  while (p < js.size() && (js[p] == ' ' || js[p] == '\t' || js[p] == '\n' || js[p] == '\r')) ++p;
  if (p >= js.size() || js[p] != ':') return false;
  ++p;
  while (p < js.size() && (js[p] == ' ' || js[p] == '\t' || js[p] == '\n' || js[p] == '\r')) ++p;
  if (p >= js.size()) return false;
  size_t e;
  if (js[p] == '"') {
    ++p;
    e = js.find('"', p);
    if (e == std::string::npos) return false;
  } else {
    e = p;
    while (e < js.size() && js[e] != ',' && js[e] != '}' && js[e] != ' ' && js[e] != '\n' && js[e] != '\r' && js[e] != '\t') ++e;
  }
  val = js.substr(p, e - p);
  return true;
}

struct JfLayout { size_t offset = 0; uint32_t keyBits = 0, keyBytes = 0, counterBytes = 0; uint64_t records = 0; };

static inline bool jfParseHeader(const char* base, size_t size, JfLayout& L, std::string& err) {
  const size_t hlen = (size_t)strtoull(std::string(base, kJfHeaderDigits).c_str(), nullptr, 10);
  if (hlen < 2 || kJfHeaderDigits + hlen > size) { err = "Jellyfish header longer than the file"; return false; }
  const std::string js(base + kJfHeaderDigits, hlen);
  std::string v;
  if (!jfField(js, "format", v)) { err = "Jellyfish header without a \"format\""; return false; }
  if (v != "binary/sorted") {
    err = "Jellyfish file of format \"" + v + "\": only \"binary/sorted\" (the output of `jellyfish count`) is read natively; "
          "use -jf2 DIR (DIR/jellyfish dump) or a text dump";
    return false;
  }
  if (!jfField(js, "key_len", v)) { err = "Jellyfish header without a \"key_len\""; return false; }
  L.keyBits = (uint32_t)strtoul(v.c_str(), nullptr, 10);
  if (!jfField(js, "counter_len", v)) { err = "Jellyfish header without a \"counter_len\""; return false; }
  L.counterBytes = (uint32_t)strtoul(v.c_str(), nullptr, 10);
  if (L.keyBits < 2 || L.keyBits > 64 || (L.keyBits & 1)) { err = "Jellyfish key_len " + std::to_string(L.keyBits) + " is not 2 x (a k-mer length up to 32)"; return false; }
  if (L.counterBytes < 1 || L.counterBytes > 8) { err = "Jellyfish counter_len " + std::to_string(L.counterBytes) + " is not 1..8 bytes"; return false; }
  L.keyBytes = (L.keyBits + 7) / 8;
  L.offset = kJfHeaderDigits + hlen;
  const size_t rec = L.keyBytes + L.counterBytes;
  if ((size - L.offset) % rec) { err = "Jellyfish body is not a whole number of " + std::to_string(rec) + "-byte records"; return false; }
  L.records = (size - L.offset) / rec;
  return true;
}

struct JfStats { int64_t nread = 0, nkept = 0; };

// Same contract as the text parser (talc_table_host.h): every record is "read"; with `filter` a record is kept if its
// count >= minc (Jellyfish.cpp:260); kept entries in file order; counts clamp at INT_MAX like the text path's.
// wantCounts/scounts as there.  false + err on anything that does not verify.
static inline bool jfParseImage(const char* base, size_t size, uint32_t K, uint32_t minc, bool filter, std::vector<uint64_t>& kmers,
                                std::vector<uint32_t>* counts, std::vector<int64_t>* scounts, JfStats& st, std::string& err) {
  JfLayout L;
  if (!jfParseHeader(base, size, L, err)) return false;
  if (L.keyBits != 2 * K) {
    err = "Jellyfish file holds " + std::to_string(L.keyBits / 2) + "-mers, -k says " + std::to_string(K);
    return false;
  }
  const size_t rec = L.keyBytes + L.counterBytes;
  const uint8_t* body = (const uint8_t*)base + L.offset;
  int T = omp_get_max_threads();
  if (T > 64) T = 64;
  if (L.records < (1u << 16)) T = 1;
  std::vector<std::vector<uint64_t>> lk(T);
  std::vector<std::vector<uint32_t>> lc(T);
  std::vector<std::vector<int64_t>> ls(T);
  std::vector<JfStats> lst(T);
  std::vector<int> bad(T, 0);
#pragma omp parallel num_threads(T)
  {
    const int t = omp_get_thread_num();
    const uint64_t lo = L.records * (uint64_t)t / (uint64_t)T, hi = L.records * (uint64_t)(t + 1) / (uint64_t)T;
    JfStats s;
    for (uint64_t i = lo; i < hi; ++i) {
      const uint8_t* r = body + i * rec;
      uint64_t key = 0, cnt = 0;
      memcpy(&key, r, L.keyBytes);                       // little-endian host (x86-64)
      memcpy(&cnt, r + L.keyBytes, L.counterBytes);
      if ((L.keyBits < 64 && (key >> L.keyBits) != 0) || cnt == 0) { bad[t] = 1; break; }
      const uint32_t c = cnt > 0x7fffffffull ? 0x7fffffffu : (uint32_t)cnt;
      s.nread++;
      if (filter && c < minc) continue;
      if (filter) s.nkept++;
      lk[t].push_back(key);
      if (counts) lc[t].push_back(c);
      if (scounts) ls[t].push_back((int64_t)c);
    }
    lst[t] = s;
  }
  for (int t = 0; t < T; ++t)
    if (bad[t]) { err = "Jellyfish record with non-zero padding bits or a zero count: not a binary/sorted count file of this layout"; return false; }
  size_t total = 0;
  for (int t = 0; t < T; ++t) total += lk[t].size();
  kmers.resize(total);
  if (counts) counts->resize(total);
  if (scounts) scounts->resize(total);
  size_t off = 0;
  for (int t = 0; t < T; ++t) {
    if (!lk[t].empty()) {
      memcpy(kmers.data() + off, lk[t].data(), lk[t].size() * 8);
      if (counts) memcpy(counts->data() + off, lc[t].data(), lc[t].size() * 4);
      if (scounts) memcpy(scounts->data() + off, ls[t].data(), ls[t].size() * 8);
    }
    off += lk[t].size();
    st.nread += lst[t].nread; st.nkept += lst[t].nkept;
  }
  return true;
}

}  // namespace talc
