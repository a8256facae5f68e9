// eppk_host.cpp — the host-side steps adjacent to the pick (include/eppk.h, last sections):
// XXH64 block-hash chain, subset-filter bitmask, round-robin fallback.  No HIP, no oracle.
#include <atomic>
#include <cstring>
#include <string>
#include <string_view>
#include <vector>

#include "../../include/eppk.h"

namespace {

// ---- XXH64 (published xxHash64; the Go module the reference lists is cespare/xxhash/v2, go.mod:6) ----
constexpr uint64_t P1 = 0x9E3779B185EBCA87ull, P2 = 0xC2B2AE3D27D4EB4Full, P3 = 0x165667B19E3779F9ull,
                   P4 = 0x85EBCA77C2B2AE63ull, P5 = 0x27D4EB2F165667C5ull;

inline uint64_t load64(const uint8_t* p) { uint64_t v; std::memcpy(&v, p, 8); return v; }  // little-endian host
inline uint32_t load32(const uint8_t* p) { uint32_t v; std::memcpy(&v, p, 4); return v; }
inline uint64_t rotl(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
inline uint64_t lane_round(uint64_t acc, uint64_t in) { return rotl(acc + in * P2, 31) * P1; }
inline uint64_t fold(uint64_t h, uint64_t acc) { return (h ^ lane_round(0, acc)) * P1 + P4; }

uint64_t xxh64(const uint8_t* p, size_t len, uint64_t seed) {
  const uint8_t* const end = p + len;
  uint64_t h;
  if (len >= 32) {
    uint64_t v[4] = {seed + P1 + P2, seed + P2, seed, seed - P1};
    for (; end - p >= 32; p += 32)
      for (int i = 0; i < 4; ++i) v[i] = lane_round(v[i], load64(p + 8 * i));
    h = rotl(v[0], 1) + rotl(v[1], 7) + rotl(v[2], 12) + rotl(v[3], 18);
    for (int i = 0; i < 4; ++i) h = fold(h, v[i]);
  } else {
    h = seed + P5;
  }
  h += (uint64_t)len;
  for (; end - p >= 8; p += 8) h = rotl(h ^ lane_round(0, load64(p)), 27) * P1 + P4;
  if (end - p >= 4) { h = rotl(h ^ ((uint64_t)load32(p) * P1), 23) * P2 + P3; p += 4; }
  for (; p < end; ++p) h = rotl(h ^ ((uint64_t)*p * P5), 11) * P1;
  h = (h ^ (h >> 33)) * P2;
  h = (h ^ (h >> 29)) * P3;
  return h ^ (h >> 32);
}

// ---- subset filter helpers (request.go:104-133) ----
// Go's strings.TrimSpace (request.go:58, :65, :109): leading and trailing Unicode White_Space, on UTF-8 --
// U+0009..000D, U+0020, U+0085, U+00A0, U+1680, U+2000..200A, U+2028, U+2029, U+202F, U+205F, U+3000.
// Length in bytes of the white-space rune that starts at s[i] (0: none).
size_t space_at(std::string_view s, size_t i) {
  const auto b = [&](size_t k) { return k < s.size() ? (unsigned char)s[k] : 0u; };
  const unsigned c = b(i);
  if (c == ' ' || (c >= '\t' && c <= '\r')) return 1;
  if (c == 0xC2 && (b(i + 1) == 0x85 || b(i + 1) == 0xA0)) return 2;
  if (c == 0xE1 && b(i + 1) == 0x9A && b(i + 2) == 0x80) return 3;
  if (c == 0xE2 && b(i + 1) == 0x80 && ((b(i + 2) >= 0x80 && b(i + 2) <= 0x8A) || b(i + 2) == 0xA8 || b(i + 2) == 0xA9 || b(i + 2) == 0xAF)) return 3;
  if (c == 0xE2 && b(i + 1) == 0x81 && b(i + 2) == 0x9F) return 3;
  if (c == 0xE3 && b(i + 1) == 0x80 && b(i + 2) == 0x80) return 3;
  return 0;
}
std::string_view trim(std::string_view s) {
  for (size_t n; !s.empty() && (n = space_at(s, 0)) != 0;) s.remove_prefix(n);
  for (;;) {                         // the last rune: 1, 2 or 3 bytes back
    size_t n = 0;
    for (size_t back = 1; back <= 3 && back <= s.size(); ++back)
      if (space_at(s, s.size() - back) == back) { n = back; break; }
    if (!n) break;
    s.remove_suffix(n);
  }
  return s;
}

// Go net.SplitHostPort acceptance as used at request.go:110.
bool split_host_port(std::string_view s, std::string_view& host, std::string_view& port) {
  const size_t colon = s.rfind(':');
  if (colon == std::string_view::npos) return false;           // missing port
  size_t scan_open = 0, scan_close = 0;
  if (s.front() == '[') {
    const size_t rb = s.find(']');
    if (rb == std::string_view::npos || rb + 1 == s.size() || rb + 1 != colon) return false;
    host = s.substr(1, rb - 1);
    scan_open = 1; scan_close = rb + 1;
  } else {
    host = s.substr(0, colon);
    if (host.find(':') != std::string_view::npos) return false; // too many colons
  }
  if (s.substr(scan_open).find('[') != std::string_view::npos) return false;
  if (s.substr(scan_close).find(']') != std::string_view::npos) return false;
  port = s.substr(colon + 1);
  return true;
}

struct Entry { std::string_view host, port; bool any_port; };

}  // namespace

extern "C" {

uint64_t eppk_xxh64(const void* data, size_t len, uint64_t seed) {
  return xxh64(static_cast<const uint8_t*>(data), len, seed);
}

int eppk_hash_prompt(const uint8_t* model, size_t model_len, const uint8_t* prompt, size_t prompt_len, uint32_t block_chars,
                     uint64_t* out, uint32_t max_out) {
  if (block_chars == 0 || (!prompt && prompt_len) || (!model && model_len) || (!out && max_out)) return EPPK_ERR_ARG;
  std::vector<uint8_t> buf((size_t)block_chars + 8);
  uint64_t prev = xxh64(model, model_len, 0);
  uint32_t n = 0;
  for (size_t off = 0; off + block_chars <= prompt_len && n < max_out; off += block_chars, ++n) {
    std::memcpy(buf.data(), prompt + off, block_chars);
    std::memcpy(buf.data() + block_chars, &prev, 8);  // LE64(h[i-1])
    prev = xxh64(buf.data(), buf.size(), 0);
    out[n] = prev;
  }
  return (int)n;
}

int eppk_subset_mask(const char* const* addrs, const char* const* ports, uint32_t n_pods, const char* filter, uint64_t* out_mask) {
  if (n_pods && (!addrs || !ports || !out_mask)) return EPPK_ERR_ARG;
  const uint32_t words = (n_pods + 63u) / 64u;
  for (uint32_t w = 0; w < words; ++w) out_mask[w] = 0;
  if (!filter) {  // no subset filter: all pods are candidates (request.go:136-137)
    for (uint32_t p = 0; p < n_pods; ++p) out_mask[p >> 6] |= 1ull << (p & 63u);
    return (int)n_pods;
  }
  // parse once: allowAllPorts / allowedPorts of request.go:107-119
  std::vector<Entry> entries;
  std::string_view rest(filter);
  for (;;) {
    const size_t comma = rest.find(',');
    std::string_view e = trim(rest.substr(0, comma));
    if (!e.empty()) {
      Entry en{};
      en.any_port = !split_host_port(e, en.host, en.port);
      if (en.any_port) en.host = e;
      entries.push_back(en);
    }
    if (comma == std::string_view::npos) break;
    rest.remove_prefix(comma + 1);
  }
  int count = 0;
  for (uint32_t p = 0; p < n_pods; ++p) {
    if (!addrs[p] || !ports[p]) return EPPK_ERR_ARG;
    const std::string_view a(addrs[p]), pt(ports[p]);
    bool ok = false;
    for (const Entry& en : entries)
      if (en.host == a && (en.any_port || en.port == pt)) { ok = true; break; }
    if (ok) { out_mask[p >> 6] |= 1ull << (p & 63u); ++count; }
  }
  return count;  // 0 => fail closed (request_test.go:335-369, :407-439)
}

// ---- subset filter on the device: the host half (tokenise + fingerprint) ------------------------------------------------
// An address (or address + port) is represented by a 128-bit fingerprint: XXH64 of its bytes under two seeds.  Exact-port
// entries hash host, a NUL byte, port -- no header value contains NUL, so an "all ports" entry can never alias one.
void eppk_addr_fingerprint(const char* host, size_t host_len, const char* port, size_t port_len, uint64_t out[2]) {
  std::string buf(host ? host : "", host ? host_len : 0);
  if (port) { buf.push_back('\0'); buf.append(port, port_len); }
  out[0] = xxh64((const uint8_t*)buf.data(), buf.size(), 0);
  out[1] = xxh64((const uint8_t*)buf.data(), buf.size(), 0x9E3779B97F4A7C15ull);
  if ((out[0] | out[1]) == 0ull) out[1] = 1ull;      // (0, 0) is the "no filter" entry
}

int eppk_subset_entries(const char* filter, uint64_t* out_keys, uint32_t cap) {
  if (!out_keys && cap) return EPPK_ERR_ARG;
  if (!filter) {                      // no subset filter: one entry that admits every pod (request.go:136-137)
    if (cap >= 1) { out_keys[0] = 0; out_keys[1] = 0; }
    return 1;
  }
  uint32_t n = 0;
  std::string_view rest(filter);
  for (;;) {
    const size_t comma = rest.find(',');
    const std::string_view e = trim(rest.substr(0, comma));
    if (!e.empty()) {                 // (same entry rules as eppk_subset_mask: request.go:107-119)
      std::string_view host, port;
      uint64_t fp[2];
      if (split_host_port(e, host, port)) eppk_addr_fingerprint(host.data(), host.size(), port.data(), port.size(), fp);
      else eppk_addr_fingerprint(e.data(), e.size(), nullptr, 0, fp);
      if (n < cap) { out_keys[2 * (size_t)n] = fp[0]; out_keys[2 * (size_t)n + 1] = fp[1]; }
      ++n;
    }
    if (comma == std::string_view::npos) break;
    rest.remove_prefix(comma + 1);
  }
  return (int)n;
}

int32_t eppk_round_robin(uint64_t* counter, uint32_t n_candidates) {
  if (!counter || n_candidates == 0) return EPPK_NO_PICK;  // server.go:91-93
  const uint64_t idx = __atomic_add_fetch(counter, 1, __ATOMIC_SEQ_CST);  // server.go:95
  return (int32_t)(idx % n_candidates);                     // server.go:96
}

}  // extern "C"
