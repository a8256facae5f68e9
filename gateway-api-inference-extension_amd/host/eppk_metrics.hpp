// eppk_metrics.hpp — the snapshot producer's input side: model-server metrics -> eppk_pod_row.
//
// the kernel scores 64-byte pod rows (include/eppk.h: eppk_pod_row); the reference's datastore.Endpoint carries
// identity only (pkg/lwepp/datastore/datastore.go:40-46) and its metrics scraper is Go (docs/proposals/1023-data-layer-architecture/
// README.md:106-163: a data source fetches the model server's Prometheus endpoint, extractors turn the body into typed attributes).
// This is the C++ twin of the EXTRACTOR half: the text a scrape returns -> the gauges of the model-server protocol
// (docs/proposals/003-model-server-protocol/README.md:28-57) -> one eppk_pod_row.  The fetch (HTTP GET /metrics per endpoint on a
// timer) stays the gateway's; what it hands over is a string.
//
//   * metric names are configuration, as the protocol says ("the exact metric names don't necessarily need to be the same"): a
//     MetricSpec is `name` or `name{label=value}` (the Triton forms of the table, README.md:30-34); defaults are vLLM's;
//   * TotalQueuedRequests / TotalRunningRequests / KVCacheUtilization: gauges; several series of one name (per-model labels): summed
//     for the two counts, maximum for the utilisation;
//   * LoRA (README.md:46-57): gauge `vllm:lora_requests_info`, VALUE = last-updated timestamp, labels max_lora /
//     running_lora_adapters / waiting_lora_adapters (comma separated): the series with the greatest value is the current one;
//     adapter names -> ids 0..127 through the caller's table (the same table the request rows use); unknown names are reported;
//   * optional cache_config_info labels block_size / num_gpu_blocks are returned beside the row (prefix-scorer configuration).
// Parsing follows the Prometheus text exposition format: `name{l1="v1",l2="v2"} value [timestamp]`, `#` comment lines, escapes
// \\ \" \n inside label values, NaN / +Inf / -Inf values.  Malformed lines are counted and skipped, never fatal.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <string_view>
#include <utility>
#include <vector>

#include "../../include/eppk.h"

namespace eppk_host {

struct MetricSpec {            // `name` or `name{label=value}`
  std::string name, label, value;
  static MetricSpec Parse(std::string_view s) {
    MetricSpec m;
    const size_t b = s.find('{');
    if (b == std::string_view::npos) { m.name = std::string(s); return m; }
    m.name = std::string(s.substr(0, b));
    std::string_view in = s.substr(b + 1);
    if (!in.empty() && in.back() == '}') in.remove_suffix(1);
    const size_t e = in.find('=');
    if (e != std::string_view::npos) {
      m.label = std::string(in.substr(0, e));
      std::string_view v = in.substr(e + 1);
      if (v.size() >= 2 && v.front() == '"' && v.back() == '"') { v.remove_prefix(1); v.remove_suffix(1); }
      m.value = std::string(v);
    }
    return m;
  }
};

struct MetricNames {           // defaults: vLLM (003-…/README.md:28-34, 46-57)
  MetricSpec queued = MetricSpec::Parse("vllm:num_requests_waiting");
  MetricSpec running = MetricSpec::Parse("vllm:num_requests_running");
  MetricSpec kv_util = MetricSpec::Parse("vllm:kv_cache_usage_perc");
  std::string lora_info = "vllm:lora_requests_info";
  std::string cache_info = "vllm:cache_config_info";
  std::string block_size_label = "block_size", num_blocks_label = "num_gpu_blocks";
};

struct Sample {                // one line of the exposition format
  std::string name;
  std::vector<std::pair<std::string, std::string>> labels;
  double value = 0.0;
  const std::string* Label(std::string_view k) const {
    for (const auto& kv : labels) if (kv.first == k) return &kv.second;
    return nullptr;
  }
};

// Parse one line; false for comments, blank and malformed lines (`malformed` says which).
inline bool ParseSample(std::string_view line, Sample* out, bool* malformed = nullptr) {
  if (malformed) *malformed = false;
  auto bad = [&] { if (malformed) *malformed = true; return false; };
  while (!line.empty() && (line.back() == '\r' || line.back() == ' ' || line.back() == '\t')) line.remove_suffix(1);
  size_t i = 0;
  while (i < line.size() && (line[i] == ' ' || line[i] == '\t')) ++i;
  if (i == line.size() || line[i] == '#') return false;
  const size_t n0 = i;
  while (i < line.size() && line[i] != '{' && line[i] != ' ' && line[i] != '\t') ++i;
  if (i == n0) return bad();
  out->name.assign(line.substr(n0, i - n0));
  out->labels.clear();
  if (i < line.size() && line[i] == '{') {
    ++i;
    for (;;) {
      while (i < line.size() && (line[i] == ' ' || line[i] == ',')) ++i;
      if (i < line.size() && line[i] == '}') { ++i; break; }
      const size_t k0 = i;
      while (i < line.size() && line[i] != '=' && line[i] != '}') ++i;
      if (i >= line.size() || line[i] != '=') return bad();
      std::string key(line.substr(k0, i - k0));
      while (!key.empty() && key.back() == ' ') key.pop_back();
      ++i;
      if (i >= line.size() || line[i] != '"') return bad();
      ++i;
      std::string val;
      bool closed = false;
      while (i < line.size()) {
        const char ch = line[i++];
        if (ch == '\\' && i < line.size()) { const char e = line[i++]; val.push_back(e == 'n' ? '\n' : e); }
        else if (ch == '"') { closed = true; break; }
        else val.push_back(ch);
      }
      if (!closed) return bad();
      out->labels.emplace_back(std::move(key), std::move(val));
    }
  }
  while (i < line.size() && (line[i] == ' ' || line[i] == '\t')) ++i;
  if (i == line.size()) return bad();
  const size_t v0 = i;
  while (i < line.size() && line[i] != ' ' && line[i] != '\t') ++i;
  const std::string vs(line.substr(v0, i - v0));
  if (vs == "NaN") out->value = std::nan("");
  else if (vs == "+Inf" || vs == "Inf") out->value = HUGE_VAL;
  else if (vs == "-Inf") out->value = -HUGE_VAL;
  else {
    char* end = nullptr;
    out->value = std::strtod(vs.c_str(), &end);
    if (end == vs.c_str() || *end != '\0') return bad();
  }
  return true;      // (an optional timestamp behind the value is ignored)
}

struct ScrapeResult {
  eppk_pod_row row{};                       // flags = 0: an active slot
  bool has_queue = false, has_running = false, has_kv = false, has_lora = false;
  uint32_t block_size = 0, num_gpu_blocks = 0;          // 0 = not reported (fall back to the prefix plugin's configuration)
  std::vector<std::string> unknown_adapters;            // named by the server, absent from the caller's table: not in the bitsets
  uint32_t malformed_lines = 0;
  bool complete() const { return has_queue && has_kv; } // what the queue / KV scorers need; LoRA metrics are optional in the protocol
};

namespace detail {
inline std::vector<std::string> SplitAdapters(const std::string& s) {      // "adapter1, adapter2" (README.md:55-56)
  std::vector<std::string> out;
  size_t i = 0;
  while (i <= s.size()) {
    size_t j = s.find(',', i);
    if (j == std::string::npos) j = s.size();
    size_t a = i, b = j;
    while (a < b && (s[a] == ' ' || s[a] == '\t')) ++a;
    while (b > a && (s[b - 1] == ' ' || s[b - 1] == '\t')) --b;
    if (b > a) out.emplace_back(s.substr(a, b - a));
    i = j + 1;
  }
  return out;
}
inline bool Matches(const MetricSpec& m, const Sample& s) {
  if (s.name != m.name) return false;
  if (m.label.empty()) return true;
  const std::string* v = s.Label(m.label);
  return v && *v == m.value;
}
inline uint32_t ToCount(double v) { return !(v > 0.0) ? 0u : v >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)(v + 0.5); }
}  // namespace detail

// The body of one model server's /metrics -> its pod row.  `adapter_ids`: adapter name -> id 0..127 (ids beyond are ignored).
inline ScrapeResult ParseModelServerMetrics(std::string_view text, const std::map<std::string, int32_t>& adapter_ids,
                                            const MetricNames& names = MetricNames()) {
  ScrapeResult r;
  double queued = 0.0, running = 0.0, kv = 0.0, lora_stamp = -HUGE_VAL;
  Sample s, lora;
  size_t pos = 0;
  while (pos <= text.size()) {
    size_t nl = text.find('\n', pos);
    if (nl == std::string_view::npos) nl = text.size();
    bool malformed = false;
    if (ParseSample(text.substr(pos, nl - pos), &s, &malformed)) {
      if (detail::Matches(names.queued, s)) { if (!std::isnan(s.value)) { queued += s.value; r.has_queue = true; } }
      else if (detail::Matches(names.running, s)) { if (!std::isnan(s.value)) { running += s.value; r.has_running = true; } }
      else if (detail::Matches(names.kv_util, s)) { if (!std::isnan(s.value)) { kv = r.has_kv ? (s.value > kv ? s.value : kv) : s.value; r.has_kv = true; } }
      else if (s.name == names.lora_info) {
        if (!std::isnan(s.value) && (!r.has_lora || s.value > lora_stamp)) { lora = s; lora_stamp = s.value; r.has_lora = true; }   // the latest series wins
      } else if (s.name == names.cache_info) {
        if (const std::string* b = s.Label(names.block_size_label)) r.block_size = (uint32_t)std::strtoul(b->c_str(), nullptr, 10);
        if (const std::string* n = s.Label(names.num_blocks_label)) r.num_gpu_blocks = (uint32_t)std::strtoul(n->c_str(), nullptr, 10);
      }
    } else if (malformed) {
      ++r.malformed_lines;
    }
    pos = nl + 1;
  }
  r.row.queue = detail::ToCount(queued);
  r.row.running = detail::ToCount(running);
  r.row.kv_util = kv;                        // a fraction in [0, 1] as vLLM reports it; the scorer clamps (SEMANTICS.md §3 KV)
  if (r.has_lora) {
    if (const std::string* m = lora.Label("max_lora")) r.row.max_lora = (uint32_t)std::strtoul(m->c_str(), nullptr, 10);
    auto fill = [&](const char* label, uint64_t (&bits)[2]) {
      const std::string* v = lora.Label(label);
      if (!v) return;
      for (const std::string& a : detail::SplitAdapters(*v)) {
        auto it = adapter_ids.find(a);
        if (it == adapter_ids.end() || it->second < 0 || it->second >= (int32_t)EPPK_MAX_ADAPTERS) {
          bool seen = false;
          for (const std::string& u : r.unknown_adapters) seen = seen || u == a;
          if (!seen) r.unknown_adapters.push_back(a);
          continue;
        }
        bits[it->second >> 6] |= 1ull << (it->second & 63);
      }
    };
    fill("running_lora_adapters", r.row.active);
    fill("waiting_lora_adapters", r.row.waiting);
  }
  return r;
}

// A whole snapshot: the scrape bodies of the endpoints, in candidate-index order; an endpoint whose scrape is missing or lacks the
// queue / KV gauges becomes a HOLE of the snapshot (EPPK_POD_INACTIVE: never a candidate) rather than a pod with made-up gauges.
inline std::vector<eppk_pod_row> BuildPodRows(const std::vector<std::string>& bodies, const std::map<std::string, int32_t>& adapter_ids,
                                              const MetricNames& names = MetricNames(), std::vector<ScrapeResult>* details = nullptr) {
  std::vector<eppk_pod_row> rows(bodies.size());
  if (details) details->clear();
  for (size_t i = 0; i < bodies.size(); ++i) {
    ScrapeResult r = ParseModelServerMetrics(bodies[i], adapter_ids, names);
    rows[i] = r.row;
    if (!r.complete()) { rows[i] = eppk_pod_row{}; rows[i].flags = EPPK_POD_INACTIVE; }
    if (details) details->push_back(std::move(r));
  }
  return rows;
}

}  // namespace eppk_host
