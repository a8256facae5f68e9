// eppk_scrape.hpp — the data-source half of the snapshot producer: poll every endpoint's /metrics, hand the bodies to collectors.
//
// The reference's data layer proposal (docs/proposals/1023-data-layer-architecture/README.md:106-163) splits the metrics scraper into
// a DataSource (Type / Start / Stop / Subscribe / UpdateEndpoints: tracks the endpoints and notifies collectors with fresh data) and
// DataCollection plugins (Extract(endpoint, data)).  This header mirrors those two interfaces in C++:
//   * MetricsDataSource: an HTTP/1.1 GET of `http://address:port/metrics` per tracked endpoint, all endpoints concurrently on a small
//     worker pool, every `interval`; a failed or timed-out scrape is delivered as an error, not dropped (the collector decides);
//   * PodRowCollector: Extract = eppk_metrics.hpp's ParseModelServerMetrics; keeps the latest row per endpoint and the time it was
//     taken; Rows(ids, max_age) returns them in candidate-index order, a stale or missing one as a hole (EPPK_POD_INACTIVE).
// What consumes the rows is eppk_snapshot_publish (GpuPicker::PublishSnapshot in eppk_host.hpp).  Plain POSIX sockets: no TLS (the
// model-server protocol's metrics endpoint is plain HTTP inside the cluster), no redirects, no keep-alive (one short GET per scrape).
#pragma once

#include <arpa/inet.h>
#include <fcntl.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <sys/socket.h>
#include <unistd.h>

#include <atomic>
#include <cerrno>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <unordered_map>

#include "eppk_metrics.hpp"

namespace eppk_host {

using SteadyClock = std::chrono::steady_clock;

// One blocking GET with a deadline for the whole exchange.  true: status 200 and *body holds the decoded entity.
inline bool HttpGet(const std::string& host, const std::string& port, const std::string& path, int timeout_ms, std::string* body,
                    std::string* err) {
  auto fail = [&](const std::string& m) { if (err) *err = m; return false; };
  const auto deadline = SteadyClock::now() + std::chrono::milliseconds(timeout_ms);
  auto left_ms = [&]() -> int {
    const auto d = std::chrono::duration_cast<std::chrono::milliseconds>(deadline - SteadyClock::now()).count();
    return d > 0 ? (int)d : 0;
  };
  addrinfo hints{}, *res = nullptr;
  hints.ai_family = AF_UNSPEC;
  hints.ai_socktype = SOCK_STREAM;
  hints.ai_flags = AI_NUMERICSERV;
  if (int rc = getaddrinfo(host.c_str(), port.c_str(), &hints, &res); rc != 0 || !res) return fail(std::string("resolve: ") + gai_strerror(rc));
  std::unique_ptr<addrinfo, decltype(&freeaddrinfo)> guard(res, freeaddrinfo);
  int fd = -1;
  std::string last = "no address";
  for (addrinfo* a = res; a; a = a->ai_next) {
    fd = socket(a->ai_family, a->ai_socktype | SOCK_NONBLOCK | SOCK_CLOEXEC, a->ai_protocol);
    if (fd < 0) { last = std::string("socket: ") + std::strerror(errno); continue; }
    int rc = connect(fd, a->ai_addr, a->ai_addrlen);
    if (rc != 0 && errno == EINPROGRESS) {
      pollfd p{fd, POLLOUT, 0};
      rc = poll(&p, 1, left_ms());
      if (rc == 1) { int e = 0; socklen_t l = sizeof e; getsockopt(fd, SOL_SOCKET, SO_ERROR, &e, &l); rc = e ? -1 : 0; errno = e; }
      else { errno = rc == 0 ? ETIMEDOUT : errno; rc = -1; }
    }
    if (rc == 0) break;
    last = std::string("connect: ") + std::strerror(errno);
    close(fd);
    fd = -1;
  }
  if (fd < 0) return fail(last);
  struct Closer { int fd; ~Closer() { close(fd); } } closer{fd};
  int one = 1;
  setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
  const std::string hostport = host.find(':') != std::string::npos ? "[" + host + "]:" + port : host + ":" + port;
  const std::string req = "GET " + path + " HTTP/1.1\r\nHost: " + hostport + "\r\nAccept: text/plain\r\nConnection: close\r\n\r\n";
  for (size_t off = 0; off < req.size();) {
    pollfd p{fd, POLLOUT, 0};
    if (poll(&p, 1, left_ms()) != 1) return fail("send: timed out");
    const ssize_t n = send(fd, req.data() + off, req.size() - off, MSG_NOSIGNAL);
    if (n < 0) { if (errno == EAGAIN || errno == EINTR) continue; return fail(std::string("send: ") + std::strerror(errno)); }
    off += (size_t)n;
  }
  std::string raw;
  size_t head_end = std::string::npos, content_len = std::string::npos;
  bool chunked = false;
  char buf[16384];
  for (;;) {
    if (head_end != std::string::npos && content_len != std::string::npos && raw.size() - head_end >= content_len) break;
    if (head_end != std::string::npos && chunked && raw.find("\r\n0\r\n", head_end - 2) != std::string::npos &&
        raw.compare(raw.size() - 4, 4, "\r\n\r\n") == 0) break;
    pollfd p{fd, POLLIN, 0};
    if (poll(&p, 1, left_ms()) != 1) return fail("receive: timed out");
    const ssize_t n = recv(fd, buf, sizeof buf, 0);
    if (n < 0) { if (errno == EAGAIN || errno == EINTR) continue; return fail(std::string("receive: ") + std::strerror(errno)); }
    if (n == 0) break;
    raw.append(buf, (size_t)n);
    if (raw.size() > (64u << 20)) return fail("receive: body over 64 MiB");
    if (head_end == std::string::npos) {
      const size_t h = raw.find("\r\n\r\n");
      if (h == std::string::npos) continue;
      head_end = h + 4;
      std::string head = raw.substr(0, h + 2);
      for (char& c : head) if (c >= 'A' && c <= 'Z') c = (char)(c - 'A' + 'a');
      if (size_t p2 = head.find("\r\ncontent-length:"); p2 != std::string::npos) content_len = (size_t)std::strtoull(head.c_str() + p2 + 17, nullptr, 10);
      if (size_t p3 = head.find("\r\ntransfer-encoding:"); p3 != std::string::npos) chunked = head.find("chunked", p3) < head.find("\r\n", p3 + 2);
    }
  }
  if (head_end == std::string::npos) return fail("receive: no response header");
  if (raw.compare(0, 5, "HTTP/") != 0) return fail("receive: not an HTTP response");
  const size_t sp = raw.find(' ');
  const int status = sp == std::string::npos ? 0 : std::atoi(raw.c_str() + sp + 1);
  if (status != 200) return fail("status " + std::to_string(status));
  if (chunked) {
    body->clear();
    size_t i = head_end;
    for (;;) {
      const size_t e = raw.find("\r\n", i);
      if (e == std::string::npos) return fail("receive: truncated chunk header");
      const size_t len = (size_t)std::strtoull(raw.c_str() + i, nullptr, 16);
      if (len == 0) break;
      if (e + 2 + len > raw.size()) return fail("receive: truncated chunk");
      body->append(raw, e + 2, len);
      i = e + 2 + len + 2;
    }
  } else {
    if (content_len != std::string::npos && raw.size() - head_end < content_len) return fail("receive: truncated body");
    body->assign(raw, head_end, content_len == std::string::npos ? std::string::npos : content_len);
  }
  return true;
}

struct ScrapeTarget {          // what the data source tracks: an endpoint id (candidate identity) and where its metrics live
  std::string id, address, port, path = "/metrics";
};

struct ScrapeData {            // `data interface{}` of DataCollection.Extract for the "metrics" source
  bool ok = false;
  std::string body, error;
  SteadyClock::time_point taken;
};

class DataCollection {         // 1023-…/README.md:107-115
 public:
  virtual ~DataCollection() = default;
  virtual void Extract(const ScrapeTarget& ep, const ScrapeData& data) = 0;
};

class MetricsDataSource {      // 1023-…/README.md:143-163 (DataSource)
 public:
  struct Options { int interval_ms = 50, timeout_ms = 1000; unsigned workers = 8; };
  explicit MetricsDataSource(Options o) : opt_(o) {}
  MetricsDataSource() : MetricsDataSource(Options()) {}
  ~MetricsDataSource() { Stop(); }
  std::string Type() const { return "metrics"; }
  void Subscribe(std::shared_ptr<DataCollection> c) { std::lock_guard<std::mutex> g(mu_); collectors_.push_back(std::move(c)); }
  void UpdateEndpoints(std::vector<ScrapeTarget> eps) { std::lock_guard<std::mutex> g(mu_); targets_ = std::move(eps); }
  // One round now, on the caller's thread + the pool: every tracked endpoint fetched once, every collector notified.  Returns the
  // number of successful scrapes.
  size_t ScrapeOnce() {
    std::vector<ScrapeTarget> eps;
    std::vector<std::shared_ptr<DataCollection>> cs;
    { std::lock_guard<std::mutex> g(mu_); eps = targets_; cs = collectors_; }
    std::vector<ScrapeData> out(eps.size());
    std::atomic<size_t> next{0}, good{0};
    auto work = [&] {
      for (size_t i; (i = next.fetch_add(1)) < eps.size();) {
        out[i].ok = HttpGet(eps[i].address, eps[i].port, eps[i].path, opt_.timeout_ms, &out[i].body, &out[i].error);
        out[i].taken = SteadyClock::now();
        good += out[i].ok;
      }
    };
    const unsigned nw = (unsigned)std::min<size_t>(opt_.workers ? opt_.workers : 1, eps.size());
    std::vector<std::thread> pool;
    for (unsigned w = 1; w < nw; ++w) pool.emplace_back(work);
    work();
    for (auto& t : pool) t.join();
    for (size_t i = 0; i < eps.size(); ++i) for (auto& c : cs) c->Extract(eps[i], out[i]);
    rounds_.fetch_add(1);
    return good.load();
  }
  void Start() {
    std::lock_guard<std::mutex> g(mu_);
    if (th_.joinable()) return;
    stop_ = false;
    th_ = std::thread([this] {
      std::unique_lock<std::mutex> lk(mu_);
      while (!stop_) {
        lk.unlock();
        const auto t0 = SteadyClock::now();
        ScrapeOnce();
        lk.lock();
        cv_.wait_until(lk, t0 + std::chrono::milliseconds(opt_.interval_ms), [this] { return stop_; });
      }
    });
  }
  void Stop() {
    { std::lock_guard<std::mutex> g(mu_); stop_ = true; }
    cv_.notify_all();
    if (th_.joinable()) th_.join();
  }
  uint64_t rounds() const { return rounds_.load(); }

 private:
  Options opt_;
  std::mutex mu_;
  std::condition_variable cv_;
  std::vector<ScrapeTarget> targets_;
  std::vector<std::shared_ptr<DataCollection>> collectors_;
  std::thread th_;
  bool stop_ = false;
  std::atomic<uint64_t> rounds_{0};
};

class PodRowCollector : public DataCollection {
 public:
  PodRowCollector(std::map<std::string, int32_t> adapter_ids, MetricNames names = MetricNames())
      : adapter_ids_(std::move(adapter_ids)), names_(std::move(names)) {}
  void Extract(const ScrapeTarget& ep, const ScrapeData& data) override {
    Entry e;
    e.taken = data.taken;
    if (data.ok) { e.result = ParseModelServerMetrics(data.body, adapter_ids_, names_); e.usable = e.result.complete(); }
    else e.error = data.error;
    std::lock_guard<std::mutex> g(mu_);
    if (!e.usable) {                     // keep the last good row (it ages out through max_age); remember why the new one failed
      auto it = latest_.find(ep.id);
      if (it != latest_.end() && it->second.usable) { it->second.error = e.error.empty() ? "incomplete scrape" : e.error; ++failures_; return; }
      ++failures_;
    }
    latest_[ep.id] = std::move(e);
  }
  // Rows in the order of `ids` (candidate-index order).  No usable scrape younger than max_age_ms -> a hole.
  std::vector<eppk_pod_row> Rows(const std::vector<std::string>& ids, int max_age_ms, uint32_t* n_holes = nullptr) const {
    std::vector<eppk_pod_row> rows(ids.size());
    const auto now = SteadyClock::now();
    uint32_t holes = 0;
    std::lock_guard<std::mutex> g(mu_);
    for (size_t i = 0; i < ids.size(); ++i) {
      auto it = latest_.find(ids[i]);
      const bool fresh = it != latest_.end() && it->second.usable &&
                         (max_age_ms <= 0 || now - it->second.taken <= std::chrono::milliseconds(max_age_ms));
      if (fresh) rows[i] = it->second.result.row;
      else { rows[i] = eppk_pod_row{}; rows[i].flags = EPPK_POD_INACTIVE; ++holes; }
    }
    if (n_holes) *n_holes = holes;
    return rows;
  }
  bool Latest(const std::string& id, ScrapeResult* out, std::string* error = nullptr) const {
    std::lock_guard<std::mutex> g(mu_);
    auto it = latest_.find(id);
    if (it == latest_.end()) return false;
    if (out) *out = it->second.result;
    if (error) *error = it->second.error;
    return it->second.usable;
  }
  void Forget(const std::string& id) { std::lock_guard<std::mutex> g(mu_); latest_.erase(id); }
  uint64_t failures() const { std::lock_guard<std::mutex> g(mu_); return failures_; }

 private:
  struct Entry { ScrapeResult result; bool usable = false; std::string error; SteadyClock::time_point taken; };
  std::map<std::string, int32_t> adapter_ids_;
  MetricNames names_;
  mutable std::mutex mu_;
  std::unordered_map<std::string, Entry> latest_;
  uint64_t failures_ = 0;
};

}  // namespace eppk_host
