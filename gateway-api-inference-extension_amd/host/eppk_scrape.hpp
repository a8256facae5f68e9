// eppk_scrape.hpp — the data-source half of the snapshot producer: poll every endpoint's /metrics, hand the bodies to collectors.
//
// The reference's data layer proposal (docs/proposals/1023-data-layer-architecture/README.md:106-163) splits the metrics scraper into
// a DataSource (Type / Start / Stop / Subscribe / UpdateEndpoints: tracks the endpoints and notifies collectors with fresh data) and
// DataCollection plugins (Extract(endpoint, data)).  This header mirrors those two interfaces in C++:
//   * MetricsDataSource: an HTTP/1.1 GET of `http://address:port/metrics` per tracked endpoint, all endpoints of a round in flight
//     together on one thread (HttpGetMany: non-blocking sockets in one poll() set), every `interval`; a failed or timed-out scrape is delivered as an error, not dropped (the collector decides);
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

// ---- the HTTP engine: many GETs in flight on ONE thread ----------------------------------------------------------------------------
// A pool of 4096 endpoints scraped every 50 ms is 80 000 exchanges a second, each of them mostly waiting for the peer: a thread per
// exchange (or a small pool of blocking GETs) spends its time in context switches.  HttpGetMany keeps up to `max_inflight`
// non-blocking sockets in one poll() set and advances each exchange's little state machine (connect -> send -> receive) as its
// socket becomes ready; every exchange has its own deadline.  HttpGet is the one-exchange case of the same code.

struct HttpRequest { std::string host, port, path; };
struct HttpResult { bool ok = false; std::string body, error; SteadyClock::time_point finished; };

namespace detail {

struct Exchange {
  enum State { Idle, Connecting, Sending, Receiving, Done } st = Idle;
  size_t index = 0;
  int fd = -1;
  std::vector<sockaddr_storage> addrs;
  std::vector<socklen_t> addr_lens;
  size_t next_addr = 0;
  std::string req, raw, last_error;
  size_t sent = 0, head_end = std::string::npos, content_len = std::string::npos;
  bool chunked = false;
  SteadyClock::time_point deadline;
};

inline void CloseFd(Exchange& x) { if (x.fd >= 0) { close(x.fd); x.fd = -1; } }

inline void Finish(Exchange& x, HttpResult* r, bool ok, std::string err) {
  CloseFd(x);
  x.st = Exchange::Done;
  r->ok = ok;
  r->error = std::move(err);
  r->finished = SteadyClock::now();
}

// response complete -> status line, entity (content-length or chunked)
inline void Decode(Exchange& x, HttpResult* r) {
  const std::string& raw = x.raw;
  if (x.head_end == std::string::npos) return Finish(x, r, false, raw.empty() ? "receive: connection closed" : "receive: no response header");
  if (raw.compare(0, 5, "HTTP/") != 0) return Finish(x, r, false, "receive: not an HTTP response");
  const size_t sp = raw.find(' ');
  const int status = sp == std::string::npos ? 0 : std::atoi(raw.c_str() + sp + 1);
  if (status != 200) return Finish(x, r, false, "status " + std::to_string(status));
  if (x.chunked) {
    r->body.clear();
    size_t i = x.head_end;
    for (;;) {
      const size_t e = raw.find("\r\n", i);
      if (e == std::string::npos) return Finish(x, r, false, "receive: truncated chunk header");
      const size_t len = (size_t)std::strtoull(raw.c_str() + i, nullptr, 16);
      if (len == 0) break;
      if (e + 2 + len > raw.size()) return Finish(x, r, false, "receive: truncated chunk");
      r->body.append(raw, e + 2, len);
      i = e + 2 + len + 2;
    }
  } else {
    if (x.content_len != std::string::npos && raw.size() - x.head_end < x.content_len) return Finish(x, r, false, "receive: truncated body");
    r->body.assign(raw, x.head_end, x.content_len == std::string::npos ? std::string::npos : x.content_len);
  }
  Finish(x, r, true, "");
}

inline bool ResponseComplete(const Exchange& x) {
  if (x.head_end == std::string::npos) return false;
  if (x.content_len != std::string::npos) return x.raw.size() - x.head_end >= x.content_len;
  return x.chunked && x.raw.size() >= x.head_end + 5 && x.raw.compare(x.raw.size() - 5, 5, "0\r\n\r\n") == 0 &&
         (x.raw.size() == x.head_end + 5 || x.raw.compare(x.raw.size() - 7, 2, "\r\n") == 0);
}

// open the next address of the exchange; false when none is left (the result is then final)
inline bool Connect(Exchange& x, HttpResult* r) {
  while (x.next_addr < x.addrs.size()) {
    const size_t a = x.next_addr++;
    x.fd = socket(x.addrs[a].ss_family, SOCK_STREAM | SOCK_NONBLOCK | SOCK_CLOEXEC, 0);
    if (x.fd < 0) { x.last_error = std::string("socket: ") + std::strerror(errno); continue; }
    int one = 1;
    setsockopt(x.fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
    const int rc = connect(x.fd, (const sockaddr*)&x.addrs[a], x.addr_lens[a]);
    if (rc == 0) { x.st = Exchange::Sending; return true; }
    if (errno == EINPROGRESS) { x.st = Exchange::Connecting; return true; }
    x.last_error = std::string("connect: ") + std::strerror(errno);
    CloseFd(x);
  }
  Finish(x, r, false, x.last_error.empty() ? "no address" : x.last_error);
  return false;
}

inline void Start(Exchange& x, const HttpRequest& q, int timeout_ms, HttpResult* r) {
  x.deadline = SteadyClock::now() + std::chrono::milliseconds(timeout_ms);
  addrinfo hints{}, *res = nullptr;
  hints.ai_family = AF_UNSPEC;
  hints.ai_socktype = SOCK_STREAM;
  hints.ai_flags = AI_NUMERICSERV | AI_NUMERICHOST;            // pod IPs: no resolver round trip
  int rc = getaddrinfo(q.host.c_str(), q.port.c_str(), &hints, &res);
  if (rc == EAI_NONAME) {                                       // a name: the (blocking) resolver
    hints.ai_flags = AI_NUMERICSERV;
    rc = getaddrinfo(q.host.c_str(), q.port.c_str(), &hints, &res);
  }
  if (rc != 0 || !res) return Finish(x, r, false, std::string("resolve: ") + gai_strerror(rc));
  for (addrinfo* a = res; a; a = a->ai_next) {
    sockaddr_storage ss{};
    std::memcpy(&ss, a->ai_addr, a->ai_addrlen);
    x.addrs.push_back(ss);
    x.addr_lens.push_back(a->ai_addrlen);
  }
  freeaddrinfo(res);
  const std::string hostport = q.host.find(':') != std::string::npos ? "[" + q.host + "]:" + q.port : q.host + ":" + q.port;
  x.req = "GET " + q.path + " HTTP/1.1\r\nHost: " + hostport + "\r\nAccept: text/plain\r\nConnection: close\r\n\r\n";
  Connect(x, r);
}

// the socket is ready (or an error is pending): advance as far as it goes without blocking
inline void Advance(Exchange& x, HttpResult* r) {
  if (x.st == Exchange::Connecting) {
    int e = 0;
    socklen_t l = sizeof e;
    getsockopt(x.fd, SOL_SOCKET, SO_ERROR, &e, &l);
    if (e) { x.last_error = std::string("connect: ") + std::strerror(e); CloseFd(x); if (!Connect(x, r)) return; if (x.st == Exchange::Connecting) return; }
    else x.st = Exchange::Sending;
  }
  if (x.st == Exchange::Sending) {
    while (x.sent < x.req.size()) {
      const ssize_t n = send(x.fd, x.req.data() + x.sent, x.req.size() - x.sent, MSG_NOSIGNAL);
      if (n < 0) { if (errno == EAGAIN || errno == EWOULDBLOCK) return; if (errno == EINTR) continue; return Finish(x, r, false, std::string("send: ") + std::strerror(errno)); }
      x.sent += (size_t)n;
    }
    x.st = Exchange::Receiving;
  }
  if (x.st == Exchange::Receiving) {
    char buf[16384];
    for (;;) {
      const ssize_t n = recv(x.fd, buf, sizeof buf, 0);
      if (n < 0) { if (errno == EAGAIN || errno == EWOULDBLOCK) return; if (errno == EINTR) continue; return Finish(x, r, false, std::string("receive: ") + std::strerror(errno)); }
      if (n == 0) return Decode(x, r);
      x.raw.append(buf, (size_t)n);
      if (x.raw.size() > (64u << 20)) return Finish(x, r, false, "receive: body over 64 MiB");
      if (x.head_end == std::string::npos) {
        const size_t h = x.raw.find("\r\n\r\n");
        if (h != std::string::npos) {
          x.head_end = h + 4;
          std::string head = x.raw.substr(0, h + 2);
          for (char& c : head) if (c >= 'A' && c <= 'Z') c = (char)(c - 'A' + 'a');
          if (size_t p2 = head.find("\r\ncontent-length:"); p2 != std::string::npos) x.content_len = (size_t)std::strtoull(head.c_str() + p2 + 17, nullptr, 10);
          if (size_t p3 = head.find("\r\ntransfer-encoding:"); p3 != std::string::npos) x.chunked = head.find("chunked", p3) < head.find("\r\n", p3 + 2);
        }
      }
      if (ResponseComplete(x)) return Decode(x, r);
    }
  }
}

}  // namespace detail

// All requests, at most `max_inflight` at a time, each within `timeout_ms` of its own start.  One thread, one poll() set.
inline std::vector<HttpResult> HttpGetMany(const std::vector<HttpRequest>& reqs, int timeout_ms, size_t max_inflight = 512) {
  std::vector<HttpResult> out(reqs.size());
  std::vector<detail::Exchange> active;
  std::vector<pollfd> fds;
  size_t next = 0;
  if (max_inflight == 0) max_inflight = 1;
  while (next < reqs.size() || !active.empty()) {
    while (active.size() < max_inflight && next < reqs.size()) {
      detail::Exchange x;
      x.index = next;
      detail::Start(x, reqs[next], timeout_ms, &out[next]);
      if (x.st == detail::Exchange::Sending) detail::Advance(x, &out[next]);        // connected at once (loopback)
      if (x.st != detail::Exchange::Done) active.push_back(std::move(x));
      ++next;
    }
    if (active.empty()) continue;
    fds.resize(active.size());
    auto now = SteadyClock::now();
    auto soonest = active[0].deadline;
    for (size_t i = 0; i < active.size(); ++i) {
      fds[i] = pollfd{active[i].fd, (short)(active[i].st == detail::Exchange::Receiving ? POLLIN : POLLOUT), 0};
      if (active[i].deadline < soonest) soonest = active[i].deadline;
    }
    const auto wait = std::chrono::duration_cast<std::chrono::milliseconds>(soonest - now).count();
    const int rc = poll(fds.data(), (nfds_t)fds.size(), wait > 0 ? (int)wait + 1 : 0);
    now = SteadyClock::now();
    for (size_t i = 0; i < active.size(); ++i) {
      detail::Exchange& x = active[i];
      HttpResult* r = &out[x.index];
      if (rc > 0 && fds[i].revents) detail::Advance(x, r);
      if (x.st != detail::Exchange::Done && now >= x.deadline)
        detail::Finish(x, r, false, x.st == detail::Exchange::Connecting ? "connect: timed out" : x.st == detail::Exchange::Sending ? "send: timed out" : "receive: timed out");
    }
    size_t w = 0;
    for (size_t i = 0; i < active.size(); ++i)
      if (active[i].st != detail::Exchange::Done) { if (w != i) active[w] = std::move(active[i]); ++w; }
    active.resize(w);
  }
  return out;
}

// One GET with a deadline for the whole exchange.  true: status 200 and *body holds the decoded entity.
inline bool HttpGet(const std::string& host, const std::string& port, const std::string& path, int timeout_ms, std::string* body,
                    std::string* err) {
  std::vector<HttpResult> r = HttpGetMany({HttpRequest{host, port, path}}, timeout_ms, 1);
  if (r[0].ok) { if (body) *body = std::move(r[0].body); return true; }
  if (err) *err = r[0].error;
  return false;
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
  struct Options { int interval_ms = 50, timeout_ms = 1000; size_t max_inflight = 512; };   // (below the usual 1024 descriptors of a process)
  explicit MetricsDataSource(Options o) : opt_(o) {}
  MetricsDataSource() : MetricsDataSource(Options()) {}
  ~MetricsDataSource() { Stop(); }
  std::string Type() const { return "metrics"; }
  void Subscribe(std::shared_ptr<DataCollection> c) { std::lock_guard<std::mutex> g(mu_); collectors_.push_back(std::move(c)); }
  void UpdateEndpoints(std::vector<ScrapeTarget> eps) { std::lock_guard<std::mutex> g(mu_); targets_ = std::move(eps); }
  // One round now, on the caller's thread: every tracked endpoint fetched once, every collector notified.  Returns the
  // number of successful scrapes.
  size_t ScrapeOnce() {
    std::vector<ScrapeTarget> eps;
    std::vector<std::shared_ptr<DataCollection>> cs;
    { std::lock_guard<std::mutex> g(mu_); eps = targets_; cs = collectors_; }
    std::vector<HttpRequest> reqs;
    reqs.reserve(eps.size());
    for (const ScrapeTarget& t : eps) reqs.push_back({t.address, t.port, t.path});
    std::vector<HttpResult> got = HttpGetMany(reqs, opt_.timeout_ms, opt_.max_inflight);
    std::vector<ScrapeData> out(eps.size());
    size_t good = 0;
    for (size_t i = 0; i < eps.size(); ++i) {
      out[i].ok = got[i].ok;
      out[i].body = std::move(got[i].body);
      out[i].error = std::move(got[i].error);
      out[i].taken = got[i].finished;
      good += out[i].ok;
    }
    for (size_t i = 0; i < eps.size(); ++i) for (auto& c : cs) c->Extract(eps[i], out[i]);
    rounds_.fetch_add(1);
    return good;
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
    std::thread t;
    { std::lock_guard<std::mutex> g(mu_); stop_ = true; t = std::move(th_); }
    cv_.notify_all();
    if (t.joinable()) t.join();
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
