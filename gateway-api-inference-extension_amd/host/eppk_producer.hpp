// eppk_producer.hpp — the snapshot producer: list the pool -> scrape every endpoint -> pod rows -> GpuPicker::PublishSnapshot.
//
// SURVEY.md §8(f)-2.  The reference's Endpoint carries identity only (pkg/lwepp/datastore/datastore.go:40-46) and its pod list is
// re-listed per request (request.go:99, datastore.go:181-193); the picker on the device scores a FROZEN snapshot, so something has
// to produce one: this class, the C++ twin of `SnapshotProducer` in the Go patch (integration/…patch: gpusnapshot.go).
//   every interval:  endpoints = list()                     the pool as the datastore sees it now
//                    MetricsDataSource::ScrapeOnce()         eppk_scrape.hpp: GET /metrics of each, concurrently
//                    PodRowCollector::Rows(max_age)          eppk_metrics.hpp: text -> eppk_pod_row; stale / failed -> not usable
//                    picker.PublishSnapshot(usable endpoints, rows, adapters, ++epoch)
// Run the picker with GpuPickerOptions.stable_slots: an endpoint keeps its candidate index from snapshot to snapshot and an endpoint
// without a usable scrape leaves a hole (the prefix index forgets it), exactly what a departed endpoint does.
// LoRA adapter names are given ids 0..127 in the order the model servers first report them (or take a fixed table).
#pragma once

#include <unordered_set>

#include "eppk_host.hpp"
#include "eppk_scrape.hpp"

namespace eppk_host {

class SnapshotProducer {
 public:
  struct Options {
    MetricsDataSource::Options scrape;        // interval 50 ms, timeout 1 s, 512 exchanges in flight
    int max_age_ms = 2000;                    // a row older than this is not published (its endpoint leaves the snapshot)
    MetricNames names;
    std::map<std::string, int32_t> adapters;  // fixed name -> id table; names beyond it are assigned the next free id
    std::string metrics_path = "/metrics";
  };
  using ListFn = std::function<std::vector<Endpoint>()>;

  SnapshotProducer(GpuPicker* picker, ListFn list, Options o)
      : picker_(picker), list_(std::move(list)), opt_(std::move(o)), src_(opt_.scrape), adapters_(opt_.adapters),
        collector_(std::make_shared<Collector>(this)) {
    src_.Subscribe(collector_);
  }
  SnapshotProducer(GpuPicker* picker, ListFn list) : SnapshotProducer(picker, std::move(list), Options()) {}
  ~SnapshotProducer() { Stop(); }

  // One round on the caller's thread.  `published` (nullable): endpoints in the snapshot this round published.
  Status RefreshOnce(size_t* published = nullptr) {
    std::vector<Endpoint> eps = list_();
    std::vector<ScrapeTarget> targets;
    targets.reserve(eps.size());
    for (const Endpoint& e : eps) targets.push_back({JoinHostPort(e.address, e.port), e.address, e.port, opt_.metrics_path});
    src_.UpdateEndpoints(targets);
    src_.ScrapeOnce();
    std::vector<Endpoint> usable;
    std::vector<eppk_pod_row> rows;
    const auto now = SteadyClock::now();
    std::unordered_map<std::string, int32_t> adapters;
    {
      std::lock_guard<std::mutex> g(mu_);
      for (size_t i = 0; i < eps.size(); ++i) {
        auto it = latest_.find(targets[i].id);
        if (it == latest_.end() || now - it->second.taken > std::chrono::milliseconds(opt_.max_age_ms)) continue;
        usable.push_back(eps[i]);
        rows.push_back(it->second.row);
      }
      std::unordered_set<std::string> listed;
      for (const ScrapeTarget& t : targets) listed.insert(t.id);
      for (auto it = latest_.begin(); it != latest_.end();) it = listed.count(it->first) ? std::next(it) : latest_.erase(it);   // endpoints that left the pool
      adapters.insert(adapters_.begin(), adapters_.end());
    }
    if (published) *published = usable.size();
    return picker_->PublishSnapshot(usable, rows, adapters, ++epoch_);
  }

  void Start() {
    std::lock_guard<std::mutex> g(run_mu_);
    if (th_.joinable()) return;
    stop_ = false;
    th_ = std::thread([this] {
      std::unique_lock<std::mutex> lk(run_mu_);
      while (!stop_) {
        lk.unlock();
        const auto t0 = SteadyClock::now();
        const Status st = RefreshOnce();
        if (!st.ok()) failed_publishes_.fetch_add(1);
        rounds_.fetch_add(1);
        lk.lock();
        cv_.wait_until(lk, t0 + std::chrono::milliseconds(opt_.scrape.interval_ms), [this] { return stop_; });
      }
    });
  }
  void Stop() {
    std::thread t;
    { std::lock_guard<std::mutex> g(run_mu_); stop_ = true; t = std::move(th_); }
    cv_.notify_all();
    if (t.joinable()) t.join();
  }
  uint64_t rounds() const { return rounds_.load(); }
  uint64_t failed_publishes() const { return failed_publishes_.load(); }
  uint64_t failed_scrapes() const { std::lock_guard<std::mutex> g(mu_); return failed_scrapes_; }
  std::map<std::string, int32_t> adapters() const { std::lock_guard<std::mutex> g(mu_); return adapters_; }

 private:
  struct Latest { eppk_pod_row row; SteadyClock::time_point taken; };
  class Collector : public DataCollection {
   public:
    explicit Collector(SnapshotProducer* p) : p_(p) {}
    void Extract(const ScrapeTarget& ep, const ScrapeData& data) override { p_->Extract(ep, data); }
   private:
    SnapshotProducer* p_;
  };
  void Extract(const ScrapeTarget& ep, const ScrapeData& data) {
    std::lock_guard<std::mutex> g(mu_);
    if (!data.ok) { ++failed_scrapes_; return; }       // the last good row stays until it is older than max_age
    ScrapeResult r = ParseModelServerMetrics(data.body, adapters_, opt_.names);
    if (!r.unknown_adapters.empty()) {                 // a new adapter name: give it the next id, read the sets again
      bool grew = false;
      for (const std::string& a : r.unknown_adapters) {
        if (adapters_.count(a) || adapters_.size() >= EPPK_MAX_ADAPTERS) continue;
        int32_t id = 0;
        for (bool taken = true; taken; id += taken) { taken = false; for (const auto& kv : adapters_) taken = taken || kv.second == id; }
        adapters_[a] = id;
        grew = true;
      }
      if (grew) r = ParseModelServerMetrics(data.body, adapters_, opt_.names);
    }
    if (!r.complete()) { ++failed_scrapes_; return; }
    latest_[ep.id] = Latest{r.row, data.taken};
  }

  GpuPicker* picker_;
  ListFn list_;
  Options opt_;
  MetricsDataSource src_;
  mutable std::mutex mu_;                       // latest_, adapters_, failed_scrapes_
  std::map<std::string, int32_t> adapters_;
  std::unordered_map<std::string, Latest> latest_;
  uint64_t failed_scrapes_ = 0;
  std::shared_ptr<Collector> collector_;
  uint64_t epoch_ = 0;
  std::mutex run_mu_;
  std::condition_variable cv_;
  std::thread th_;
  bool stop_ = false;
  std::atomic<uint64_t> rounds_{0}, failed_publishes_{0};
};

}  // namespace eppk_host
