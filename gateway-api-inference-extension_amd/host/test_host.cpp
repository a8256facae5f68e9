// test_host.cpp — exercises the C++ host mirror (eppk_host.hpp).
//   ./test_host cpu   : fake backend, no GPU — batching, concurrency, fail-open, Unavailable, masks, round robin
//   ./test_host gpu   : real libeppk backend on device 0 — concurrent Pick() through the batcher must equal
//                       a direct eppk_pick_batch on the same rows (oracle parity of the kernel: tests/test_gpu_parity.py)
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <thread>
#include <utility>

#include "eppk_host.hpp"

using namespace eppk_host;

#define CHECK(cond)                                                                                              \
  do {                                                                                                           \
    if (!(cond)) { std::fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #cond); std::exit(1); } \
  } while (0)

// Fake backend: pick = lowest candidate index whose queue gauge is minimal (enough to see masks and snapshots work).
class FakeBackend : public Backend {
 public:
  int Publish(const eppk_pod_row* rows, uint32_t n, uint64_t) override { rows_.assign(rows, rows + n); return EPPK_OK; }
  int PickBatch(const void*, uint32_t n, const uint64_t* mask, int32_t* picks, double* scores) override {
    calls++;
    if (fail) return EPPK_ERR_DEVICE;
    const uint32_t P = (uint32_t)rows_.size(), W = (P + 63u) / 64u;
    for (uint32_t r = 0; r < n; ++r) {
      int32_t best = -1;
      for (uint32_t p = 0; p < P; ++p) {
        if (rows_[p].flags & EPPK_POD_INACTIVE) continue;
        if (mask && !((mask[(size_t)r * W + (p >> 6)] >> (p & 63u)) & 1u)) continue;
        if (best < 0 || rows_[p].queue < rows_[(size_t)best].queue) best = (int32_t)p;
      }
      picks[r] = best;
      scores[r] = 0.0;
    }
    return EPPK_OK;
  }
  // k candidates in ascending (queue, index) order
  int PickTopK(const void*, uint32_t n, const uint64_t* mask, uint32_t k, int32_t* picks, double* scores) override {
    calls++;
    if (fail) return EPPK_ERR_DEVICE;
    const uint32_t P = (uint32_t)rows_.size(), W = (P + 63u) / 64u;
    for (uint32_t r = 0; r < n; ++r) {
      std::vector<uint32_t> c;
      for (uint32_t p = 0; p < P; ++p)
        if (!(rows_[p].flags & EPPK_POD_INACTIVE) && (!mask || ((mask[(size_t)r * W + (p >> 6)] >> (p & 63u)) & 1u))) c.push_back(p);
      std::stable_sort(c.begin(), c.end(), [&](uint32_t a, uint32_t b) { return rows_[a].queue < rows_[b].queue; });
      for (uint32_t i = 0; i < k; ++i) { picks[(size_t)r * k + i] = i < c.size() ? (int32_t)c[i] : -1; scores[(size_t)r * k + i] = 0.0; }
    }
    return EPPK_OK;
  }
  int IndexInsert(const uint64_t* hashes, const uint32_t* pods, uint32_t n) override {
    std::lock_guard<std::mutex> g(learn_mu);
    if (index_full) return EPPK_ERR_INDEX_FULL;
    for (uint32_t i = 0; i < n; ++i) learned.emplace_back(hashes[i], pods[i]);
    return EPPK_OK;
  }
  int IndexRemovePod(uint32_t pod) override {
    std::lock_guard<std::mutex> g(learn_mu);
    removed.push_back(pod);
    return EPPK_OK;
  }
  int IndexAdvanceEpoch(uint32_t* e) override { *e = ++epoch; return EPPK_OK; }
  int IndexEvictOlder(uint32_t min_epoch, uint32_t* n) override {
    std::lock_guard<std::mutex> g(learn_mu);
    evict_calls.push_back(min_epoch);
    *n = 3;
    return EPPK_OK;
  }
  std::string LastError() const override { return "fake failure"; }
  std::atomic<uint32_t> epoch{1};
  std::vector<uint32_t> evict_calls;
  std::vector<uint32_t> removed;
  std::mutex learn_mu;
  std::vector<std::pair<uint64_t, uint32_t>> learned;
  std::atomic<bool> index_full{false};
  std::vector<eppk_pod_row> rows_;
  std::atomic<int> calls{0};
  std::atomic<bool> fail{false};
};

// ... with the pipelined form (two staging sets) and the device-side eviction: records the ORDER of the backend calls
class PipelinedFake : public FakeBackend {
 public:
  void* StageRows(uint32_t set) override { return buf[set & 1u].data(); }
  int StageBegin(uint32_t set, uint32_t n, bool) override { log("B" + std::to_string(set)); n_[set & 1u] = n; return EPPK_OK; }
  int StageEnd(uint32_t set, int32_t* picks, double* scores) override {
    log("E" + std::to_string(set));
    return PickBatch(nullptr, n_[set & 1u], nullptr, picks, scores);
  }
  int IndexAdvanceEpoch(uint32_t* e) override { log("T"); return FakeBackend::IndexAdvanceEpoch(e); }
  int IndexEvictOlder(uint32_t m, uint32_t* n) override { log("S"); return FakeBackend::IndexEvictOlder(m, n); }
  int IndexEvictOlderAsync(uint32_t m) override { log("A" + std::to_string(m)); return EPPK_OK; }
  void log(const std::string& op) { std::lock_guard<std::mutex> g(log_mu); ops.push_back(op); }
  std::mutex log_mu;
  std::vector<std::string> ops;
  std::vector<unsigned char> buf[2] = {std::vector<unsigned char>(1 << 16), std::vector<unsigned char>(1 << 16)};
  uint32_t n_[2] = {0, 0};
};

static std::vector<Endpoint> make_endpoints(int n) {
  std::vector<Endpoint> v;
  for (int i = 0; i < n; ++i) {
    Endpoint e;
    e.address = "10.0.0." + std::to_string(i + 1);
    e.port = "8080";
    v.push_back(e);
  }
  return v;
}

static int run_cpu() {
  {  // RoundRobinPicker: request_test.go:50-88 and server.go:91-96
    auto eps = make_endpoints(2);
    std::vector<const Endpoint*> c{&eps[0], &eps[1]};
    RoundRobinPicker rr;
    PickResult r1, r2, r3;
    CHECK(rr.Pick({}, c, &r1).ok() && rr.Pick({}, c, &r2).ok() && rr.Pick({}, c, &r3).ok());
    CHECK(r1.endpoint != r2.endpoint && r1.endpoint == r3.endpoint && r1.endpoint == "10.0.0.2:8080");
    CHECK(rr.Pick({}, {}, &r1).code == Code::Unavailable);
    CHECK(JoinHostPort("::1", "80") == "[::1]:80");
  }
  auto fake = new FakeBackend();
  GpuPickerOptions opt;
  opt.max_pods = 64;
  opt.max_blocks = 4;
  opt.max_batch = 32;
  opt.window = std::chrono::microseconds(2000);
  GpuPicker gp(std::unique_ptr<Backend>(fake), opt);
  auto eps = make_endpoints(5);
  std::vector<eppk_pod_row> rows(5);
  std::memset(rows.data(), 0, rows.size() * sizeof(eppk_pod_row));
  const uint32_t q[5] = {9, 3, 7, 1, 5};
  for (int i = 0; i < 5; ++i) rows[(size_t)i].queue = q[i];
  std::vector<const Endpoint*> all;
  for (auto& e : eps) all.push_back(&e);

  // before any snapshot: fail open to round robin (never an error to the stream)
  PickResult pr;
  CHECK(gp.Pick({}, all, &pr).ok() && gp.fail_opens() == 1);
  CHECK(gp.PublishSnapshot(eps, rows, {{"adapter-a", 3}}, 1).ok());

  // all candidates -> min queue is pod 3; subset {0,2} -> pod 2; empty candidate slice -> Unavailable
  CHECK(gp.Pick({}, all, &pr).ok() && pr.endpoint == "10.0.0.4:8080");
  std::vector<const Endpoint*> sub{&eps[0], &eps[2]};
  CHECK(gp.Pick({}, sub, &pr).ok() && pr.endpoint == "10.0.0.3:8080");
  CHECK(gp.Pick({}, {}, &pr).code == Code::Unavailable);
  Endpoint stranger;
  stranger.address = "192.168.0.1";
  stranger.port = "1";
  std::vector<const Endpoint*> unknown{&stranger};  // candidates unknown to the snapshot: no scoreable endpoint -> the picker fails OPEN
  {                                                 // (round robin over the request's candidates; only an empty list fails closed)
    const uint64_t fo = gp.fail_opens();
    CHECK(gp.Pick({}, unknown, &pr).ok() && pr.endpoint == "192.168.0.1:1" && gp.fail_opens() == fo + 1);
  }

  // concurrency: 16 threads x 200 picks are batched (fewer backend calls than picks), all correct
  const int before = fake->calls.load();
  std::vector<std::thread> th;
  std::atomic<int> bad{0};
  for (int t = 0; t < 16; ++t)
    th.emplace_back([&, t] {
      for (int i = 0; i < 200; ++i) {
        PickResult r;
        const bool use_sub = ((t + i) & 1) != 0;
        Status s = gp.Pick({}, use_sub ? sub : all, &r);
        if (!s.ok() || r.endpoint != (use_sub ? "10.0.0.3:8080" : "10.0.0.4:8080")) bad++;
      }
    });
  for (auto& x : th) x.join();
  CHECK(bad.load() == 0);
  const int calls = fake->calls.load() - before;
  CHECK(calls < 3200 && gp.largest_batch() > 1 && gp.largest_batch() <= 32);

  // backend failure: every request still gets an endpoint from the round-robin fallback
  fake->fail = true;
  const uint64_t fo = gp.fail_opens();
  CHECK(gp.Pick({}, all, &pr).ok() && !pr.endpoint.empty() && gp.fail_opens() == fo + 1);
  fake->fail = false;
  CHECK(gp.Pick({}, all, &pr).ok() && pr.endpoint == "10.0.0.4:8080");
  {  // snapshots published WHILE requests are in flight: a request prepares its candidate mask on its own thread against the snapshot
     // it saw in Pick(); when another one is current by the time its batch is built, the dispatcher rebuilds the mask -- every pick
     // must still be a candidate of its request and an endpoint of one of the two snapshots, whatever the interleaving.
    auto eps_b = make_endpoints(5);
    std::vector<Endpoint> snap_b{eps_b[4], eps_b[2], eps_b[0]};          // other order, other size: every index means another endpoint
    std::vector<eppk_pod_row> rows_b(3);
    std::memset(rows_b.data(), 0, rows_b.size() * sizeof(eppk_pod_row));
    rows_b[0].queue = 2; rows_b[1].queue = 1; rows_b[2].queue = 3;        // shortest queue: 10.0.0.3
    std::atomic<bool> stop_pub{false};
    std::atomic<int> wrong{0}, served{0};
    std::thread publisher([&] {
      for (uint64_t e = 10; !stop_pub.load(); ++e) {
        if (!(e & 1 ? gp.PublishSnapshot(snap_b, rows_b, {}, e) : gp.PublishSnapshot(eps, rows, {{"adapter-a", 3}}, e)).ok()) wrong++;
        std::this_thread::sleep_for(std::chrono::microseconds(300));
      }
    });
    std::vector<std::thread> callers;
    for (int t = 0; t < 8; ++t)
      callers.emplace_back([&, t] {
        std::vector<const Endpoint*> mine;                                   // thread t may use every endpoint but number t % 5
        for (int i = 0; i < 5; ++i) if (i != t % 5) mine.push_back(&eps[(size_t)i]);
        for (int i = 0; i < 300; ++i) {
          PickRequest rq;
          rq.body = std::string(200 + (size_t)i, (char)('a' + t));
          PickResult r;
          const Status st = gp.Pick(rq, mine, &r);
          if (!st.ok()) { wrong++; continue; }
          bool is_cand = false;
          for (const Endpoint* e : mine) is_cand = is_cand || JoinHostPort(e->address, e->port) == r.endpoint;
          if (!is_cand) wrong++;
          // the shortest queue among this thread's candidates, under either snapshot
          const std::string a = t % 5 == 3 ? "10.0.0.2:8080" : "10.0.0.4:8080", b = t % 5 == 2 ? "10.0.0.5:8080" : "10.0.0.3:8080";
          if (r.endpoint != a && r.endpoint != b) wrong++;
          served++;
        }
      });
    for (auto& c : callers) c.join();
    stop_pub = true;
    publisher.join();
    CHECK(wrong.load() == 0 && served.load() == 2400);
    CHECK(gp.PublishSnapshot(eps, rows, {{"adapter-a", 3}}, 999).ok());
  }
  std::printf("host cpu ok: %d backend calls for 3200 concurrent picks, largest batch %llu\n", calls,
              (unsigned long long)gp.largest_batch());
  {  // ordered fallbacks (PickResult.Fallbacks, server.go:74) through the micro-batcher
    auto eps = make_endpoints(5);
    std::vector<eppk_pod_row> rows(5);
    std::memset(rows.data(), 0, rows.size() * sizeof(eppk_pod_row));
    const uint32_t q[5] = {7, 1, 9, 3, 5};
    for (int i = 0; i < 5; ++i) rows[(size_t)i].queue = q[i];
    GpuPickerOptions opt;
    opt.max_pods = 64; opt.max_blocks = 4; opt.max_batch = 8; opt.fallbacks = 2;
    GpuPicker gp(std::unique_ptr<Backend>(new FakeBackend()), opt);
    CHECK(gp.PublishSnapshot(eps, rows, {}, 1).ok());
    std::vector<const Endpoint*> c{&eps[0], &eps[2], &eps[3], &eps[4]};   // pod 1 (the global best) is not a candidate
    PickResult r;
    PickRequest rq;
    CHECK(gp.Pick(rq, c, &r).ok());
    CHECK(r.endpoint == "10.0.0.4:8080");
    CHECK(r.fallbacks.size() == 2 && r.fallbacks[0] == "10.0.0.5:8080" && r.fallbacks[1] == "10.0.0.1:8080");
    std::vector<const Endpoint*> one{&eps[2]};                            // fewer candidates than requested fallbacks
    CHECK(gp.Pick(rq, one, &r).ok() && r.endpoint == "10.0.0.3:8080" && r.fallbacks.empty());
  }
  {  // learn_prefixes: after a batch the picked pod is recorded for every block hash of the request's prompt
    auto eps = make_endpoints(3);
    std::vector<eppk_pod_row> rows(3);
    std::memset(rows.data(), 0, rows.size() * sizeof(eppk_pod_row));
    const uint32_t q[3] = {4, 2, 6};
    for (int i = 0; i < 3; ++i) rows[(size_t)i].queue = q[i];
    GpuPickerOptions opt;
    opt.max_pods = 64; opt.max_blocks = 4; opt.max_batch = 8; opt.block_chars = 8; opt.learn_prefixes = true;
    auto fk = new FakeBackend();
    GpuPicker gp(std::unique_ptr<Backend>(fk), opt);
    CHECK(gp.PublishSnapshot(eps, rows, {}, 1).ok());
    std::vector<const Endpoint*> c{&eps[0], &eps[1], &eps[2]};
    PickRequest rq;
    rq.model = "base";
    rq.body = "0123456789abcdefXYZ";          // two full 8-character blocks (+ a 3-character tail that is not a block)
    PickResult r;
    CHECK(gp.Pick(rq, c, &r).ok() && r.endpoint == "10.0.0.2:8080");
    uint64_t want[4];
    const int nb = eppk_hash_prompt((const uint8_t*)rq.model.data(), rq.model.size(), (const uint8_t*)rq.body.data(), rq.body.size(), 8, want, 4);
    CHECK(nb == 2);
    {
      std::lock_guard<std::mutex> g(fk->learn_mu);
      CHECK(fk->learned.size() == 2 && fk->learned[0] == std::make_pair(want[0], 1u) && fk->learned[1] == std::make_pair(want[1], 1u));
    }
    PickRequest empty;                        // no blocks: nothing to learn, no backend call needed
    CHECK(gp.Pick(empty, c, &r).ok());
    { std::lock_guard<std::mutex> g(fk->learn_mu); CHECK(fk->learned.size() == 2); }
    fk->index_full = true;                    // a full table never fails the pick; the drop is counted
    CHECK(gp.Pick(rq, c, &r).ok() && r.endpoint == "10.0.0.2:8080" && gp.learn_drops() == 1 && gp.fail_opens() == 0);
  }
  {  // stable_slots: an endpoint keeps its candidate index across snapshots; a freed slot is wiped from the index, is never
     // picked while it is a hole, and goes to the next newcomer
    GpuPickerOptions opt;
    opt.max_pods = 8; opt.max_blocks = 4; opt.max_batch = 8; opt.stable_slots = true;
    auto fk = new FakeBackend();
    GpuPicker gp(std::unique_ptr<Backend>(fk), opt);
    auto eps = make_endpoints(4);                       // A=.1 B=.2 C=.3 D=.4
    auto row = [](uint32_t q) { eppk_pod_row r; std::memset(&r, 0, sizeof r); r.queue = q; return r; };
    CHECK(gp.PublishSnapshot({eps[0], eps[1], eps[2]}, {row(5), row(1), row(7)}, {}, 1).ok());
    CHECK(gp.SlotOf("10.0.0.1:8080") == 0 && gp.SlotOf("10.0.0.2:8080") == 1 && gp.SlotOf("10.0.0.3:8080") == 2);
    std::vector<const Endpoint*> abc{&eps[0], &eps[1], &eps[2]};
    PickResult r;
    CHECK(gp.Pick({}, abc, &r).ok() && r.endpoint == "10.0.0.2:8080");            // B has the shortest queue
    // B leaves: C keeps slot 2, slot 1 becomes a hole (published with EPPK_POD_INACTIVE: the library forgets it in the index)
    CHECK(gp.PublishSnapshot({eps[0], eps[2]}, {row(5), row(7)}, {}, 2).ok());
    CHECK(gp.SlotOf("10.0.0.3:8080") == 2 && gp.SlotOf("10.0.0.2:8080") == -1);
    CHECK(fk->rows_.size() == 3 && (fk->rows_[1].flags & EPPK_POD_INACTIVE) && !(fk->rows_[0].flags & EPPK_POD_INACTIVE) && !(fk->rows_[2].flags & EPPK_POD_INACTIVE));
    std::vector<const Endpoint*> ac{&eps[0], &eps[2]};
    CHECK(gp.Pick({}, ac, &r).ok() && r.endpoint == "10.0.0.1:8080");             // never the hole
    CHECK(gp.Pick({}, abc, &r).ok() && r.endpoint == "10.0.0.1:8080");            // a stale candidate (B) is simply not scoreable
    // D arrives and takes the hole; A leaves at the same time; C is still slot 2
    CHECK(gp.PublishSnapshot({eps[2], eps[3]}, {row(7), row(3)}, {}, 3).ok());
    CHECK(gp.SlotOf("10.0.0.3:8080") == 2 && gp.SlotOf("10.0.0.1:8080") == -1);
    const int32_t d = gp.SlotOf("10.0.0.4:8080");
    CHECK(d == 0 || d == 1);                                                       // one of the two freed slots (the lowest: 0)
    CHECK(d == 0);
    CHECK(fk->rows_.size() == 3 && (fk->rows_[1].flags & EPPK_POD_INACTIVE) && !(fk->rows_[0].flags & EPPK_POD_INACTIVE));   // A's slot went to D, B's is still a hole
    std::vector<const Endpoint*> cd{&eps[2], &eps[3]};
    CHECK(gp.Pick({}, cd, &r).ok() && r.endpoint == "10.0.0.4:8080");
    // everything leaves, then one endpoint comes back: the table shrinks to nothing and starts again at slot 0
    CHECK(gp.PublishSnapshot({}, {}, {}, 4).ok() && fk->rows_.empty());
    CHECK(gp.PublishSnapshot({eps[1]}, {row(2)}, {}, 5).ok() && gp.SlotOf("10.0.0.2:8080") == 0 && fk->rows_.size() == 1);
    // trailing holes shrink the published table
    CHECK(gp.PublishSnapshot({eps[1], eps[0], eps[2]}, {row(2), row(4), row(6)}, {}, 6).ok() && fk->rows_.size() == 3);
    CHECK(gp.PublishSnapshot({eps[1]}, {row(2)}, {}, 7).ok() && fk->rows_.size() == 1);
    // more endpoints than max_pods is an error, not a crash
    auto many = make_endpoints(9);
    std::vector<eppk_pod_row> mrows(9, row(1));
    CHECK(!gp.PublishSnapshot(many, mrows, {}, 8).ok());
  }
  {  // index ageing: the dispatcher ticks the epoch between batches and evicts what is older than `index_keep_epochs`
    GpuPickerOptions opt;
    opt.max_pods = 8; opt.max_blocks = 4; opt.max_batch = 4;
    opt.index_epoch_interval = std::chrono::microseconds(1);   // every batch is later than the previous tick
    opt.index_keep_epochs = 2;
    auto fk = new FakeBackend();
    GpuPicker gp(std::unique_ptr<Backend>(fk), opt);
    auto eps = make_endpoints(2);
    std::vector<eppk_pod_row> rows(2);
    std::memset(rows.data(), 0, rows.size() * sizeof(eppk_pod_row));
    CHECK(gp.PublishSnapshot(eps, rows, {}, 1).ok());
    std::vector<const Endpoint*> c{&eps[0], &eps[1]};
    PickResult r;
    for (int i = 0; i < 4; ++i) {
      CHECK(gp.Pick({}, c, &r).ok());
      std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
    std::lock_guard<std::mutex> g(fk->learn_mu);
    // epochs 2, 3, 4, 5 after the four batches; eviction starts once the epoch exceeds keep (3 -> evict < 1, 4 -> < 2, 5 -> < 3)
    CHECK(fk->epoch.load() == 5u);
    CHECK(fk->evict_calls.size() == 3 && fk->evict_calls[0] == 1u && fk->evict_calls[1] == 2u && fk->evict_calls[2] == 3u);
    CHECK(gp.evicted() == 9u);
  }
  {  // ... and with a pipelined backend that offers the device-side eviction: the tick lands while a batch is between Begin and End
     // (no collect in front of it), the eviction is the asynchronous one, and the synchronous form is never called
    GpuPickerOptions opt;
    opt.max_pods = 8; opt.max_blocks = 4; opt.max_batch = 4;
    opt.index_epoch_interval = std::chrono::microseconds(1);
    opt.index_keep_epochs = 1;
    auto fk = new PipelinedFake();
    GpuPicker gp(std::unique_ptr<Backend>(fk), opt);
    auto eps = make_endpoints(2);
    std::vector<eppk_pod_row> rows(2);
    std::memset(rows.data(), 0, rows.size() * sizeof(eppk_pod_row));
    CHECK(gp.PublishSnapshot(eps, rows, {}, 1).ok());
    std::vector<const Endpoint*> c{&eps[0], &eps[1]};
    PickResult r;
    for (int i = 0; i < 4; ++i) {
      CHECK(gp.Pick({}, c, &r).ok() && !r.endpoint.empty());
      std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
    std::lock_guard<std::mutex> g(fk->log_mu);
    int in_flight = 0, async = 0;
    for (const std::string& op : fk->ops) {
      if (op[0] == 'B') ++in_flight;
      if (op[0] == 'E') --in_flight;
      CHECK(op[0] != 'S');                                   // never the synchronous eviction
      if (op[0] == 'A') { ++async; CHECK(in_flight >= 1); }   // ... and the asynchronous one with a batch in flight
    }
    CHECK(async >= 2 && gp.evictions_async() == (uint64_t)async && gp.evicted() == 0u);
  }
  return 0;
}

static int run_gpu() {
  SchedulerProfile prof;
  prof.scorers = {{EPPK_SCORER_QUEUE, 2}, {EPPK_SCORER_KV, 2}, {EPPK_SCORER_LORA, 1}, {EPPK_SCORER_PREFIX, 3}};
  GpuPickerOptions opt;
  opt.max_pods = 256;
  opt.max_blocks = 8;
  opt.max_batch = 512;
  opt.window = std::chrono::microseconds(500);
  eppk_cfg cfg = MakeCfg(prof, opt, 1024, 0);
  std::string err;
  auto be = LibEppkBackend::Create(cfg, &err);
  if (!be) {
    std::fprintf(stderr, "create failed: %s\n", err.c_str());
    return 1;
  }
  eppk_ctx* ctx = be->ctx();
  const int P = 200;
  auto eps = make_endpoints(P);
  for (int i = 0; i < P; ++i) eps[(size_t)i].address = "10.1.0." + std::to_string(i);
  std::vector<eppk_pod_row> rows((size_t)P);
  std::memset(rows.data(), 0, rows.size() * sizeof(eppk_pod_row));
  uint64_t x = 88172645463325252ull;
  auto rnd = [&] { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
  for (int i = 0; i < P; ++i) {
    rows[(size_t)i].queue = (uint32_t)(rnd() % 16);
    rows[(size_t)i].kv_util = (double)(rnd() % 1025) / 1024.0;
    rows[(size_t)i].max_lora = 4;
    rows[(size_t)i].active[0] = 1ull << (rnd() % 8);
  }
  // prompts: 6 shared system prompts; cache prompt g (for its adapter) on pods {g, g+10, g+20}
  std::vector<std::string> sys;
  for (int g = 0; g < 6; ++g) sys.push_back(std::string(256, (char)('a' + g)));
  std::unordered_map<std::string, int32_t> adapters;
  for (int a = 0; a < 8; ++a) adapters["adapter-" + std::to_string(a)] = a;
  GpuPicker gp(std::move(be), opt);
  CHECK(gp.PublishSnapshot(eps, rows, adapters, 1).ok());
  for (int g = 0; g < 6; ++g) {
    const std::string model = "adapter-" + std::to_string(g);
    uint64_t h[8];
    int n = eppk_hash_prompt((const uint8_t*)model.data(), model.size(), (const uint8_t*)sys[(size_t)g].data(), sys[(size_t)g].size(), 64, h, 8);
    CHECK(n == 4);
    for (int pod : {g, g + 10, g + 20})
      for (int i = 0; i < n; ++i) {
        uint32_t pp = (uint32_t)pod;
        CHECK(eppk_index_insert(ctx, &h[i], &pp, 1) == EPPK_OK);
      }
  }
  std::vector<const Endpoint*> all;
  for (auto& e : eps) all.push_back(&e);

  // direct batch = ground truth for the batcher plumbing
  const int N = 384;
  std::vector<PickRequest> reqs((size_t)N);
  std::vector<uint8_t> rowsbuf((size_t)N * 72, 0);
  for (int i = 0; i < N; ++i) {
    const int g = i % 6;
    PickRequest& rq = reqs[(size_t)i];
    rq.model = (i % 5 == 0) ? "base" : "adapter-" + std::to_string(g);
    rq.body = sys[(size_t)g] + std::string(128, (char)('0' + i % 10)) + std::to_string(i) + std::string(100, 'z');
    eppk_req_hdr hdr;
    auto it = adapters.find(rq.model);
    hdr.adapter = it == adapters.end() ? -1 : it->second;
    int nb = eppk_hash_prompt((const uint8_t*)rq.model.data(), rq.model.size(), (const uint8_t*)rq.body.data(), rq.body.size(), 64,
                              (uint64_t*)(rowsbuf.data() + (size_t)i * 72 + 8), 8);
    hdr.n_blocks = (uint32_t)nb;
    std::memcpy(rowsbuf.data() + (size_t)i * 72, &hdr, 8);
  }
  std::vector<int32_t> want((size_t)N);
  std::vector<double> sc((size_t)N);
  CHECK(eppk_pick_batch(ctx, rowsbuf.data(), (uint32_t)N, nullptr, want.data(), sc.data()) == EPPK_OK);
  int prefix_wins = 0;
  for (int i = 0; i < N; ++i) {
    const int g = i % 6, w = want[(size_t)i];
    if (i % 5 != 0 && (w == g || w == g + 10 || w == g + 20)) prefix_wins++;
  }

  std::vector<std::string> got((size_t)N);
  std::vector<std::thread> th;
  for (int t = 0; t < 8; ++t)
    th.emplace_back([&, t] {
      for (int i = t; i < N; i += 8) {
        PickResult r;
        Status s = gp.Pick(reqs[(size_t)i], all, &r);
        got[(size_t)i] = s.ok() ? r.endpoint : "ERR";
      }
    });
  for (auto& t : th) t.join();
  for (int i = 0; i < N; ++i) CHECK(got[(size_t)i] == JoinHostPort(eps[(size_t)want[(size_t)i]].address, "8080"));
  CHECK(gp.fail_opens() == 0 && gp.batches() < (uint64_t)N);
  CHECK(prefix_wins > 0);
  std::printf("host gpu ok: %d concurrent picks in %llu batches equal the direct batch; %d picks landed on a prefix-cached pod\n", N,
              (unsigned long long)gp.batches(), prefix_wins);
  return 0;
}

int main(int argc, char** argv) {
  const std::string mode = argc > 1 ? argv[1] : "cpu";
  return mode == "gpu" ? run_gpu() : run_cpu();
}
