// eppk_host.hpp — C++ host side above the C ABI, mirroring the reference's picker / scheduler interfaces
// for this path (the reference is Go; no Go toolchain in this image, so its compiled-code host side is
// written in C++ — the Go twin is in INTEGRATION.md).
//
// Mirrored reference shapes (same names, argument meaning, error behaviour):
//   Endpoint                pkg/lwepp/datastore/datastore.go:40-46
//   PickRequest/PickResult  pkg/lwepp/handlers/server.go:65-77
//   EndpointPicker.Pick     pkg/lwepp/handlers/server.go:79-82    (goroutine-safe, one call per request)
//   RoundRobinPicker        pkg/lwepp/handlers/server.go:84-101
//   Scorer/WeightedScorer/SchedulerProfile/Picker
//                           docs/proposals/0845-scheduler-architecture-proposal/interfaces/interface.go:70-142
//   error codes             codes.Unavailable (server.go:91-93), codes.Internal (server.go:143-146)
//
// GpuPicker turns concurrent per-request Pick() calls into batches for eppk_pick_batch (one fused HIP
// kernel per batch) and FAILS OPEN to RoundRobinPicker on any backend error (a GPU
// picker must never take the stream down).  Nothing here scores on the CPU.
#pragma once

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/eppk.h"

namespace eppk_host {

// ---- reference data shapes -------------------------------------------------------------------------

struct Endpoint {  // datastore.go:40-46
  std::string namespaced_name, pod_name, address, port;
  std::map<std::string, std::string> labels;
};

struct PickRequest {  // server.go:65-69
  std::map<std::string, std::vector<std::string>> headers;
  std::string body;
  std::string model;  // TargetModel: base model or LoRA adapter name
};

struct PickResult {  // server.go:72-77
  std::string endpoint;  // "ip:port"
  std::vector<std::string> fallbacks;
};

enum class Code { OK = 0, Unavailable = 14, Internal = 13 };  // grpc codes used by the reference
struct Status {
  Code code = Code::OK;
  std::string message;
  bool ok() const { return code == Code::OK; }
};

inline std::string JoinHostPort(const std::string& host, const std::string& port) {  // net.JoinHostPort, server.go:99
  if (host.find(':') != std::string::npos || host.find('%') != std::string::npos) return "[" + host + "]:" + port;
  return host + ":" + port;
}

class EndpointPicker {  // server.go:79-82
 public:
  virtual ~EndpointPicker() = default;
  virtual Status Pick(const PickRequest& req, const std::vector<const Endpoint*>& endpoints, PickResult* out) = 0;
};

class RoundRobinPicker : public EndpointPicker {  // server.go:84-101
 public:
  Status Pick(const PickRequest&, const std::vector<const Endpoint*>& endpoints, PickResult* out) override {
    const int32_t idx = eppk_round_robin(&rr_index_, (uint32_t)endpoints.size());
    if (idx < 0) return {Code::Unavailable, "no endpoints available"};
    out->endpoint = JoinHostPort(endpoints[(size_t)idx]->address, endpoints[(size_t)idx]->port);
    out->fallbacks.clear();
    return {};
  }

 private:
  uint64_t rr_index_ = 0;
};

// ---- scheduler profile (interface.go:70-79, :132-135) ------------------------------------------------

struct WeightedScorer {  // WeightedScorer{Scorer, weight int}
  eppk_scorer_kind kind;
  int32_t weight;
};

struct SchedulerProfile {             // filters: the candidate subset; picker: best-score, lowest index on ties
  std::vector<WeightedScorer> scorers;  // order fixes the fp summation order (SEMANTICS.md §2)
};

// ---- backend seam: the C ABI, or a fake in CPU-only tests ---------------------------------------------

class Backend {
 public:
  virtual ~Backend() = default;
  virtual int Publish(const eppk_pod_row* rows, uint32_t n_pods, uint64_t epoch) = 0;
  virtual int PickBatch(const void* reqs, uint32_t n, const uint64_t* mask, int32_t* picks, double* scores) = 0;
  // Optional: a buffer of the backend's own (pinned, DMA-able) in which the caller may BUILD up to max_batch request rows, and the
  // pick over the first n rows in it without candidate masks (eppk_host_staging / eppk_pick_batch_staged: no host copy on the way
  // to the device).  nullptr = not offered.
  virtual void* StagingRows() { return nullptr; }
  virtual int PickStaged(uint32_t, int32_t*, double*) { return EPPK_ERR_ARG; }
  // Optional: the PIPELINED form of the same (eppk_pick_stage_*): two such buffers (set 0 / 1), each with its own stream -- the
  // rows of one batch are uploaded while the other batch is scored.  Begin returns at once; with `learn` the post-route index update
  // (IndexInsert of every (block hash, pick) pair) is chained behind the pick ON THE DEVICE and the next Begin of either set scores
  // against it.  End waits for the set's picks.  nullptr from StageRows = not offered.
  virtual void* StageRows(uint32_t /*set*/) { return nullptr; }
  virtual int StageBegin(uint32_t /*set*/, uint32_t /*n*/, bool /*learn*/) { return EPPK_ERR_ARG; }
  virtual int StageEnd(uint32_t /*set*/, int32_t*, double*) { return EPPK_ERR_ARG; }
  // k ordered candidates per request (pick + fallbacks), picks/scores hold n*k entries (eppk_pick_topk)
  virtual int PickTopK(const void* reqs, uint32_t n, const uint64_t* mask, uint32_t k, int32_t* picks, double* scores) = 0;
  // prefix index: pods[i] has cached the block with hash hashes[i] ("hash(chunk i): append server", 0602-…/README.md:101-108)
  virtual int IndexInsert(const uint64_t* hashes, const uint32_t* pods, uint32_t n) = 0;
  // prefix index: forget everything recorded for candidate index `pod` (its slot is about to be handed to another endpoint)
  virtual int IndexRemovePod(uint32_t pod) = 0;
  // prefix index ageing: tick the epoch that stamps later inserts / drop every hash last stamped before min_epoch
  virtual int IndexAdvanceEpoch(uint32_t* new_epoch) = 0;
  virtual int IndexEvictOlder(uint32_t min_epoch, uint32_t* n_evicted) = 0;
  // Optional: the same eviction enqueued on the device, without a count and without a wait -- allowed while a pipelined batch is
  // between StageBegin and StageEnd (eppk_index_evict_older_device: it queues behind that batch's pick and update and ahead of the
  // next Begin's).  EPPK_ERR_ARG = not offered: the dispatcher then collects the batch in flight first and evicts synchronously.
  virtual int IndexEvictOlderAsync(uint32_t /*min_epoch*/) { return EPPK_ERR_ARG; }
  virtual std::string LastError() const = 0;
};

class LibEppkBackend : public Backend {  // include/eppk.h
 public:
  static std::unique_ptr<LibEppkBackend> Create(const eppk_cfg& cfg, std::string* err) {
    eppk_ctx* c = nullptr;
    if (eppk_create(&cfg, &c) != EPPK_OK) { if (err) *err = eppk_last_error(nullptr); return nullptr; }
    return std::unique_ptr<LibEppkBackend>(new LibEppkBackend(c));
  }
  ~LibEppkBackend() override { eppk_destroy(ctx_); }
  int Publish(const eppk_pod_row* rows, uint32_t n, uint64_t epoch) override { return eppk_snapshot_publish(ctx_, rows, n, epoch); }
  int PickBatch(const void* reqs, uint32_t n, const uint64_t* mask, int32_t* picks, double* scores) override {
    return eppk_pick_batch(ctx_, reqs, n, mask, picks, scores);
  }
  void* StagingRows() override {
    if (!staging_ && eppk_host_staging(ctx_, &staging_, nullptr) != EPPK_OK) staging_ = nullptr;
    return staging_;
  }
  int PickStaged(uint32_t n, int32_t* picks, double* scores) override { return eppk_pick_batch_staged(ctx_, n, 0, picks, scores); }
  void* StageRows(uint32_t set) override {
    if (set >= EPPK_STAGE_SETS) return nullptr;
    if (!stage_[set] && eppk_pick_stage_buffers(ctx_, set, &stage_[set], nullptr) != EPPK_OK) stage_[set] = nullptr;
    return stage_[set];
  }
  int StageBegin(uint32_t set, uint32_t n, bool learn) override { return eppk_pick_stage_begin(ctx_, set, n, 0, learn ? EPPK_PICK_LEARN : 0u); }
  int StageEnd(uint32_t set, int32_t* picks, double* scores) override { return eppk_pick_stage_end(ctx_, set, picks, scores); }
  int PickTopK(const void* reqs, uint32_t n, const uint64_t* mask, uint32_t k, int32_t* picks, double* scores) override {
    return eppk_pick_topk(ctx_, reqs, n, mask, k, picks, scores);
  }
  int IndexInsert(const uint64_t* hashes, const uint32_t* pods, uint32_t n) override { return eppk_index_insert(ctx_, hashes, pods, n); }
  int IndexRemovePod(uint32_t pod) override { return eppk_index_remove_pod(ctx_, pod); }
  int IndexAdvanceEpoch(uint32_t* e) override { return eppk_index_advance_epoch(ctx_, e); }
  int IndexEvictOlder(uint32_t min_epoch, uint32_t* n) override { return eppk_index_evict_older(ctx_, min_epoch, n); }
  int IndexEvictOlderAsync(uint32_t min_epoch) override { return eppk_index_evict_older_device(ctx_, min_epoch, nullptr); }
  std::string LastError() const override { return eppk_last_error(ctx_); }
  eppk_ctx* ctx() { return ctx_; }

 private:
  explicit LibEppkBackend(eppk_ctx* c) : ctx_(c) {}
  eppk_ctx* ctx_;
  void* staging_ = nullptr;       // the context's pinned request-row buffer (eppk_host_staging), fetched on first use
  void* stage_[EPPK_STAGE_SETS] = {nullptr, nullptr};   // the two pipelined sets (eppk_pick_stage_buffers)
};

// The same seam over a DEVICE GROUP (include/eppk.h eppk_group_*): one picker over several GPUs, the batch sharded by request, the
// replicated index learning on every member behind the gathered picks.  Everything the dispatcher uses on one context exists here:
// the two pipelined staging sets with LEARN, ordered fallbacks, ageing without a drain.
class LibEppkGroupBackend : public Backend {
 public:
  static std::unique_ptr<LibEppkGroupBackend> Create(const eppk_cfg& cfg, const std::vector<int32_t>& devices, uint32_t gather_mode, std::string* err,
                                                     uint32_t min_shard = 0) {
    eppk_group* g = nullptr;
    if (eppk_group_create(&cfg, devices.data(), (uint32_t)devices.size(), gather_mode, &g) != EPPK_OK) { if (err) *err = eppk_group_last_error(nullptr); return nullptr; }
    if (min_shard) (void)eppk_group_set_min_shard(g, min_shard);
    return std::unique_ptr<LibEppkGroupBackend>(new LibEppkGroupBackend(g));
  }
  ~LibEppkGroupBackend() override { eppk_group_destroy(g_); }
  int Publish(const eppk_pod_row* rows, uint32_t n, uint64_t epoch) override { return eppk_group_snapshot_publish(g_, rows, n, epoch); }
  int PickBatch(const void* reqs, uint32_t n, const uint64_t* mask, int32_t* picks, double* scores) override {
    return eppk_group_pick_batch(g_, reqs, n, mask, picks, scores, 0u);
  }
  void* StageRows(uint32_t set) override {
    if (set >= EPPK_STAGE_SETS) return nullptr;
    if (!stage_[set] && eppk_group_pick_stage_buffers(g_, set, &stage_[set], nullptr) != EPPK_OK) stage_[set] = nullptr;
    return stage_[set];
  }
  int StageBegin(uint32_t set, uint32_t n, bool learn) override { return eppk_group_pick_stage_begin(g_, set, n, 0, learn ? EPPK_PICK_LEARN : 0u); }
  int StageEnd(uint32_t set, int32_t* picks, double* scores) override { return eppk_group_pick_stage_end(g_, set, picks, scores); }
  int PickTopK(const void* reqs, uint32_t n, const uint64_t* mask, uint32_t k, int32_t* picks, double* scores) override {
    return eppk_group_pick_topk(g_, reqs, n, mask, k, picks, scores);
  }
  int IndexInsert(const uint64_t* hashes, const uint32_t* pods, uint32_t n) override { return eppk_group_index_insert(g_, hashes, pods, n); }
  int IndexRemovePod(uint32_t pod) override { return eppk_group_index_remove_pod(g_, pod); }
  int IndexAdvanceEpoch(uint32_t* e) override { return eppk_group_index_advance_epoch(g_, e); }
  int IndexEvictOlder(uint32_t min_epoch, uint32_t* n) override { return eppk_group_index_evict_older(g_, min_epoch, n); }
  int IndexEvictOlderAsync(uint32_t min_epoch) override { return eppk_group_index_evict_older_device(g_, min_epoch); }
  std::string LastError() const override { return eppk_group_last_error(g_); }
  eppk_group* group() { return g_; }

 private:
  explicit LibEppkGroupBackend(eppk_group* g) : g_(g) {}
  eppk_group* g_;
  void* stage_[EPPK_STAGE_SETS] = {nullptr, nullptr};
};

// ---- GpuPicker: EndpointPicker over batched picks -------------------------------------------------------

struct GpuPickerOptions {
  uint32_t max_pods = 4096, max_blocks = 32, max_batch = 4096, block_chars = 64;
  std::chrono::microseconds window{200};  // how long the dispatcher waits to fill a batch
  uint32_t fallbacks = 0;                 // PickResult.Fallbacks entries to fill (server.go:74), 0..EPPK_MAX_TOPK-1
  // After every batch record that the picked pod now holds the request's prompt blocks (SEMANTICS.md §6; "the approximate
  // prefix cache index is updated after a request is routed", 0602-…/README.md:101-108).  Needs a context created with an
  // index (index_slots > 0); a full table is not an error of the pick: the update is dropped and counted.
  bool learn_prefixes = false;
  // Stable candidate slots under churn.  The prefix index stores candidate indices, so an endpoint must
  // keep its index from one snapshot to the next: with this option an endpoint ("ip:port") keeps the slot it was first given,
  // a slot freed by a departed endpoint is handed to the next newcomer, and the slots that are empty in between are published as
  // HOLES (eppk_pod_row.flags = EPPK_POD_INACTIVE): the library keeps a hole out of every candidate set, out of the QUEUE
  // normalisers and the top tables, and forgets it in the prefix index at publish (SEMANTICS.md §6b) -- no candidate masks, so a
  // batch with holes stays on the unmasked fast route.
  bool stable_slots = false;
  // Ageing of the learned prefixes -- "mimicking a similar cache eviction strategy of the model server (e.g., LRU)",
  // 0602-…/README.md:82.  Every `index_epoch_interval` the dispatcher ticks the index epoch (between two batches) and drops the
  // hashes that were not re-inserted during the last `index_keep_epochs` epochs.  0 = no ageing.  (A keep beyond the library's stamp
  // window is clamped to it: EPPK_INDEX_EPOCH_WINDOW, include/eppk.h.)
  std::chrono::microseconds index_epoch_interval{0};
  uint32_t index_keep_epochs = 8;
};

class GpuPicker : public EndpointPicker {
 public:
  GpuPicker(std::unique_ptr<Backend> backend, const GpuPickerOptions& opt)
      : be_(std::move(backend)), opt_(ClampOptions(opt)), stride_(8u + 8u * opt.max_blocks), th_([this] { Loop(); }) {}
  static GpuPickerOptions ClampOptions(GpuPickerOptions o) {
    if (o.index_keep_epochs > EPPK_INDEX_EPOCH_WINDOW) o.index_keep_epochs = EPPK_INDEX_EPOCH_WINDOW;
    return o;
  }
  ~GpuPicker() override {
    { std::lock_guard<std::mutex> g(mu_); stop_ = true; }
    cv_.notify_all();
    th_.join();
  }

  // Publish the frozen snapshot: endpoints[i] is candidate index i and carries rows[i]'s gauges.
  // adapters: LoRA adapter name -> id (0..127); any other model name is the base model.
  Status PublishSnapshot(const std::vector<Endpoint>& endpoints, const std::vector<eppk_pod_row>& rows,
                         const std::unordered_map<std::string, int32_t>& adapters, uint64_t epoch) {
    if (endpoints.size() != rows.size()) return {Code::Internal, "endpoints/rows size mismatch"};
    auto snap = std::make_shared<Snapshot>();
    snap->adapters = adapters;
    // serialised with the dispatcher (contexts are single-caller), and never while a pipelined batch is between Begin and End
    // (include/eppk.h: no publish then): the dispatcher sees publish_waiting_, collects what is in flight and lets us through
    std::unique_lock<std::mutex> bg(be_mu_);
    ++publish_waiting_;
    be_cv_.wait(bg, [&] { return inflight_ == 0; });
    --publish_waiting_;
    struct Wake { std::condition_variable& cv; ~Wake() { cv.notify_all(); } } wake{be_cv_};   // (the dispatcher may be waiting for us)
    if (!opt_.stable_slots) {
      snap->endpoints = endpoints;
      snap->n_active = (uint32_t)endpoints.size();
      for (size_t i = 0; i < endpoints.size(); ++i) snap->Index(JoinHostPort(endpoints[i].address, endpoints[i].port), endpoints[i], (uint32_t)i);
      if (be_->Publish(rows.data(), (uint32_t)rows.size(), epoch) != EPPK_OK) return {Code::Internal, be_->LastError()};
    } else {
      // 1. endpoints that left free their slots (lowest slot first is reused first: the table stays as dense as the churn allows)
      std::unordered_map<std::string, size_t> now;
      for (size_t i = 0; i < endpoints.size(); ++i) now[JoinHostPort(endpoints[i].address, endpoints[i].port)] = i;
      if (now.size() != endpoints.size()) return {Code::Internal, "duplicate endpoint address in snapshot"};
      {  // refuse before anything is changed: newcomers must fit into freed + free + never-used slots
        size_t staying = 0;
        for (const auto& kv : slot_of_) staying += now.count(kv.first);
        const size_t newcomers = endpoints.size() - staying, leavers = slot_of_.size() - staying;
        if (newcomers > leavers + free_slots_.size() + (size_t)(opt_.max_pods - n_slots_)) return {Code::Internal, "more endpoints than max_pods"};
      }
      // A slot that stays empty is published as a hole and the library scrubs it out of the prefix index.  A slot that is handed
      // to a newcomer in THIS publish, or trimmed off the end of the table, never appears as a hole: it is scrubbed here, or the
      // next endpoint on it would inherit the departed endpoint's learned prefixes (false prefix affinity).
      std::vector<uint32_t> vacated;
      for (auto it = slot_of_.begin(); it != slot_of_.end();) {
        if (now.count(it->first)) { ++it; continue; }
        free_slots_.push_back(it->second);
        vacated.push_back(it->second);
        it = slot_of_.erase(it);
      }
      std::sort(free_slots_.begin(), free_slots_.end(), std::greater<uint32_t>());   // back() = lowest free slot
      // 2. newcomers take free slots, then fresh ones
      for (size_t i = 0; i < endpoints.size(); ++i) {
        const std::string key = JoinHostPort(endpoints[i].address, endpoints[i].port);
        if (slot_of_.count(key)) continue;
        uint32_t slot;
        if (!free_slots_.empty()) { slot = free_slots_.back(); free_slots_.pop_back(); }
        else if (n_slots_ < opt_.max_pods) slot = n_slots_++;
        else return {Code::Internal, "more endpoints than max_pods"};
        slot_of_[key] = slot;
      }
      while (n_slots_ > 0 && std::find(free_slots_.begin(), free_slots_.end(), n_slots_ - 1) != free_slots_.end()) {   // trailing holes: shrink
        free_slots_.erase(std::find(free_slots_.begin(), free_slots_.end(), n_slots_ - 1));
        --n_slots_;
      }
      for (uint32_t v : vacated) {     // reused at once, or beyond the new end of the table: not a hole of this snapshot -> scrub now
        const bool stays_hole = v < n_slots_ && std::find(free_slots_.begin(), free_slots_.end(), v) != free_slots_.end();
        if (!stays_hole && be_->IndexRemovePod(v) != EPPK_OK) return {Code::Internal, be_->LastError()};
      }
      // 3. rows by slot; a hole is an all-zero row flagged EPPK_POD_INACTIVE
      std::vector<eppk_pod_row> by_slot(n_slots_);
      std::vector<bool> active(n_slots_, false);
      snap->endpoints.assign(n_slots_, Endpoint());
      for (size_t i = 0; i < endpoints.size(); ++i) {
        const std::string key = JoinHostPort(endpoints[i].address, endpoints[i].port);
        const uint32_t slot = slot_of_[key];
        by_slot[slot] = rows[i];
        active[slot] = true;
        snap->endpoints[slot] = endpoints[i];
        snap->Index(key, endpoints[i], slot);
      }
      snap->n_active = (uint32_t)endpoints.size();
      for (uint32_t sidx = 0; sidx < n_slots_; ++sidx)
        if (!active[sidx]) { std::memset(&by_slot[sidx], 0, sizeof(eppk_pod_row)); by_slot[sidx].flags = EPPK_POD_INACTIVE; }
      if (be_->Publish(by_slot.data(), n_slots_, epoch) != EPPK_OK) return {Code::Internal, be_->LastError()};
    }
    std::lock_guard<std::mutex> g(mu_);
    snap_ = snap;
    return {};
  }

  // candidate index of an endpoint in the current snapshot (-1: unknown) -- tests and diagnostics
  int32_t SlotOf(const std::string& addr_port) {
    std::lock_guard<std::mutex> g(mu_);
    if (!snap_) return -1;
    auto it = snap_->by_addr.find(addr_port);
    return it == snap_->by_addr.end() ? -1 : (int32_t)it->second;
  }

  // server.go:79-82 — safe to call from many threads; returns when this request's batch has been picked.
  Status Pick(const PickRequest& req, const std::vector<const Endpoint*>& endpoints, PickResult* out) override {
    if (endpoints.empty()) return {Code::Unavailable, "no endpoints available"};  // server.go:91-93
    Slot slot;
    slot.req = &req;
    slot.cands = &endpoints;
    // The per-request work that does not need the device runs HERE, on the request's own thread: hashing the prompt (2 KiB of XXH64
    // per request) and turning the candidate slice into a bitmask (one look-up per candidate: O(pods) when nothing was filtered --
    // what the reference's PodList costs per request too, request.go:99).  The dispatcher thread only copies rows and masks.
    slot.hashes.resize(opt_.max_blocks);
    const int nb = eppk_hash_prompt((const uint8_t*)req.model.data(), req.model.size(), (const uint8_t*)req.body.data(), req.body.size(), opt_.block_chars,
                                    slot.hashes.data(), opt_.max_blocks);
    slot.n_blocks = nb < 0 ? 0u : (uint32_t)nb;
    {
      std::shared_ptr<Snapshot> snap;
      { std::lock_guard<std::mutex> sg(mu_); snap = snap_; }
      if (snap) { BuildMask(*snap, endpoints, &slot.mask, &slot.found); slot.mask_snap = std::move(snap); }
    }
    std::unique_lock<std::mutex> g(mu_);
    queue_.push_back(&slot);
    cv_.notify_all();
    slot.cv.wait(g, [&] { return slot.done; });   // (its own condition variable: a finished batch wakes its requests, not every waiter)
    g.unlock();
    if (slot.fail_open) {  // backend error: never take the stream down, fall back to the reference picker
      fail_open_count_.fetch_add(1, std::memory_order_relaxed);
      return rr_.Pick(req, endpoints, out);
    }
    if (slot.pick < 0) {
      // The request HAS candidates (an empty list failed closed at the top) but the snapshot scores none of them -- every scrape
      // failed, a wrong metrics port, endpoints the snapshot has not seen yet: the GPU picker fails OPEN, like on a backend error --
      // it must never be what takes serving down.  A pick of -1 with scoreable candidates cannot happen.
      if (slot.found == 0) {
        fail_open_count_.fetch_add(1, std::memory_order_relaxed);
        return rr_.Pick(req, endpoints, out);
      }
      return {Code::Unavailable, "no endpoints available"};
    }
    out->endpoint = slot.endpoint;
    out->fallbacks = std::move(slot.fallbacks);
    return {};
  }

  uint64_t batches() const { return batches_.load(); }
  uint64_t fail_opens() const { return fail_open_count_.load(); }
  uint64_t largest_batch() const { return largest_batch_.load(); }
  uint64_t learn_drops() const { return learn_drops_.load(); }
  uint64_t evicted() const { return evicted_.load(); }                    // hashes dropped by SYNCHRONOUS evictions (the device-side form has no count)
  uint64_t evictions_async() const { return evictions_async_.load(); }   // evictions queued on the device behind a batch in flight

 private:
  struct Snapshot {
    std::vector<Endpoint> endpoints;  // by candidate index (stable_slots: by slot, holes are default-constructed)
    uint32_t n_active = 0;            // endpoints that are candidates (== endpoints.size() unless there are holes)
    std::unordered_map<std::string, uint32_t> by_addr;
    // the same map keyed by a 64-bit fingerprint of (address, port): the per-candidate look-up of BuildMask without building the
    // "ip:port" string (an allocation per candidate and request).  A hit is verified against the endpoint's strings; two endpoints
    // with one fingerprint make it ambiguous (kAmbiguous) and fall back to by_addr.
    static constexpr uint32_t kAmbiguous = 0xFFFFFFFFu;
    std::unordered_map<uint64_t, uint32_t> by_fp;
    static uint64_t Fingerprint(const std::string& address, const std::string& port) {
      uint64_t h = 0xCBF29CE484222325ull;                                        // FNV-1a over address, a separator, port
      for (unsigned char c : address) { h ^= c; h *= 0x100000001B3ull; }
      h ^= 0xFFu; h *= 0x100000001B3ull;
      for (unsigned char c : port) { h ^= c; h *= 0x100000001B3ull; }
      return h;
    }
    void Index(const std::string& key, const Endpoint& e, uint32_t slot) {
      by_addr[key] = slot;
      auto ins = by_fp.emplace(Fingerprint(e.address, e.port), slot);
      if (!ins.second && ins.first->second != slot) ins.first->second = kAmbiguous;
    }
    // candidate index of an endpoint, or -1
    int64_t Find(const Endpoint& e) const {
      auto f = by_fp.find(Fingerprint(e.address, e.port));
      if (f != by_fp.end() && f->second != kAmbiguous) {
        const Endpoint& mine = endpoints[f->second];
        if (mine.address == e.address && mine.port == e.port) return f->second;
      }
      auto a = by_addr.find(JoinHostPort(e.address, e.port));                   // ambiguous or colliding fingerprint, or unknown
      return a == by_addr.end() ? -1 : (int64_t)a->second;
    }
    std::unordered_map<std::string, int32_t> adapters;
  };
  struct Slot {
    const PickRequest* req = nullptr;
    const std::vector<const Endpoint*>* cands = nullptr;
    // prepared by the CALLER's thread in Pick() (every request thread works on its own request; the dispatcher only copies):
    std::vector<uint64_t> hashes;             // the block-hash chain of the prompt
    uint32_t n_blocks = 0;
    std::shared_ptr<Snapshot> mask_snap;      // the snapshot `mask` was built against (the dispatcher rebuilds it if a publish came between)
    std::vector<uint64_t> mask;               // candidate bitmask over that snapshot's indices
    uint32_t found = 0;                       // distinct candidates known to that snapshot
    bool done = false, fail_open = false;
    std::condition_variable cv;               // signalled under mu_ when `done` is set
    int32_t pick = -1;
    std::string endpoint;
    std::vector<std::string> fallbacks;
  };

  // candidate slice -> bitmask over the snapshot's indices (the subset filter already ran: request.go:104-133)
  static void BuildMask(const Snapshot& snap, const std::vector<const Endpoint*>& cands, std::vector<uint64_t>* mask, uint32_t* found) {
    const uint32_t P = (uint32_t)snap.endpoints.size(), W = (P + 63u) / 64u;
    mask->assign(W ? W : 1, 0);
    *found = 0;
    for (const Endpoint* e : cands) {
      const int64_t idx = snap.Find(*e);
      if (idx < 0) continue;  // candidate unknown to this snapshot: not scoreable
      const uint64_t bit = 1ull << (idx & 63);
      if (!((*mask)[(size_t)idx >> 6] & bit)) ++*found;
      (*mask)[(size_t)idx >> 6] |= bit;
    }
  }

  // One batch of the pipelined path between StageBegin and StageEnd.
  struct InFlight {
    std::vector<Slot*> batch;
    std::shared_ptr<Snapshot> snap;      // the endpoint table the device scored against
    uint32_t set = 0;
    bool active = false;
  };

  // picks -> "ip:port" (+ fallbacks) of every request of a batch; Slot fields other than `done` are read by the owner only after `done`
  static void Resolve(const std::vector<Slot*>& batch, const Snapshot& snap, const int32_t* picks, uint32_t k) {
    for (size_t i = 0; i < batch.size(); ++i) {
      batch[i]->pick = picks[i * k];
      batch[i]->fallbacks.clear();
      for (uint32_t f = 0; f < k; ++f) {
        const int32_t p = picks[i * k + f];
        if (p < 0) break;
        const Endpoint& e = snap.endpoints[(size_t)p];
        if (f == 0) batch[i]->endpoint = JoinHostPort(e.address, e.port);
        else batch[i]->fallbacks.push_back(JoinHostPort(e.address, e.port));
      }
    }
  }

  void Finish(std::vector<Slot*>& batch, bool failed) {          // wake the requests of a batch (takes mu_)
    std::lock_guard<std::mutex> g(mu_);
    for (Slot* s : batch) { s->fail_open = failed; s->done = true; s->cv.notify_one(); }   // under mu_: the Slot outlives the call
    batches_.fetch_add(1);
    if (batch.size() > largest_batch_.load()) largest_batch_.store(batch.size());
  }

  // collect a pipelined batch: wait for its picks, resolve, wake its requests
  void Collect(InFlight& f, std::vector<int32_t>& picks, std::vector<double>& scores) {
    bool failed;
    {
      std::lock_guard<std::mutex> bg(be_mu_);
      picks.resize(f.batch.size());
      scores.resize(f.batch.size());
      failed = be_->StageEnd(f.set, picks.data(), scores.data()) != EPPK_OK;
      if (!failed) Resolve(f.batch, *f.snap, picks.data(), 1u);
      --inflight_;
    }
    be_cv_.notify_all();                 // (a publisher may be waiting for the pipeline to drain)
    Finish(f.batch, failed);
    f.active = false;
    f.snap.reset();
  }

  // The dispatcher.  Two batches can be in the backend at once when it offers the pipelined staging sets (and the batch needs
  // neither candidate masks nor fallback lists): the rows of batch k + 1 are built and uploaded while batch k is scored, batch k is
  // collected afterwards -- the device is busy while requests keep arriving, instead of idling through PCIe and row construction.
  void Loop() {
    std::vector<Slot*> batch;
    std::vector<uint8_t> rows;
    std::vector<uint64_t> mask;
    std::vector<int32_t> picks;
    std::vector<double> scores;
    std::vector<uint64_t> learn_h;
    std::vector<uint32_t> learn_p;
    InFlight fly;
    uint32_t next_set = 0;
    auto next_tick = std::chrono::steady_clock::now() + opt_.index_epoch_interval;
    std::unique_lock<std::mutex> g(mu_);
    for (;;) {
      cv_.wait(g, [&] { return stop_ || !queue_.empty() || fly.active; });
      if (queue_.empty()) {
        if (fly.active) { g.unlock(); Collect(fly, picks, scores); g.lock(); continue; }   // nothing to overlap it with: collect
        if (stop_) return;
        continue;
      }
      if (!fly.active && queue_.size() < opt_.max_batch && !stop_)  // give concurrent callers a moment to join the batch (while a batch is
        cv_.wait_for(g, opt_.window, [&] { return stop_ || queue_.size() >= opt_.max_batch; });   // in flight they have been joining already)
      const size_t n = queue_.size() < opt_.max_batch ? queue_.size() : opt_.max_batch;
      batch.assign(queue_.begin(), queue_.begin() + (long)n);
      queue_.erase(queue_.begin(), queue_.begin() + (long)n);
      g.unlock();  // row building and the kernel run without the queue lock
      bool failed = false, begun = false;
      {
        // be_mu_ serialises the context (include/eppk.h: one caller per context) and pins the snapshot:
        // the endpoint table read here is the one the device scores against.
        std::unique_lock<std::mutex> bg(be_mu_);
        while (publish_waiting_ > 0) {          // a publish wants the context: let the pipeline drain first, then let it through
          if (fly.active) { bg.unlock(); Collect(fly, picks, scores); bg.lock(); }
          else be_cv_.wait(bg, [&] { return publish_waiting_ == 0; });
        }
        std::shared_ptr<Snapshot> snap;
        { std::lock_guard<std::mutex> sg(mu_); snap = snap_; }
        failed = !snap;
        const uint32_t k = 1u + (opt_.fallbacks < EPPK_MAX_TOPK ? opt_.fallbacks : EPPK_MAX_TOPK - 1u);
        uint32_t P = 0, W = 0;
        bool any_mask = false;
        if (!failed) {
          // masks first: they decide which entry point the batch takes, and with it the buffer its rows are built in.  A batch that
          // cannot be pipelined (masks or fallbacks) has to collect the batch in flight first, and collecting RELEASES be_mu_: a
          // publish that was waiting for the pipeline to drain may come through in that window.  The masks, the mask width and the
          // endpoint table this batch resolves its picks against must all belong to the snapshot the device scores against, so the
          // snapshot is read again behind the collect and the masks are rebuilt when it changed (`mask_snap` tells).
          for (;;) {
            P = (uint32_t)snap->endpoints.size(); W = (P + 63u) / 64u;
            any_mask = false;
            mask.assign(n * (size_t)(W ? W : 1), 0);
            for (size_t i = 0; i < n; ++i) {
              Slot& sl = *batch[i];
              if (sl.mask_snap != snap) { BuildMask(*snap, *sl.cands, &sl.mask, &sl.found); sl.mask_snap = snap; }   // a publish came between Pick() and this batch
              std::memcpy(mask.data() + i * (size_t)(W ? W : 1), sl.mask.data(), (size_t)(W ? W : 1) * 8u);
              if (sl.found != snap->n_active) any_mask = true;   // (all ACTIVE endpoints are candidates: no mask; holes are the library's business)
            }
            if ((k == 1 && !any_mask && be_->StageRows(next_set) != nullptr) || !fly.active) break;    // pipelined, or nothing in flight
            bg.unlock(); Collect(fly, picks, scores); bg.lock();                                      // keep the batches in order on the context
            std::shared_ptr<Snapshot> now;
            { std::lock_guard<std::mutex> sg(mu_); now = snap_; }
            if (now == snap) break;
            snap = now;                                                                                // published meanwhile: once more
            if (!snap) break;
          }
          failed = !snap;
        }
        if (!failed) {
          // rows are built in the backend's pinned buffers when it offers them: the pipelined set, else the single staging buffer
          // (a batch without candidate masks then goes to the device without another host copy)
          const bool plain = k == 1 && !any_mask;
          uint8_t* rowp = plain ? (uint8_t*)be_->StageRows(next_set) : nullptr;
          const bool pipelined = rowp != nullptr;
          if (!pipelined) rowp = (uint8_t*)be_->StagingRows();
          const bool staged = rowp != nullptr;
          if (staged) std::memset(rowp, 0, n * stride_);
          else { rows.assign(n * stride_, 0); rowp = rows.data(); }
          for (size_t i = 0; i < n; ++i) {
            Slot& sl = *batch[i];
            eppk_req_hdr hdr;
            auto it = snap->adapters.find(sl.req->model);
            hdr.adapter = it == snap->adapters.end() ? EPPK_ADAPTER_BASE : it->second;
            hdr.n_blocks = sl.n_blocks;
            std::memcpy(rowp + i * stride_, &hdr, sizeof hdr);
            std::memcpy(rowp + i * stride_ + 8, sl.hashes.data(), (size_t)sl.n_blocks * 8u);
          }
          if (pipelined) {
            // batch k + 1 goes to the device; batch k (the other set) is collected below, after be_mu_ is released
            if (be_->StageBegin(next_set, (uint32_t)n, opt_.learn_prefixes) == EPPK_OK) {
              begun = true;
              ++inflight_;
            } else {
              failed = true;
            }
          } else {
            // (the batch in flight was collected above, before this batch's masks were final)
            picks.resize(n * k);
            scores.resize(n * k);
            int rc;
            if (staged && plain) {
              rc = be_->PickStaged((uint32_t)n, picks.data(), scores.data());
            } else {
              if (staged) { rows.assign(rowp, rowp + n * stride_); rowp = rows.data(); }     // (the other entry points copy FROM caller memory INTO that buffer)
              rc = k == 1 ? be_->PickBatch(rowp, (uint32_t)n, any_mask ? mask.data() : nullptr, picks.data(), scores.data())
                          : be_->PickTopK(rowp, (uint32_t)n, any_mask ? mask.data() : nullptr, k, picks.data(), scores.data());
            }
            failed = rc != EPPK_OK;
            if (!failed) Resolve(batch, *snap, picks.data(), k);
            if (!failed && opt_.learn_prefixes) {  // index[hash[r][i]] gains pick[r], for every block of every routed request
              learn_h.clear();
              learn_p.clear();
              for (size_t i = 0; i < n; ++i) {
                const int32_t p = picks[i * k];
                if (p < 0) continue;
                eppk_req_hdr hdr;
                std::memcpy(&hdr, rowp + i * stride_, sizeof hdr);
                const uint64_t* h = (const uint64_t*)(rowp + i * stride_ + 8);
                for (uint32_t b = 0; b < hdr.n_blocks; ++b) { learn_h.push_back(h[b]); learn_p.push_back((uint32_t)p); }
              }
              if (!learn_h.empty() && be_->IndexInsert(learn_h.data(), learn_p.data(), (uint32_t)learn_h.size()) != EPPK_OK)
                learn_drops_.fetch_add(1, std::memory_order_relaxed);  // (e.g. EPPK_ERR_INDEX_FULL: the picks themselves stand)
            }
          }
          if (begun) {                          // hand the batch over to the pipeline; the previous one is collected now
            InFlight prev = std::move(fly);
            fly.batch = batch; fly.snap = snap; fly.set = next_set; fly.active = true;
            next_set ^= 1u;
            if (prev.active) { bg.unlock(); Collect(prev, picks, scores); bg.lock(); }
          }
          // Ageing between two batches.  With a batch in flight: the epoch tick is host side only and the eviction goes onto the
          // device behind that batch (IndexEvictOlderAsync: no count, the pipeline keeps running); a backend without it -- and the
          // synchronous form, which reports how many hashes went -- needs the pipeline empty: collect first.
          if (opt_.index_epoch_interval.count() > 0 && std::chrono::steady_clock::now() >= next_tick) {
            next_tick = std::chrono::steady_clock::now() + opt_.index_epoch_interval;
            uint32_t epoch = 0, gone = 0;
            bool ticked = false, done = false;
            if (fly.active) {
              ticked = be_->IndexAdvanceEpoch(&epoch) == EPPK_OK;
              if (ticked && epoch > opt_.index_keep_epochs) {
                done = be_->IndexEvictOlderAsync(epoch - opt_.index_keep_epochs) == EPPK_OK;
                if (done) evictions_async_.fetch_add(1, std::memory_order_relaxed);
              } else {
                done = ticked;
              }
            }
            if (!done) {
              if (fly.active) { bg.unlock(); Collect(fly, picks, scores); bg.lock(); }
              if ((ticked || be_->IndexAdvanceEpoch(&epoch) == EPPK_OK) && epoch > opt_.index_keep_epochs &&
                  be_->IndexEvictOlder(epoch - opt_.index_keep_epochs, &gone) == EPPK_OK)
                evicted_.fetch_add(gone, std::memory_order_relaxed);
            }
          }
        }
      }
      if (!begun) Finish(batch, failed);
      g.lock();
    }
  }

  std::unique_ptr<Backend> be_;
  GpuPickerOptions opt_;
  size_t stride_;
  RoundRobinPicker rr_;
  std::mutex mu_;     // queue + snap_ pointer
  std::mutex be_mu_;  // backend context + snapshot consistency; always taken BEFORE mu_
  std::condition_variable be_cv_;   // with be_mu_: the pipeline drained (inflight_ == 0) / the publisher is through (publish_waiting_ == 0)
  uint32_t inflight_ = 0;           // pipelined batches between StageBegin and StageEnd (be_mu_)
  uint32_t publish_waiting_ = 0;    // PublishSnapshot calls waiting for the pipeline to drain (be_mu_)
  std::condition_variable cv_;
  std::vector<Slot*> queue_;
  std::shared_ptr<Snapshot> snap_;
  // stable_slots (guarded by be_mu_): "ip:port" -> slot across snapshots, freed slots, high-water mark
  std::unordered_map<std::string, uint32_t> slot_of_;
  std::vector<uint32_t> free_slots_;
  uint32_t n_slots_ = 0;
  bool stop_ = false;
  std::atomic<uint64_t> batches_{0}, fail_open_count_{0}, largest_batch_{0}, learn_drops_{0}, evicted_{0}, evictions_async_{0};
  std::thread th_;  // last member: started after everything above is constructed
};

inline eppk_cfg MakeCfg(const SchedulerProfile& profile, const GpuPickerOptions& opt, uint32_t index_slots, int device) {
  eppk_cfg cfg;
  std::memset(&cfg, 0, sizeof cfg);
  cfg.struct_size = sizeof cfg;
  cfg.device = device;
  cfg.max_pods = opt.max_pods;
  cfg.max_blocks = opt.max_blocks;
  cfg.max_batch = opt.max_batch;
  cfg.index_slots = index_slots;
  cfg.n_scorers = (uint32_t)profile.scorers.size();
  for (size_t k = 0; k < profile.scorers.size() && k < EPPK_MAX_SCORERS; ++k) {
    cfg.chain[k].kind = (uint32_t)profile.scorers[k].kind;
    cfg.chain[k].weight = profile.scorers[k].weight;
  }
  return cfg;
}

// ---- the scheduling cycle with several profiles (interface.go:55-111, README.md:55-85) -------------------------------------
//
//   Scheduler.Schedule:  loop { profiles = ProfileHandler.Pick(request, profiles, results so far); run each: Filter* -> Score* ->
//   Picker }  until Pick returns nothing;  then ProfileHandler.ProcessResults -> SchedulingResult{ProfileResults, PrimaryProfileName}.
//
// Batched here: a round asks the handler for every request of the batch, groups the requests by profile and runs each profile
// ONCE over its group -- one libeppk context per profile (a context IS one SchedulerProfile: weighted scorer chain + picker), one
// fused kernel per profile and round.  A profile's Filter plugins (`is-prefill`, `has-required-accelerator`, ... of
// examples/example.yaml) depend on the endpoint only, so they are applied when the snapshot is published: endpoints a profile
// filters out are HOLES of that profile's snapshot (EPPK_POD_INACTIVE) -- never candidates, at no per-request cost.
// Request-dependent filtering (the subset hint) stays a per-request candidate mask (GpuPicker above).

struct Request {  // interface.go:35-44
  std::string request_id, target_model, prompt;
  std::map<std::string, std::string> headers;
};

struct ScoredEndpoint {  // interface.go:50-53
  const Endpoint* endpoint = nullptr;
  double score = 0.0;
};

struct SchedulingResult {  // interface.go:81-84
  std::map<std::string, std::vector<const Endpoint*>> profile_results;
  std::string primary_profile_name;
};

enum class PickerKind { BestScore, RandomTopK };   // examples/example.yaml `selection: best-score | random-top-3`

struct ProfileSpec {
  std::string name;
  std::function<bool(const Endpoint&)> filter;   // conjunction of the profile's Filter plugins (interface.go:113-118); empty = all
  std::vector<WeightedScorer> scorers;           // interface.go:132-135, order = summation order
  PickerKind picker = PickerKind::BestScore;     // interface.go:137-142
  uint32_t k = 3;                                // random-top-k
};

class ProfileHandler {  // interface.go:91-111
 public:
  virtual ~ProfileHandler() = default;
  // the profiles to run next for this request, given what has run already (empty = the cycle is over)
  virtual std::vector<std::string> Pick(const Request& request, const std::vector<std::string>& profiles,
                                        const std::map<std::string, std::vector<ScoredEndpoint>>& execution_results) = 0;
  virtual SchedulingResult ProcessResults(const Request& request, const std::map<std::string, std::vector<ScoredEndpoint>>& profile_results) = 0;
};

// `profileSelection: disagg-token-length` of examples/example.yaml: every request runs the decode profile; a prompt of at least
// `threshold` characters additionally runs the prefill profile (prefill / decode disaggregation); decode is primary.
class DisaggTokenLengthHandler : public ProfileHandler {
 public:
  DisaggTokenLengthHandler(std::string prefill, std::string decode, size_t threshold) : prefill_(std::move(prefill)), decode_(std::move(decode)), threshold_(threshold) {}
  std::vector<std::string> Pick(const Request& request, const std::vector<std::string>&, const std::map<std::string, std::vector<ScoredEndpoint>>& done) override {
    if (!done.empty()) return {};
    std::vector<std::string> run{decode_};
    if (request.prompt.size() >= threshold_) run.push_back(prefill_);
    return run;
  }
  SchedulingResult ProcessResults(const Request&, const std::map<std::string, std::vector<ScoredEndpoint>>& results) override {
    SchedulingResult out;
    for (const auto& kv : results) {
      auto& v = out.profile_results[kv.first];
      for (const ScoredEndpoint& e : kv.second) v.push_back(e.endpoint);
    }
    out.primary_profile_name = decode_;
    return out;
  }

 private:
  std::string prefill_, decode_;
  size_t threshold_;
};

class Scheduler {  // interface.go:55-66
 public:
  struct Options { uint32_t max_pods = 4096, max_blocks = 32, max_batch = 4096, block_chars = 64, index_slots = 0; int device = 0; };

  // one context per profile; the handler is borrowed
  Status Configure(const std::vector<ProfileSpec>& profiles, ProfileHandler* handler, const Options& opt) {
    opt_ = opt; handler_ = handler; profiles_.clear(); names_.clear();
    for (const ProfileSpec& ps : profiles) {
      SchedulerProfile sp; sp.scorers = ps.scorers;
      GpuPickerOptions go; go.max_pods = opt.max_pods; go.max_blocks = opt.max_blocks; go.max_batch = opt.max_batch;
      eppk_cfg cfg = MakeCfg(sp, go, opt.index_slots, opt.device);
      eppk_ctx* c = nullptr;
      if (eppk_create(&cfg, &c) != EPPK_OK) return {Code::Internal, std::string("profile ") + ps.name + ": " + eppk_last_error(nullptr)};
      profiles_.push_back(Prof{ps, std::shared_ptr<eppk_ctx>(c, eppk_destroy)});
      names_.push_back(ps.name);
    }
    return {};
  }

  // endpoints[i] is candidate index i in every profile; a profile's filter turns the endpoints it rejects into holes
  Status PublishSnapshot(const std::vector<Endpoint>& endpoints, const std::vector<eppk_pod_row>& rows,
                         const std::unordered_map<std::string, int32_t>& adapters, uint64_t epoch) {
    if (endpoints.size() != rows.size()) return {Code::Internal, "endpoints/rows size mismatch"};
    endpoints_ = endpoints; adapters_ = adapters;
    for (Prof& p : profiles_) {
      std::vector<eppk_pod_row> r = rows;
      for (size_t i = 0; i < r.size(); ++i)
        if (p.spec.filter && !p.spec.filter(endpoints_[i])) r[i].flags |= EPPK_POD_INACTIVE;
      if (eppk_snapshot_publish(p.ctx.get(), r.data(), (uint32_t)r.size(), epoch) != EPPK_OK) return {Code::Internal, eppk_last_error(p.ctx.get())};
    }
    return {};
  }

  // "hash(chunk i): append server" for one profile's prefix index (0602-…/README.md:101-108)
  Status IndexInsert(const std::string& profile, const uint64_t* hashes, const uint32_t* pods, uint32_t n) {
    for (Prof& p : profiles_)
      if (p.spec.name == profile) return eppk_index_insert(p.ctx.get(), hashes, pods, n) == EPPK_OK ? Status{} : Status{Code::Internal, eppk_last_error(p.ctx.get())};
    return {Code::Internal, "unknown profile"};
  }

  // Scheduler.Schedule for a batch; statuses[r] = Unavailable when the primary profile left request r without an endpoint
  // ("the framework will return an error ... if the endpoints are filtered to zero", interface.go:114-115).
  Status ScheduleBatch(const std::vector<Request>& requests, uint64_t seed, std::vector<SchedulingResult>* out, std::vector<Status>* statuses) {
    const size_t n = requests.size();
    std::vector<std::map<std::string, std::vector<ScoredEndpoint>>> results(n);
    const size_t stride = 8u + 8u * opt_.max_blocks;
    for (int round = 0; round < 16; ++round) {        // (a handler that never stops is cut off)
      std::map<std::string, std::vector<uint32_t>> group;
      for (size_t r = 0; r < n; ++r)
        for (const std::string& name : handler_->Pick(requests[r], names_, results[r]))
          if (!results[r].count(name)) group[name].push_back((uint32_t)r);
      if (group.empty()) break;
      for (Prof& p : profiles_) {
        auto g = group.find(p.spec.name);
        if (g == group.end()) continue;
        const std::vector<uint32_t>& idx = g->second;
        for (size_t lo = 0; lo < idx.size(); lo += opt_.max_batch) {
          const uint32_t m = (uint32_t)std::min<size_t>(opt_.max_batch, idx.size() - lo);
          // best-score profiles build their rows straight in the context's pinned staging buffer (eppk_host_staging: no host copy
          // between here and the DMA); the other pickers take a plain buffer
          uint8_t* rows = nullptr;
          bool staged = false;
          if (p.spec.picker == PickerKind::BestScore) {
            void* st = nullptr;
            staged = eppk_host_staging(p.ctx.get(), &st, nullptr) == EPPK_OK && st != nullptr;
            if (staged) { rows = (uint8_t*)st; std::memset(rows, 0, (size_t)m * stride); }
          }
          if (!staged) { rows_.assign((size_t)m * stride, 0); rows = rows_.data(); }
          for (uint32_t i = 0; i < m; ++i) {
            const Request& rq = requests[idx[lo + i]];
            eppk_req_hdr hdr;
            auto it = adapters_.find(rq.target_model);
            hdr.adapter = it == adapters_.end() ? EPPK_ADAPTER_BASE : it->second;
            const int nb = eppk_hash_prompt((const uint8_t*)rq.target_model.data(), rq.target_model.size(), (const uint8_t*)rq.prompt.data(), rq.prompt.size(),
                                            opt_.block_chars, (uint64_t*)(rows + (size_t)i * stride + 8), opt_.max_blocks);
            hdr.n_blocks = nb < 0 ? 0u : (uint32_t)nb;
            std::memcpy(rows + (size_t)i * stride, &hdr, sizeof hdr);
          }
          picks_.resize(m); scores_.resize(m);
          // the random-top-k rule hashes a request's index in the batch handed to the library: this profile's group, in request order
          const int rc = staged ? eppk_pick_batch_staged(p.ctx.get(), m, 0, picks_.data(), scores_.data())
                         : p.spec.picker == PickerKind::BestScore
                             ? eppk_pick_batch(p.ctx.get(), rows, m, nullptr, picks_.data(), scores_.data())
                             : eppk_pick_random_topk(p.ctx.get(), rows, m, nullptr, p.spec.k, seed + lo, picks_.data(), scores_.data());
          if (rc != EPPK_OK) return {Code::Internal, eppk_last_error(p.ctx.get())};
          for (uint32_t i = 0; i < m; ++i) {
            auto& res = results[idx[lo + i]][p.spec.name];     // (present even when empty: the profile has run)
            if (picks_[i] >= 0) res.push_back(ScoredEndpoint{&endpoints_[(size_t)picks_[i]], scores_[i]});
          }
        }
      }
    }
    out->resize(n); statuses->assign(n, Status{});
    for (size_t r = 0; r < n; ++r) {
      (*out)[r] = handler_->ProcessResults(requests[r], results[r]);
      auto pr = (*out)[r].profile_results.find((*out)[r].primary_profile_name);
      if (pr == (*out)[r].profile_results.end() || pr->second.empty()) (*statuses)[r] = {Code::Unavailable, "no endpoints available"};
    }
    return {};
  }

  eppk_ctx* context(const std::string& profile) {
    for (Prof& p : profiles_) if (p.spec.name == profile) return p.ctx.get();
    return nullptr;
  }

 private:
  struct Prof { ProfileSpec spec; std::shared_ptr<eppk_ctx> ctx; };
  Options opt_;
  ProfileHandler* handler_ = nullptr;
  std::vector<Prof> profiles_;
  std::vector<std::string> names_;
  std::vector<Endpoint> endpoints_;
  std::unordered_map<std::string, int32_t> adapters_;
  std::vector<uint8_t> rows_;
  std::vector<int32_t> picks_;
  std::vector<double> scores_;
};

}  // namespace eppk_host
