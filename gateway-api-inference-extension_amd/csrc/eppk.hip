// eppk.hip — C ABI of libeppk (include/eppk.h): context, snapshot re-layout, launches.
//
// The product path.  There is NO CPU fallback in this file: if HIP is unavailable every entry point
// that needs the device fails with EPPK_ERR_DEVICE.  Nothing here includes, links or calls oracle/.
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off (binary64 mul/add never fused;
// SEMANTICS.md §2) — see __graft_entry__.build().
#define EPPK_MAIN_UNIT
#include "eppk_kernels.hip.h"
#include "eppk_pick_inst.hip.h"

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/eppk.h"

using namespace eppk;

namespace {

std::mutex g_err_mu;
std::string g_create_err;
std::mutex g_lds_mu;
std::map<std::pair<int, const void*>, size_t> g_lds_limit;   // (device, kernel) -> dynamic-LDS limit set so far (occupancy_of)

// One device allocation per snapshot buffer; the arrays below are views into it (fast kernel: one buffer descriptor).
struct SnapBuf {
  uint8_t*  blob = nullptr;
  double*   base = nullptr;
  double*   post[2] = {nullptr, nullptr};   // products of the pod-only scorers behind LORA / PREFIX (GEN fast path)
  uint32_t* queue = nullptr;
  double*   kv = nullptr;
  void*     thi_t = nullptr;   // [129][64] LW LoRA tier planes (eppk_kernels.hip.h: KSnap)
  void*     tlo_t = nullptr;
  void*     qmin_t = nullptr;  // [64] LW pods at the global min / max queue depth
  void*     qmax_t = nullptr;
  void*     act_t = nullptr;   // [64] LW active slots (holes clear)
  uint64_t* nat = nullptr;     // [3][64] u64 natural-layout active / qmin / qmax sets (masked list routes)
  uint32_t* qrange = nullptr;  // [2] min / max queue depth over the active pods
  double*   topv = nullptr;   // [129][64]
  uint32_t* topi = nullptr;   // [129][64]
};

struct SnapLayout {            // byte offsets inside a snapshot blob
  size_t base = 0, post0 = 0, post1 = 0, queue = 0, kv = 0, thi = 0, tlo = 0, qmin = 0, qmax = 0, act = 0, nat = 0, qrange = 0, topv = 0, topi = 0, bytes = 0;
};

}  // namespace

struct eppk_ctx {
  eppk_cfg cfg{};
  hipStream_t stream = nullptr;
  int num_cu = 256;

  // geometry
  int lw_bytes = 8;       // lane word: 2 (P<=1024), 4 (P<=2048), 8 (P<=4096)
  int npl = 6;            // counter bit planes: 6 (B<=63) or 9 (B<=256)
  uint32_t pwn = 0;       // per-wave prefix table entries
  uint32_t stride = 0;    // request row stride in bytes
  uint32_t jmax = 0;      // ceil(max_pods/64)

  // chain analysis
  bool canonical = false; // [QUEUE|KV]* ++ tail, tail ⊂ {LORA, PREFIX} each at most once
  uint32_t n_lead = 0;
  bool lead_queue = false;  // a QUEUE scorer is fused into base[]
  bool has_l = false, has_p = false, p_first = false;
  bool gen = false;        // pod-only scorers behind LORA / PREFIX: interpreted tail (GEN instantiation)
  KChain postc{};          // those scorers (n <= 2)
  KTail tail{};
  KChain kchain{};
  double* pterm = nullptr;  // device [(B+1)*(B+1)] exact prefix terms (fast path, max_blocks <= 64)
  uint32_t pterm_ld = 0;

  // snapshot (double buffered: a publish never overwrites the rows an in-flight pick reads)
  SnapBuf snap[2];
  SnapLayout lay;
  int cur = 0;
  bool have_snapshot = false;
  uint32_t n_pods = 0;
  uint64_t epoch = 0;
  uint32_t assumed_epochs = 0;     // SEMANTICS.md §2b (eppk_set_assumed_load); 0 = off
  int32_t* d_rs_pick = nullptr; double* d_rs_score = nullptr; size_t rs_cap = 0;   // fallback lists of eppk_pick_random_topk

  // prefix index
  uint64_t* keys = nullptr;
  void* bitmaps = nullptr;
  uint32_t slots = 0, shift = 0, limit = 0;
  size_t rows_bytes = 0, index_bytes = 0;   // rows | keys in one allocation
  uint32_t* rstamps = nullptr;              // [2] exact stamps of the two reserved rows (hashes 0 / ~0: no bucket header); every other key's stamp is
                                            // a TAG in its bucket header (eppk_kernels.hip.h: "stamps as tags")
  uint32_t min_live = 0;                    // no live key is stamped before this epoch (the largest eviction horizon so far): the tags' window
  uint32_t* lists = nullptr;                // [slots + 4][16] short pod lists: where a set with at most kListCap members lives (always maintained),
                                            // then the SET TABLE [sets_cap][16]: interned copies of the listed sets, named by the set ids in the bucket lines
  uint32_t sets_cap = 0;                    // lines of the set table (a power of two)
  uint32_t* set_ctl = nullptr;              // device [2]: set-table lines in use, sets that found no line
  uint32_t* h_set_report = nullptr;         // pinned [2]: the same two, as index_canon_kernel last left them (read without synchronising)
  uint32_t* h_set_report_dev = nullptr;
  uint32_t set_fail_seen = 0;               // sets without a line as of the last rebuild
  uint32_t set_passes = 0;                  // canon passes since the last rebuild (a table whose LIVE sets fill half of it is not rebuilt on every update)
  bool list_routes = true;                  // EPPK_LISTS=0: the pick kernels' list routes are off (every request takes the dense route)
  uint32_t* sortwl = nullptr;               // work list of index_canon_kernel: cursors[2] | lost | arrived | slots[sortwl_cap]
  uint32_t sortwl_cap = 0, sort_uses = 0;
  eppk::IxLaunch* d_ixl = nullptr;          // the capacity verdict of the insert launch in flight (index_budget_kernel)
  uint32_t index_epoch = 1;
  unsigned long long* stats = nullptr;  // device [4 + kStatBanks*2*kStatSlots]: scratch, live keys, non-empty words, dropped inserts, then
                                        // banks of per-wave {hits, lookups}: consecutive launches use different banks, so pick
                                        // kernels overlapping on two streams never share a slot
  uint32_t stat_bank = 0;
  uint32_t* d_status = nullptr;         // sticky launch-status flags (eppk_launch_status)
  unsigned long long* ixc = nullptr;    // [kIxShards][8] sharded index counters: live keys, non-empty words, dropped inserts, evicted

  // subset filter on the device: fingerprints of the published endpoints (eppk_snapshot_set_addresses), entry staging
  uint64_t* d_at = nullptr; uint32_t* d_av = nullptr; uint32_t at_slots = 0; bool have_addrs = false; uint32_t addr_n = 0;
  uint64_t* d_sk = nullptr; uint32_t* d_so = nullptr; size_t sk_cap = 0, so_cap = 0;

  // staging for the host-buffer entry point
  void* h_reqs = nullptr; uint64_t* h_mask = nullptr; int32_t* h_pick = nullptr; double* h_score = nullptr;
  void* d_reqs = nullptr; uint64_t* d_mask = nullptr; int32_t* d_pick = nullptr; double* d_score = nullptr;
  void* h_reqs_dev = nullptr; uint64_t* h_mask_dev = nullptr; int32_t* h_pick_dev = nullptr; double* h_score_dev = nullptr;   // the pinned buffers as the device addresses them
  void* d_tmp = nullptr; size_t d_tmp_bytes = 0;  // index insert staging
  // the pipelined host path (eppk_pick_stage_*): per set its own pinned + device buffers, stream and events
  struct StageSet {
    hipStream_t st = nullptr; hipEvent_t picked = nullptr;
    hipStream_t st_copy = nullptr; hipEvent_t copied = nullptr; bool copy_pending = false;   // zero-copy pick + LEARN: the rows' device copy rides beside the pick
    uint32_t* h_bad = nullptr; uint32_t* h_bad_dev = nullptr; hipStream_t st_check = nullptr; hipEvent_t checked = nullptr;   // device-side row check (eppk_ctx::RowCheck)
    bool check_pending = false, check_side = false;
    void* h_reqs = nullptr; uint64_t* h_mask = nullptr; int32_t* h_pick = nullptr; double* h_score = nullptr;
    void* d_reqs = nullptr; uint64_t* d_mask = nullptr; int32_t* d_pick = nullptr; double* d_score = nullptr;
    void* h_reqs_dev = nullptr; uint64_t* h_mask_dev = nullptr; int32_t* h_pick_dev = nullptr; double* h_score_dev = nullptr;
    uint32_t n = 0; bool busy = false, had_mask = false;
    uint32_t row_base = 0;       // a group member's set: batch index of the first row its device-side check looked at
    bool resident = false; uint32_t res_unit = 0, res_seq = 0;   // the batch between begin and end was rung into a resident workgroup (no launch)
  };
  StageSet stage[2];
  hipStream_t learn_words_stream = nullptr; hipEvent_t learn_words_free = nullptr;   // d_learn is ONE buffer: a pick + update pair on another stream waits for the last pair's update
  hipEvent_t learned = nullptr; bool learn_pending = false;   // recorded behind the latest LEARN update; every later pick, index update and
                                                              // publish of this context -- on whatever stream -- is ordered behind it (learn_fence)
  // The resident small-batch kernel (EPPK_RESIDENT=1; eppk_kernels.hip.h: pick_resident_kernel): pinned control block, device argument
  // block, a stream of its own; res_seq = the last doorbell value rung.
  bool resident_on = false, res_args_dirty = true;
  uint32_t resident_quad_from = 8;           // smallest batch that rings the quad form (EPPK_RESIDENT_QUAD_FROM).  Same box, p50 of the bare call,
                                             // fast / quad form: 1 request 11.0 / 11.9 us, 4: 11.1 / 11.9, 8: 12.3 / 11.9, 16: 13.3 / 12.0, 32: 15.5 / 13.4,
                                             // 64: 27.9 / 19.4 (scripts/res_sweep.py, profiles/r04_resident_latency.txt)
  uint32_t resident_max = 32, res_gen = 0;   // (EPPK_RESIDENT_MAX; 64 by default where the quad form of the kernel runs: eppk_create)
  // Two resident workgroups at most, one per FORM of the kernel, each with its control block and stream: [0] pick_fast_kernel's body
  // (a wavefront per request: up to 16 requests), [1] pick_quad_kernel's body (four requests per wavefront: beyond 16, where that
  // route exists).  A batch rings the one that suits it; each leaves by itself when idle and is started again on demand.
  static constexpr uint32_t kResUnits = 7u, kResSlots = 4u;
  hipStream_t res_slot_stream[kResSlots] = {nullptr, nullptr, nullptr, nullptr};   // high-priority streams: one hardware queue per running resident kernel
  int32_t res_slot_unit[kResSlots] = {-1, -1, -1, -1};                               // which unit runs on the slot (-1: free)
  uint64_t res_clock = 0;                                                            // LRU clock of the units (ResidentUnit::last_rung)
  uint8_t* d_res_rows = nullptr; uint32_t* d_res_learn = nullptr; uint32_t* d_res_sortwl = nullptr;   // LEARN units: row copies, learn words, sort work lists (per unit)
  struct ResidentUnit {
    eppk::ResidentCtl* h_ctl = nullptr; eppk::ResidentCtl* h_ctl_dev = nullptr; hipStream_t stream = nullptr;
    bool running = false; uint32_t seq = 0;   // seq = the last doorbell value rung
    bool pending = false; uint32_t pending_seq = 0;   // a doorbell rung (eppk_pick_stage_begin) and not collected yet
    bool updating = false; uint32_t update_seq = 0;   // LEARN units: the index update behind that doorbell has not been seen finished yet
    int32_t slot = -1; uint64_t last_rung = 0;        // the stream slot it runs on; when it was last rung (the slots are handed out LRU)
    uint32_t misses = 0;                              // calls in a row that wanted this unit while every slot was taken (resident_admit)
  } res[kResUnits];
  eppk::ResidentArgs* d_res_args = nullptr;
  uint32_t* d_res_wl = nullptr; uint32_t res_wl_cap = 0;     // the resident workgroup's work list (its pick_quad_kernel form): total[32] | cnt[16] | list[16][cap]
  uint64_t res_batches = 0, res_starts = 0;
  uint32_t* d_learn = nullptr; size_t learn_cap = 0;          // learn words of the pick in front of a LEARN update (pick_quad_kernel<..., LEARN>)
  bool quiet_rows = false;        // a host-buffer launch is being enqueued: its kernels raise "row out of range" on a word of their own
                                  // (the call reports the row itself), not on the sticky flag of the *_device entry points
  uint32_t host_flags = 0;        // sticky launch-status flags raised by the host side (EPPK_LAUNCH_LEARN_FAILED)
  void* d_tk_reqs = nullptr; uint64_t* d_tk_mask = nullptr; int32_t* d_tk_pick = nullptr; double* d_tk_score = nullptr;  // eppk_pick_topk
  eppk_pod_row* h_rows = nullptr; eppk_pod_row* d_rows = nullptr;  // raw pod rows of a publish (pinned staging + device copy)

  // measurement
  bool prof = false;
  uint32_t prof_every = 1, prof_tick = 0;   // eppk_profile_enable(on = N > 1): only every Nth pick launch carries events and probe counts
  std::vector<hipEvent_t> ev;  // start/stop pairs
  size_t ev_used = 0;
  uint64_t fixed_bytes = 0;  // per-launch request/pod/pick bytes accumulated while profiling
  uint32_t launches = 0;

  uint64_t h_holes[64] = {0};         // holes of the last published snapshot, lane-transposed (bit j of word l = pod j*64+l)
  void* d_rm = nullptr;               // [64] LW device copy of a scrub mask

  hipEvent_t last_done = nullptr;     // completion event riding on the most recent pick launch (profiling), else null
  hipStream_t last_stream = nullptr;  // the stream of that launch
  hipEvent_t wait_ev = nullptr;       // eppk_stream_wait_pick's own event (when the launch carried none)

  // cached launch geometry: resident workgroups per CU of every (kernel, workgroup size, LDS bytes) this context has launched.  Setting
  // the dynamic-LDS attribute and asking for the occupancy are host calls of ~10 us each: a dispatcher that alternates between
  // masked and unmasked batches, or between learning and plain picks, would pay them at every switch with a one-entry cache.
  struct Occ { const void* fn; uint32_t threads; size_t lds; int per_cu; };
  std::vector<Occ> occ;
  // pick_quad_kernel (four requests per wavefront) + the work list of what it defers to pick_fast_kernel<WL>
  // (eppk_kernels.hip.h: KWork)
  bool quad_on = true;            // EPPK_QUAD=0 switches it off (every request through pick_fast_kernel)
  uint32_t quad_min = 4096;       // smallest batch that takes the route (EPPK_QUAD_MIN overrides).  With a second launch behind every batch
                                  // the route only paid off from ~24k requests on; in its one-launch form it wins from 4k on (C5 rows, us
                                  // per step, fast kernel vs quad: 4k 6.8 / 5.9, 8k 8.4 / 6.7, 16k 12.2 / 9.1: profiles/r03_j_quad_min.txt)
  uint32_t quad_threads = 512;    // EPPK_QUAD_THREADS overrides (tuning knob; <= the kernel's launch bound)
  // Host-buffer picks of at most this many requests run ZERO-COPY: the kernel reads the request rows (and the mask) straight out of
  // the pinned staging buffer and writes picks and scores straight into the pinned result buffers -- one launch instead of
  // upload + launch + two downloads.  A small batch is all latency -- host-observed p50 of eppk_pick_batch_staged, rows freshly written
  // by the caller (C5 snapshot, profiles/r03_z_small_batch_latency.txt): 16 requests 25.8 -> 21.1 us, 128: 33.2 -> 21.7, 2048: 44.9 ->
  // 33.  The shader reads host memory at ~28 GB/s, the copy engine at 55: once the copy path had lost its host validation loop, its
  // chunking and its download copies, it won from ~3000 requests on (4096: 54.7 vs 62.6 us; 2048: 41.7 vs 33.1).
  // EPPK_ZERO_COPY_MAX overrides, 0 = off.
  uint32_t zero_copy_max = 3072;
  // Request rows that arrive in PINNED memory are range-checked ON THE DEVICE (rows_check_kernel: a thread per row header, the lowest
  // bad row into a pinned word by atomicMin) instead of by a host loop in front of the launch: that loop touches one cache line per row
  // -- 300-500 us for a 64k-request batch, more than the rows' PCIe time, and a good part of a mid-sized zero-copy batch.
  // The kernels give a row out of range EPPK_NO_PICK and the index update skips it, exactly as on the *_device entry points; the call
  // still fails with EPPK_ERR_ARG naming the row (from _end), and delivers nothing.  Batches of at most host_check_max rows keep the
  // host loop (cheaper than a second launch and its event: 2048 requests 33 us with the loop, 41-45 with the kernel;
  // 64k staged: 478 vs 402 us, profiles/r03_z_small_batch_latency.txt).  EPPK_HOST_CHECK_MAX overrides.
  struct RowCheck { uint32_t* h_bad = nullptr; uint32_t* h_bad_dev = nullptr; hipStream_t st = nullptr; hipEvent_t done = nullptr;
                    bool pending = false, side = false; const char* who = nullptr; };
  RowCheck check;
  uint32_t host_check_max = 2048;
  // One work-list buffer per STREAM that has launched picks (launches of one stream are ordered, so a buffer is never written
  // while an earlier launch still reads it; launches of different streams never share one).  More than kDeferSets distinct streams:
  // the later ones stay on pick_fast_kernel.
  struct DeferSet { hipStream_t st = nullptr; bool used = false; uint32_t* d = nullptr; size_t words = 0; uint32_t uses = 0; };   // d: total[2] | cnt[segs] | list[segs][cap]
  DeferSet dsets[8];
  // Reports: every quad launch owns the next slot of a ring of pinned host words; the work-list kernel stores the launch's deferred
  // count there (kReportPending until then); the host consumes the slots in order, whenever it passes by.
  volatile uint32_t* h_reports = nullptr;      // pinned [kReportRing]
  uint32_t rep_n[4096] = {0};                  // requests of the launch that owns the slot (kReportRing entries)
  uint8_t rep_masked[4096] = {0};              // ... and whether it was a masked batch
  uint64_t rep_unread = 0;                     // first launch whose report has not been consumed
  uint32_t quad_backoff = 0, quad_backoff_len = 0;
  uint32_t wl_hint = 0xFFFFFFFFu;   // decaying maximum of the recent launches' deferred counts (0xFFFFFFFF: no report seen yet): sizes the work-list pass
  bool quad_tail_on = true;         // EPPK_QUAD_TAIL=0: always the two-launch form (pick_quad_kernel + work-list pass)
  bool quad_pause_on = true;        // EPPK_QUAD_PAUSE=0: a launch that deferred a large part of its batch does not pause the route (measurement knob)
  uint64_t quad_tail_launches = 0;  // launches that took the one-launch form (pick_quad_kernel<TAIL>)
  uint64_t quad_launches = 0, quad_deferred_seen = 0;
  uint32_t fast_threads = 1024;  // workgroup size of the fast kernel (EPPK_FAST_THREADS overrides: tuning knob)
  size_t max_lds = 65536;        // LDS a workgroup may use (160 KB on gfx950)
  int max_wg_per_cu = 0;         // EPPK_MAX_WG_PER_CU: cap on resident workgroups per CU (0 = what the occupancy query allows; tuning knob)

  std::string err;
};

namespace {

constexpr uint32_t kStatBanks = 4;
constexpr uint32_t kEpochWindow = EPPK_INDEX_EPOCH_WINDOW;  // largest age (in index epochs) a live hash may reach: its stamp is an 8-bit tag (eppk_kernels.hip.h: kTagMod)
constexpr uint32_t kDeferSets = 8;       // streams with a work-list buffer of their own (eppk_ctx::dsets)
constexpr uint32_t kReportRing = 4096;   // quad launches whose deferred-count report may be outstanding (a host that enqueues far ahead of
                                         // the device: bench.py is a few hundred launches ahead; a full ring = the fast kernel for that launch)
constexpr uint32_t kReportPending = 0xFFFFFFFFu;

// Consume the reports that have arrived, in launch order; steer the back-off: a launch that deferred more than 1/4 of its batch
// (quad pass + a work-list pass over a quarter of the batch costs about what the fast kernel alone does; 1/8 for a masked batch, whose
// deferred requests are the ones that need the exact dense evaluation: 110 vs 80 us at 1/8-density masks, measured) pauses the route (the pause doubles while that keeps happening), one that deferred (almost) nothing resets the pause length.
void quad_consume_reports(eppk_ctx* c) {
  while (c->rep_unread < c->quad_launches) {
    const uint32_t slot = (uint32_t)(c->rep_unread % kReportRing);
    const uint32_t v = c->h_reports[slot];
    if (v == kReportPending) break;
    c->quad_deferred_seen += v;
    {                                      // what the next work-list passes should expect: the largest recent count, decaying by 1/8 per report
      const uint32_t dec = c->wl_hint == 0xFFFFFFFFu ? 0u : c->wl_hint - (c->wl_hint + 7u) / 8u;
      c->wl_hint = v > dec ? v : dec;
    }
    const uint32_t n = c->rep_n[slot];
    if (c->quad_pause_on && v > (c->rep_masked[slot] ? n / 8u : n / 4u)) {
      if (c->quad_backoff == 0) {
        c->quad_backoff_len = c->quad_backoff_len ? (c->quad_backoff_len < 4096u ? c->quad_backoff_len * 2u : 4096u) : 64u;
        c->quad_backoff = c->quad_backoff_len;
      }
    } else if (v <= n / 64u) {
      c->quad_backoff_len = 0;
    }
    ++c->rep_unread;
  }
}

int fail(eppk_ctx* c, int code, const std::string& msg) {
  if (c) c->err = msg;
  else { std::lock_guard<std::mutex> g(g_err_mu); g_create_err = msg; }
  return code;
}

#define HIPCHK(c, expr)                                                                          \
  do {                                                                                           \
    hipError_t e_ = (expr);                                                                      \
    if (e_ != hipSuccess)                                                                        \
      return fail((c), EPPK_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));      \
  } while (0)

inline double h_clamp01(double s) {
  if (!(s >= 0.0)) return 0.0;
  if (s > 1.0) return 1.0;
  return s;
}

KSnap make_ksnap(const eppk_ctx* c) {
  const SnapBuf& s = c->snap[c->cur];
  KSnap k{};
  k.base = s.base; k.post[0] = s.post[0]; k.post[1] = s.post[1]; k.queue = s.queue; k.kv = s.kv;
  k.thi_t = s.thi_t; k.tlo_t = s.tlo_t;
  k.topv = s.topv; k.topi = s.topi;
  // (host-buffer launches: the blob descriptor ends 8 bytes early, so that the fast kernel's "last dword" is a spare word of the blob's
  // tail instead of the sticky status word -- see eppk_ctx::quiet_rows)
  k.blob = s.blob; k.blob_bytes = (uint32_t)c->lay.bytes - (c->quiet_rows ? 8u : 0u);
  k.qmin_t = s.qmin_t; k.qmax_t = s.qmax_t; k.act_t = s.act_t; k.nat = s.nat; k.lead_queue = c->lead_queue ? 1u : 0u;
  k.pterm = c->pterm; k.pterm_ld = c->pterm_ld;
  k.n_pods = c->n_pods; k.J = (c->n_pods + 63u) / 64u;
  k.qrange = s.qrange;
  k.status = c->d_status + (c->quiet_rows ? 1 : 0);
  return k;
}

KIndex make_kindex(const eppk_ctx* c) {
  KIndex k{};
  k.keys = c->keys; k.bitmaps = c->bitmaps; k.slots = c->slots; k.shift = c->shift;
  k.small = (c->slots && c->index_bytes < (1ull << 32)) ? 1u : 0u;
  k.table_bytes = k.small ? (uint32_t)c->index_bytes : 0u;
  k.keys_off = k.small ? (uint32_t)c->rows_bytes : 0u;
  // the pick reads the lists (and the set table behind them) through one raw buffer descriptor: only while that is below 4 GiB
  k.sets_mask = c->sets_cap ? c->sets_cap - 1u : 0u;
  k.lists = (c->lists && c->list_routes && ((size_t)c->slots + 4u + c->sets_cap) * 64u < (1ull << 32)) ? c->lists : nullptr;
  k.lists_all = c->lists;
  return k;
}

// ---- kernel dispatch ---------------------------------------------------------------------------
// The pick kernels are instantiated in six translation units (eppk_pick_inst.hip.h: one per lane-word type and counter-plane
// count) so that the build compiles them in parallel; each exports one look-up function.

const void* pick_kernel_ptr(const eppk_ctx* c, bool fast, bool masked, bool topk) {
  eppk::PickVariant v{};
  v.fast = fast; v.masked = masked; v.topk = topk;
  v.big = c->slots != 0 && c->index_bytes >= (1ull << 32);   // index of 4 GiB and more: rows through wave-uniform 64-bit bases
  v.has_l = c->has_l; v.has_p = c->has_p; v.p_first = c->p_first; v.gen = c->gen;
  const bool six = c->npl == 6;
  switch (c->lw_bytes) {
    case 2: return six ? eppk::pick_kernel_u16_6(v) : eppk::pick_kernel_u16_9(v);
    case 4: return six ? eppk::pick_kernel_u32_6(v) : eppk::pick_kernel_u32_9(v);
    default: return six ? eppk::pick_kernel_u64_6(v) : eppk::pick_kernel_u64_9(v);
  }
}

// Resident workgroups per CU of `fn` at this workgroup size and dynamic-LDS size (cached per context: eppk_ctx::occ); the first
// use of a combination also raises the kernel's dynamic-LDS limit.
int occupancy_of(eppk_ctx* c, const void* fn, uint32_t threads, size_t lds, int* per_cu_out) {
  for (const eppk_ctx::Occ& o : c->occ)
    if (o.fn == fn && o.threads == threads && o.lds == lds) { *per_cu_out = o.per_cu; return EPPK_OK; }
  {   // the kernel's dynamic-LDS limit is a property of the FUNCTION on this device, shared by every context of the process: raised
      // when a launch needs more, never lowered (another context's larger launch geometry of the same kernel stays valid)
    std::lock_guard<std::mutex> g(g_lds_mu);
    size_t& limit = g_lds_limit[{c->cfg.device, fn}];
    if (lds > limit) {
      HIPCHK(c, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      limit = lds;
    }
  }
  int per_cu = 0;
  HIPCHK(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, (int)threads, lds));
  if (c->max_wg_per_cu && per_cu > c->max_wg_per_cu) per_cu = c->max_wg_per_cu;
  if (per_cu < 1) per_cu = 1;
  c->occ.push_back({fn, threads, lds, per_cu});
  *per_cu_out = per_cu;
  return EPPK_OK;
}

// topk == 1: the pick; topk > 1: ordered fallbacks (d_pick / d_score hold n_reqs * topk entries): extra selection rounds of the
// fast kernel for fused chains, the TOPK generic kernel otherwise
template <typename F> int by_lane_word(const eppk_ctx* c, F&& f);

// The post-route update of an EPPK_PICK_LEARN batch (budget, insert and list-sort kernels) runs on its staging set's PRIVATE stream and
// may still be running when eppk_pick_stage_end returns.  Whatever touches the index or reads it afterwards -- a pick, an insert, an
// eviction, a removal, a trim, a clear, a publish (index scrub), the self check -- is ordered behind it here, on the stream it is
// about to use: the update's kernels assume that nothing disappears while they run and share one sort work list and capacity verdict.
int resident_drain(eppk_ctx* c);
int learn_fence(eppk_ctx* c, hipStream_t st) {
  // (a small batch rung into a resident workgroup by eppk_pick_stage_begin and not collected yet reads index and snapshot OUTSIDE
  // stream order: it is answered before anything of this context is launched that could change them)
  if (c->resident_on) { const int rcd = resident_drain(c); if (rcd) return rcd; }
  if (!c->learn_pending) return EPPK_OK;
  if (hipEventQuery(c->learned) == hipSuccess) { c->learn_pending = false; return EPPK_OK; }
  HIPCHK(c, hipStreamWaitEvent(st, c->learned, 0));
  return EPPK_OK;
}

int resident_park(eppk_ctx* c);

// d_learn (nullable): where pick_quad_kernel<..., LEARN> leaves its learn words for the index update that follows (learn_picks);
// *wrote_learn says whether this launch took that route and wrote them (else the update gets no words and takes the whole path).
int launch_pick(eppk_ctx* c, const void* d_reqs, uint32_t n_reqs, const uint64_t* d_mask, int32_t* d_pick,
                double* d_score, hipStream_t st, uint32_t topk = 1, uint32_t* d_learn = nullptr, bool* wrote_learn = nullptr) {
  if (wrote_learn) *wrote_learn = false;
  const bool masked = d_mask != nullptr;
  // masked batches use the fast kernel's MASKED instantiation, indexes of 4 GiB and more its BIG one
  const bool fast = c->canonical;   // (ordered fallbacks: extra selection rounds of the same kernel; generic TOPK kernel otherwise)
  const void* fn = pick_kernel_ptr(c, fast, masked, topk > 1);   // (replaced by its work-list instantiation behind pick_quad_kernel)
  KSnap sn = make_ksnap(c);
  KIndex ix = make_kindex(c);
  const uint32_t threads = fast ? c->fast_threads : 512u, wpb = threads / 64;
  // fast kernel LDS: base[J*64] f64 + the exact prefix-term table; generic: queue/kv + per-wave ratio tables
  uint32_t pwn = c->pwn;
  size_t lds;
  if (fast) {
    pwn = (c->pterm && c->has_p) ? (c->cfg.max_blocks + 1u) * c->pterm_ld : 0u;
    lds = (size_t)sn.J * 64u * 8u + 32u + (size_t)pwn * 8u + (c->gen ? (size_t)sn.J * 64u * 16u : 0u) +   // base | lw[4] | pterm | post0 | post1
          (size_t)wpb * 64u * (size_t)c->lw_bytes;                                                         // | 64 lane words per wavefront (set_from_list)
    if (c->has_p && c->npl == 6 && topk == 1 && ix.lists) {                                                // | per-wave pod histogram (SPARSE)
      const size_t hist = (size_t)wpb * sn.J * 64u;
      if (lds + hist <= c->max_lds) lds += hist;
      else ix.lists = nullptr;               // (interpreted tail at P = 4096: no room in the 160 KB -> dense rows only)
    }
    // (ordered fallbacks use the uniform-lists route only: no histogram)
  } else {
    lds = (size_t)sn.J * 64u * 12u + (size_t)wpb * c->pwn * 8u + (size_t)wpb * 64u * (size_t)c->lw_bytes;   // ... | 64 lane words per wavefront (set_from_list)
  }
  // occupancy-sized persistent grid (cached per kernel/LDS size: these are host calls on the launch path)
  int occ_per_cu = 1;
  { const int rco = occupancy_of(c, fn, threads, lds, &occ_per_cu); if (rco) return rco; }
  uint32_t grid = (n_reqs + wpb - 1) / wpb;
  const uint32_t cap = (uint32_t)c->num_cu * (uint32_t)occ_per_cu;
  if (grid > cap) grid = cap;
  if (grid > kStatSlots / wpb) grid = kStatSlots / wpb;
  if (grid < 1) grid = 1;

  const bool prof_now = c->prof && (c->prof_tick++ % c->prof_every) == 0u;     // (sampled profiling: eppk_profile_enable)
  unsigned long long* stats = prof_now ? c->stats + (size_t)(c->stat_bank++ % kStatBanks) * 2u * kStatSlots : nullptr;
  // pick_quad_kernel first (four requests per wavefront: the common shape of a request), then the fast kernel's work-list
  // instantiation over what it deferred.  Skipped for a while when a recent launch deferred a large part of its batch (a workload
  // of differing or overflowed lists: the quad pass is wasted on it); the pause doubles while that keeps happening.
  bool quad = fast && c->quad_on && c->has_p && c->npl == 6 && !c->gen &&
              c->pterm && ix.lists && ix.slots != 0u && c->cfg.max_blocks >= 1 && n_reqs >= c->quad_min;
  if (quad) quad_consume_reports(c);
  if (quad && c->quad_backoff) { --c->quad_backoff; quad = false; }
  eppk_ctx::DeferSet* dset = nullptr;
  if (quad) {                              // this stream's work-list buffer, a free report slot
    for (uint32_t i = 0; i < kDeferSets && !dset; ++i)
      if (c->dsets[i].used && c->dsets[i].st == st) dset = &c->dsets[i];
    for (uint32_t i = 0; i < kDeferSets && !dset; ++i)
      if (!c->dsets[i].used) { dset = &c->dsets[i]; dset->used = true; dset->st = st; }
    if (!dset || c->quad_launches - c->rep_unread >= kReportRing) quad = false;     // (a ninth stream; a full ring of launches without a report: the fast kernel)
  }
  const void* quad_fn = nullptr;
  uint32_t quad_grid = 0, defer_cap = 0, quad_segs = 0, rep_slot = 0;
  size_t quad_lds = 0;
  // ONE launch (pick_quad_kernel<TAIL>: every workgroup scores what its own wavefronts deferred, right behind its loop) for picks,
  // masked picks and ordered fallbacks alike; EPPK_QUAD_TAIL=0 keeps the two-launch form (pick_quad_kernel + a work-list pass).
  bool tail = false;
  if (quad) {
    const bool tkq = topk > 1;
    const uint32_t qwpb = c->quad_threads / 64u;
    tail = c->quad_tail_on;
    if (tail && d_learn && !tkq)
      quad_fn = masked ? eppk::pick_quad_tail_learn_masked(c->lw_bytes, c->has_l, c->p_first) : eppk::pick_quad_tail_learn(c->lw_bytes, c->has_l, c->p_first);
    else if (tail)
      quad_fn = c->lw_bytes == 2 ? eppk::pick_quad_tail_u16(c->has_l, c->p_first, masked, tkq) : c->lw_bytes == 4 ? eppk::pick_quad_tail_u32(c->has_l, c->p_first, masked, tkq)
                                                                                                               : eppk::pick_quad_tail_u64(c->has_l, c->p_first, masked, tkq);
    else
      quad_fn = c->lw_bytes == 2 ? eppk::pick_quad_u16(c->has_l, c->p_first, masked, tkq) : c->lw_bytes == 4 ? eppk::pick_quad_u32(c->has_l, c->p_first, masked, tkq)
                                                                                                            : eppk::pick_quad_u64(c->has_l, c->p_first, masked, tkq);
    // LDS: base[] | lw[4] | pterm | one "listed" bit per pod for each of the 4 rows of each wavefront
    //      (masked: + the snapshot's three natural-layout sets + the candidate words of each row)
    //      then the crossbar scratch of the set-id look-up: 512 bytes per wavefront (eppk_kernels.hip.h: quad_xbar_off)
    quad_lds = (size_t)eppk::quad_xbar_off(sn.J, pwn, qwpb, masked) + (size_t)qwpb * 512u;
    if (tail) {     // ... or the fast kernel's layout for a workgroup of this size, whichever is larger: base | lw | pterm | scratch | histogram
      const size_t fast_lds = (size_t)sn.J * 64u * 8u + 32u + (size_t)pwn * 8u + (size_t)qwpb * 64u * (size_t)c->lw_bytes + (size_t)qwpb * sn.J * 64u;
      if (fast_lds > quad_lds) quad_lds = fast_lds;
    }
    int quad_per_cu = 1;
    { const int rco = occupancy_of(c, quad_fn, c->quad_threads, quad_lds, &quad_per_cu); if (rco) return rco; }
    const uint32_t nblk = (n_reqs + 3u) / 4u;
    quad_grid = (nblk + qwpb - 1) / qwpb;
    const uint32_t qcap = (uint32_t)c->num_cu * (uint32_t)quad_per_cu;
    if (quad_grid > qcap) quad_grid = qcap;
    if (quad_grid > kStatSlots / qwpb) quad_grid = kStatSlots / qwpb;
    if (quad_grid < 1) quad_grid = 1;
    quad_segs = quad_grid * qwpb;
    defer_cap = 4u * ((nblk + quad_segs - 1) / quad_segs);
    // header: total[2] | done counters[17] (TAIL) | cnt | list | masked batches: the rows a wavefront scores itself when its loop is over
    // (pick_quad_body: my_xr / my_xs -- an index and 16 lanes x 8 bytes of state per request)
    const size_t words = 32u + (size_t)quad_segs + (size_t)quad_segs * defer_cap * (masked ? 34u : 1u);
    if (words > dset->words) {             // grow this stream's buffer (rare: its first launch, or a larger batch than ever before)
      { const int rcp = resident_park(c); if (rcp) return rcp; }      // (hipFree waits for the device)
      HIPCHK(c, hipStreamSynchronize(st));
      if (dset->d) HIPCHK(c, hipFree(dset->d));
      dset->d = nullptr; dset->words = 0;
      HIPCHK(c, hipMalloc((void**)&dset->d, words * 4u));
      HIPCHK(c, hipMemsetAsync(dset->d, 0, 128, st)); // the two total counters and the done counters -- on the LAUNCH stream: a null-stream memset is not ordered
                                                       // ahead of kernels on a non-blocking stream (the first launch of a stream could read a
                                                       // garbage total: found when the work-list pass got faster, tests/test_gpu_quad.py)
      dset->words = words; dset->uses = 0;
    }
    rep_slot = (uint32_t)(c->quad_launches % kReportRing);
    c->h_reports[rep_slot] = kReportPending;
    c->rep_n[rep_slot] = n_reqs;
    c->rep_masked[rep_slot] = masked ? 1 : 0;
    ++c->quad_launches;
  }
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (prof_now) {
    if (c->ev_used + 2 > c->ev.size()) {
      hipEvent_t a, b;
      HIPCHK(c, hipEventCreate(&a));
      HIPCHK(c, hipEventCreate(&b));
      c->ev.push_back(a);
      c->ev.push_back(b);
    }
    e0 = c->ev[c->ev_used];
    e1 = c->ev[c->ev_used + 1];
    c->ev_used += 2;
  }
  // While profiling, the start/stop events ride on the kernel's own dispatch packet (hipExtLaunchKernel): its begin/end
  // timestamps, no extra marker packets between back-to-back launches.
  const uint8_t* reqs8 = (const uint8_t*)d_reqs;
  uint32_t stride = c->stride;
  if (fast) {
    KTail tl = c->tail;
    KChain chf = c->kchain;
    eppk::KWork wk{};
    if (quad) {
      // two alternating total counters per buffer: a launch adds to one and zeroes the other for the buffer's next launch
      uint32_t* d_total = dset->d + (dset->uses & 1u);
      uint32_t* d_total_next = dset->d + ((dset->uses + 1u) & 1u);
      ++dset->uses;
      uint32_t* d_cnt = dset->d + 32;
      uint32_t* d_list = dset->d + 32 + quad_segs;
      uint32_t* d_done = dset->d + 2;       // TAIL: workgroups that have reported in (back to zero when the launch ends)
      uint32_t* h_rep = (uint32_t*)&c->h_reports[rep_slot];
      uint32_t* learn_arg = (tail && topk == 1) ? d_learn : nullptr;
      void* qargs[] = {&sn, &ix, &tl, &reqs8, &stride, &n_reqs, &pwn, &d_mask, &d_pick, &d_score, &stats, &d_cnt, &d_list, &defer_cap, &d_total, &d_total_next,
                       &topk, &d_done, &h_rep, &chf, &learn_arg};
      if (tail) {                           // one launch: every workgroup is its own work-list pass
        HIPCHK(c, hipExtLaunchKernel(quad_fn, dim3(quad_grid), dim3(c->quad_threads), qargs, quad_lds, st, e0, e1, 0));
        if (wrote_learn) *wrote_learn = learn_arg != nullptr;
        ++c->quad_tail_launches;
        c->last_done = e1;
        c->last_stream = st;
        if (prof_now) {
          c->fixed_bytes += (uint64_t)c->n_pods * sizeof(eppk_pod_row) + (uint64_t)n_reqs * ((uint64_t)c->stride + 4u);
          c->launches++;
        }
        return EPPK_OK;
      }
      HIPCHK(c, hipExtLaunchKernel(quad_fn, dim3(quad_grid), dim3(c->quad_threads), qargs, quad_lds, st, e0, nullptr, 0));
      e0 = nullptr;                         // (the pair is timed from the quad kernel's start to the work-list kernel's end)
      wk.cnt = d_cnt; wk.list = d_list; wk.total = d_total; wk.report = (uint32_t*)&c->h_reports[rep_slot]; wk.cap = defer_cap; wk.n_segs = quad_segs;
      // the work-list instantiation of the same fast kernel (same LDS, same geometry)
      const bool big = c->slots != 0 && c->index_bytes >= (1ull << 32);
      if (topk > 1 && masked)
        fn = c->lw_bytes == 2 ? eppk::pick_fast_wl_topk_masked_u16(c->has_l, c->p_first) : c->lw_bytes == 4 ? eppk::pick_fast_wl_topk_masked_u32(c->has_l, c->p_first)
                                                                                                            : eppk::pick_fast_wl_topk_masked_u64(c->has_l, c->p_first);
      else if (topk > 1)
        fn = c->lw_bytes == 2 ? eppk::pick_fast_wl_topk_u16(c->has_l, c->p_first, big) : c->lw_bytes == 4 ? eppk::pick_fast_wl_topk_u32(c->has_l, c->p_first, big)
                                                                                                          : eppk::pick_fast_wl_topk_u64(c->has_l, c->p_first, big);
      else
        fn = c->lw_bytes == 2 ? eppk::pick_fast_wl_u16(c->has_l, c->p_first, big, masked) : c->lw_bytes == 4 ? eppk::pick_fast_wl_u32(c->has_l, c->p_first, big, masked)
                                                                                                             : eppk::pick_fast_wl_u64(c->has_l, c->p_first, big, masked);
      int wl_per_cu = 1;
      { const int rco = occupancy_of(c, fn, threads, lds, &wl_per_cu); if (rco) return rco; }
      grid = (uint32_t)c->num_cu * (uint32_t)wl_per_cu;
      if (grid > kStatSlots / wpb) grid = kStatSlots / wpb;
      if (grid * wpb > quad_segs) grid = (quad_segs + wpb - 1) / wpb;      // one wavefront per segment at most
      // The pass is grid-stride over the segments, so ANY grid is correct; its size only has to fit what the quad kernel deferred.  A
      // full grid of 1024-thread workgroups (~100 KB of LDS each) that all find an empty list took 8.5 us per launch
      // (profiles/r02_kernel_stats.csv) -- dispatch cost, next to the other stream's quad kernel.  So: 16 workgroups (256 wavefronts:
      // up to a few thousand deferred requests at a handful per wavefront) unless the recent launches' reports ask for more.  The host
      // cannot wait for THIS launch's count, and a driver that enqueues a whole run ahead sees no report at all: the default has to be
      // the cheap one.  A workload that defers a large part of its batches pauses the quad route altogether (quad_consume_reports).
      {
        uint32_t want = 16u;
        if (c->wl_hint != 0xFFFFFFFFu && (2u * c->wl_hint + wpb - 1u) / wpb > want) want = (2u * c->wl_hint + wpb - 1u) / wpb;
        if (grid > want) grid = want;
      }
      if (grid < 1) grid = 1;
    }
    void* args[] = {&sn, &ix, &tl, &reqs8, &stride, &n_reqs, &pwn, &d_mask, &chf, &d_pick, &d_score, &stats, &topk, &wk};
    HIPCHK(c, hipExtLaunchKernel(fn, dim3(grid), dim3(threads), args, lds, st, e0, e1, 0));
  } else {
    KChain ch = c->kchain;
    void* args[] = {&sn, &ix, &ch, &reqs8, &stride, &n_reqs, &pwn, &d_mask, &d_pick, &d_score, &stats, &topk};
    HIPCHK(c, hipExtLaunchKernel(fn, dim3(grid), dim3(threads), args, lds, st, e0, e1, 0));
  }
  c->last_done = e1;
  c->last_stream = st;
  if (prof_now) {
    c->fixed_bytes += (uint64_t)c->n_pods * sizeof(eppk_pod_row) + (uint64_t)n_reqs * ((uint64_t)c->stride + 4u);
    c->launches++;
  }
  return EPPK_OK;
}

int rebuild_snapshot(eppk_ctx* c, uint32_t n_pods, hipStream_t st);

int validate_rows(eppk_ctx* c, const char* who, const void* reqs, uint32_t n_reqs, uint32_t first_row = 0);

// ---- the resident small-batch kernels (EPPK_RESIDENT=1) ------------------------------------------------------------------------------
// Units (eppk_ctx::res[]): 0 = pick_fast_kernel's body (plain picks below EPPK_RESIDENT_QUAD_FROM requests), 1 = pick_quad_kernel's body
// (plain picks), 2 = ... with candidate masks, 3 = ... with ordered fallbacks, 4 = both, 5 / 6 = single picks (plain / masked) followed by
// the post-route index update, applied by the resident workgroup itself (EPPK_PICK_LEARN).  Each is a kernel of its own behind a
// doorbell of its own, started by the first batch that needs it; an idle one leaves by itself.
constexpr uint32_t kResFast = 0u, kResQuad = 1u, kResMasked = 2u, kResTopk = 3u, kResTopkMasked = 4u, kResLearn = 5u, kResLearnMasked = 6u;
// LDS of a resident workgroup (sized for max_pods: the kernels outlive publishes; ONE size for every unit -- the argument block
// carries it): pick_fast_kernel's layout for 16 wavefronts, or pick_quad_kernel<MASKED>'s where that is larger.  *hist_fits = the
// per-wave pod histogram of the list routes fits as well (else the kernel is handed an index without list routes).
size_t resident_lds(const eppk_ctx* c, bool* hist_fits) {
  const uint32_t wpb = 16u, J = (c->cfg.max_pods + 63u) / 64u;
  const uint32_t pwn = (c->cfg.max_blocks + 1u) * c->pterm_ld;
  const size_t front = (size_t)J * 64u * 8u + 32u + (size_t)pwn * 8u;                                       // base | lw | pterm
  size_t lds = front + (size_t)wpb * 64u * (size_t)c->lw_bytes;                                             // | scratch
  const size_t hist = (size_t)wpb * J * 64u;
  *hist_fits = lds + hist <= c->max_lds;
  if (*hist_fits) lds += hist;
  const size_t quad_masked = front + (size_t)wpb * 4u * J * 8u + 192u * 8u + (size_t)wpb * 4u * J * 8u;     // | listed bits | natural sets | candidate words
  if (*hist_fits && quad_masked > lds && quad_masked <= c->max_lds) lds = quad_masked;
  // the quad body's crossbar scratch (512 bytes per wavefront) at the END of the allocation: behind everything either body keeps all-zero
  lds = (lds + 15u) & ~(size_t)15u;
  if (*hist_fits && lds + (size_t)wpb * 512u <= c->max_lds) lds += (size_t)wpb * 512u;
  else *hist_fits = false;                      // (no room: the resident units run pick_fast_kernel's body alone)
  return lds;
}
// Park them: ring "quit" and wait for the workgroups to leave.  In front of every device-wide wait of the library's own (a
// hipDeviceSynchronize would otherwise sit out the kernels' idle timeout), and in eppk_destroy.
static const bool g_res_dbg = getenv("EPPK_RESIDENT_DEBUG") != nullptr && atoi(getenv("EPPK_RESIDENT_DEBUG")) != 2;     // (read once: the macro sits on the latency path)
// EPPK_RESIDENT_DEBUG=2: nothing is printed on the latency path (a write(2) between two doorbells changes what is measured); the waits and the
// device stamps of the last doorbells are kept and printed by eppk_destroy
static const bool g_res_log = getenv("EPPK_RESIDENT_DEBUG") != nullptr && atoi(getenv("EPPK_RESIDENT_DEBUG")) == 2;
struct ResLog { uint32_t unit, seq; float wait_us; uint32_t st[4]; };
static std::vector<ResLog> g_res_logs;
#define RES_DBG(...) do { if (g_res_dbg) { std::fprintf(stderr, "[eppk resident] " __VA_ARGS__); std::fprintf(stderr, "\n"); std::fflush(stderr); } } while (0)
int resident_drain(eppk_ctx* c);
int resident_park(eppk_ctx* c) {
  { const int rcd = resident_drain(c); if (rcd) return rcd; }       // (a batch rung by eppk_pick_stage_begin and not yet collected is answered first)
  for (eppk_ctx::ResidentUnit& u : c->res) {                // (every doorbell first, then the waits)
    if (!u.running) continue;
    RES_DBG("park: bell %u done %u state %u", u.h_ctl->bell, u.h_ctl->done, u.h_ctl->state);
    __atomic_store_n(&u.h_ctl->bell, eppk::kResQuit, __ATOMIC_RELEASE);
  }
  for (eppk_ctx::ResidentUnit& u : c->res) {
    if (!u.running) continue;
    HIPCHK(c, hipStreamSynchronize(u.stream));
    RES_DBG("parked: state %u", u.h_ctl->state);
    u.running = false;
    if (u.slot >= 0) { c->res_slot_unit[u.slot] = -1; u.slot = -1; }
  }
  return EPPK_OK;
}
// ONE unit leaves (its slot goes to another): answered and updated first, then "quit".
int resident_park_one(eppk_ctx* c, uint32_t unit) {
  eppk_ctx::ResidentUnit& u = c->res[unit];
  if (!u.running) {          // (a unit whose restart failed keeps no slot: two units on one stream is the 38 ms queue blocking the slots exist to avoid)
    if (u.slot >= 0) { c->res_slot_unit[u.slot] = -1; u.slot = -1; }
    return EPPK_OK;
  }
  { const int rcd = resident_drain(c); if (rcd) return rcd; }
  __atomic_store_n(&u.h_ctl->bell, eppk::kResQuit, __ATOMIC_RELEASE);
  HIPCHK(c, hipStreamSynchronize(u.stream));
  u.running = false;
  if (u.slot >= 0) { c->res_slot_unit[u.slot] = -1; u.slot = -1; }
  return EPPK_OK;
}
int device_sync(eppk_ctx* c) {
  const int rc = resident_park(c);
  if (rc) return rc;
  HIPCHK(c, hipDeviceSynchronize());
  return EPPK_OK;
}
// The form of the resident kernels: pick_quad_kernel's body (four requests per wavefront) wherever a launch would take that route --
// the list routes are on and their LDS fits beside the tables -- else pick_fast_kernel's body alone (plain picks only).  Fixed for the
// life of a context.
bool resident_quad(const eppk_ctx* c) {
  bool hist_fits = false;
  (void)resident_lds(c, &hist_fits);
  return c->quad_on && hist_fits && c->slots != 0u && make_kindex(c).lists != nullptr;
}
// k = entries per request (1: the pick); learn: the post-route index update chained behind the pick (single picks)
bool resident_eligible(const eppk_ctx* c, uint32_t n_reqs, bool masked, uint32_t k = 1u, bool learn = false) {
  // (n_reqs <= 255 for EVERY shape: the doorbell's upper half carries n in eight bits -- res_bell_hi)
  if (!(c->resident_on && n_reqs != 0 && n_reqs <= c->resident_max && n_reqs <= 255u && c->canonical && c->has_p && c->npl == 6 && !c->gen && c->pterm &&
        c->assumed_epochs == 0 && c->cfg.max_blocks >= 1)) return false;
  if (!masked && k == 1u && !learn) return true;
  if (learn && (k != 1u || !c->slots || !c->have_snapshot)) return false;
  return k <= EPPK_MAX_TOPK && resident_quad(c);     // the variants exist for the quad form
}
uint32_t resident_unit_of(const eppk_ctx* c, uint32_t n_reqs, bool masked, uint32_t k, bool learn);
bool resident_admit(eppk_ctx* c, uint32_t unit);
// what the entry points ask: may this batch take the latency path NOW?
bool resident_takes(eppk_ctx* c, uint32_t n_reqs, bool masked, uint32_t k = 1u, bool learn = false) {
  return resident_eligible(c, n_reqs, masked, k, learn) && resident_admit(c, resident_unit_of(c, n_reqs, masked, k, learn));
}
int resident_ensure(eppk_ctx* c) {            // control blocks, argument block, streams (each made once: a call that failed half-way is resumed)
  if (c->d_res_args) return EPPK_OK;
  if (!c->d_res_wl) {   // the work list of the quad forms: 16 wavefronts, each with room for every request it can meet (4 per block, its share of the blocks)
    const uint32_t nblk = (c->resident_max + 3u) / 4u, per_wave = (nblk + 15u) / 16u;
    c->res_wl_cap = 4u * (per_wave ? per_wave : 1u);
    const size_t words = 32u + 16u + 16u * (size_t)c->res_wl_cap * 34u;       // (... | masked single picks: the rows a wavefront parks, pick_quad_body my_xr / my_xs)
    uint32_t* wl = nullptr;
    HIPCHK(c, hipMalloc((void**)&wl, words * 4u * eppk_ctx::kResUnits));      // (one work list per unit: two units may be scoring at once)
    if (hipMemset(wl, 0, words * 4u * eppk_ctx::kResUnits) != hipSuccess) { (void)hipFree(wl); return fail(c, EPPK_ERR_DEVICE, "resident path: hipMemset of the work lists failed"); }
    c->d_res_wl = wl;
  }
  for (eppk_ctx::ResidentUnit& u : c->res) {
    if (!u.h_ctl) {
      eppk::ResidentCtl* h = nullptr;
      HIPCHK(c, hipHostMalloc((void**)&h, sizeof(eppk::ResidentCtl), hipHostMallocDefault));
      std::memset(h, 0, sizeof(eppk::ResidentCtl));
      u.h_ctl = h;
    }
    if (!u.h_ctl_dev) HIPCHK(c, hipHostGetDevicePointer((void**)&u.h_ctl_dev, u.h_ctl, 0));
  }
  // The streams the resident kernels are launched on: kResSlots of them, at the HIGHEST stream priority.  The runtime multiplexes the
  // streams of one priority class over a few hardware queues (four by default), and a kernel that never ends blocks its queue: a second
  // resident kernel -- or an ordinary launch of this context -- that lands on the same hardware queue starts when the first one
  // leaves, i.e. after its idle time-out (measured: 38 ms instead of 12 us, round 5).  Streams of another priority class come from a
  // pool of their own: the first kResSlots high-priority streams each get a hardware queue to themselves, away from every launched kernel.
  // More units than slots may never be alive at once (resident_start parks the least recently rung one).
  for (hipStream_t& st : c->res_slot_stream) {
    if (st) continue;
    int lo = 0, hi = 0;
    if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess || hipStreamCreateWithPriority(&st, hipStreamNonBlocking, hi) != hipSuccess) {
      (void)hipGetLastError();
      st = nullptr;
      HIPCHK(c, hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    }
  }
  if (c->slots && !c->d_res_sortwl) {       // LEARN units (two at most): a device copy of a batch's rows, its learn words, a sort work list each
    const size_t rm = c->resident_max ? c->resident_max : 1u;
    HIPCHK(c, hipMalloc((void**)&c->d_res_rows, 2u * rm * c->stride));
    HIPCHK(c, hipMalloc((void**)&c->d_res_learn, 2u * rm * 4u));
    const size_t wl_words = 4u + rm * (c->cfg.max_blocks ? c->cfg.max_blocks : 1u);
    uint32_t* wl = nullptr;
    HIPCHK(c, hipMalloc((void**)&wl, 2u * wl_words * 4u));
    if (hipMemset(wl, 0, 2u * wl_words * 4u) != hipSuccess) { (void)hipFree(wl); return fail(c, EPPK_ERR_DEVICE, "resident path: hipMemset of the sort work lists failed"); }
    c->d_res_sortwl = wl;
  }
  HIPCHK(c, hipMalloc((void**)&c->d_res_args, sizeof(eppk::ResidentArgs) * eppk_ctx::kResUnits));
  return EPPK_OK;
}
// `outstanding`: a doorbell has been rung that the workgroup which just left did not answer (the in-call restart of resident_wait): the
// new workgroup must take it at once (seen = seq - 1).  Otherwise the last doorbell HAS been answered: seen = seq, or the fresh
// workgroup would score that batch a second time -- the stale count of the doorbell word against whatever rows the caller is writing
// into the staging buffer for its next call (~10 us on the path meant to save them, results written behind the new batch's end).
int resident_start(eppk_ctx* c, uint32_t unit, bool outstanding = false) {
  eppk_ctx::ResidentUnit& u = c->res[unit];
  if (u.running) return EPPK_OK;
  { const int rce = resident_ensure(c); if (rce) return rce; }
  if (u.slot < 0) {                 // a stream slot: a free one (a unit that left by itself still holds its slot: taken back first), else the least recently rung unit's
    for (uint32_t sl = 0; sl < eppk_ctx::kResSlots; ++sl) {
      const int32_t other = c->res_slot_unit[sl];
      if (other >= 0 && c->res[other].running && !c->res[other].pending && __atomic_load_n(&c->res[other].h_ctl->state, __ATOMIC_ACQUIRE) == eppk::kResExited) {
        HIPCHK(c, hipStreamSynchronize(c->res[other].stream));
        c->res[other].running = false; c->res[other].slot = -1; c->res_slot_unit[sl] = -1;
      }
    }
    int32_t pick = -1;
    for (uint32_t sl = 0; sl < eppk_ctx::kResSlots && pick < 0; ++sl) if (c->res_slot_unit[sl] < 0) pick = (int32_t)sl;
    if (pick < 0) {
      uint32_t lru = 0; uint64_t best = ~0ull;
      for (uint32_t sl = 0; sl < eppk_ctx::kResSlots; ++sl) { const uint64_t t = c->res[c->res_slot_unit[sl]].last_rung; if (t < best) { best = t; lru = sl; } }
      const int rcp = resident_park_one(c, (uint32_t)c->res_slot_unit[lru]);
      if (rcp) return rcp;
      pick = (int32_t)lru;
    }
    u.slot = pick; c->res_slot_unit[pick] = (int32_t)unit; u.stream = c->res_slot_stream[pick];
  }
  const void* fn = unit == kResFast ? eppk::pick_resident(c->lw_bytes, c->has_l, c->p_first)
                 : unit == kResQuad ? eppk::pick_resident_quad(c->lw_bytes, c->has_l, c->p_first)
                 : unit >= kResLearn ? eppk::pick_resident_quad_learn(c->lw_bytes, c->has_l, c->p_first, unit == kResLearnMasked)
                                    : eppk::pick_resident_quad_variant(c->lw_bytes, c->has_l, c->p_first, unit == kResMasked || unit == kResTopkMasked, unit == kResTopk || unit == kResTopkMasked);
  const uint32_t threads = 1024u;
  bool hist_fits = false;
  const size_t lds = resident_lds(c, &hist_fits);
  int per_cu = 0;
  { const int rco = occupancy_of(c, fn, threads, lds, &per_cu); if (rco) return rco; }
  const uint32_t seen = (outstanding && u.seq != 0u) ? u.seq - 1u : u.seq;
  uint32_t bell_now = __atomic_load_n(&u.h_ctl->bell, __ATOMIC_ACQUIRE);
  if (bell_now == eppk::kResQuit) __atomic_store_n(&u.h_ctl->bell, seen, __ATOMIC_RELEASE);
  __atomic_store_n(&u.h_ctl->state, eppk::kResRunning, __ATOMIC_RELEASE);
  eppk::ResidentCtl* ctl = u.h_ctl_dev;
  const eppk::ResidentArgs* args = c->d_res_args + unit;
  uint32_t seen_arg = seen;
  unsigned long long max_idle = 30000ull;                 // ~20-50 ms of polls over PCIe, then the workgroup leaves by itself
  if (const char* e = getenv("EPPK_RESIDENT_IDLE_POLLS")) { const long long v = atoll(e); if (v > 0) max_idle = (unsigned long long)v; }
  void* kargs[] = {&ctl, &args, &seen_arg, &max_idle};
  RES_DBG("start unit %u: seen %u bell %u lds %zu", unit, seen_arg, u.h_ctl->bell, lds);
  HIPCHK(c, hipExtLaunchKernel(fn, dim3(1), dim3(threads), kargs, lds, u.stream, nullptr, nullptr, 0));
  u.running = true;
  ++c->res_starts;
  return EPPK_OK;
}
// Wait for the answer to doorbell `seq` of `unit` (restarting a workgroup that left just as the doorbell rang).
int resident_wait(eppk_ctx* c, uint32_t unit, uint32_t seq, const char* who) {
  eppk_ctx::ResidentUnit& u = c->res[unit];
  const auto t0 = std::chrono::steady_clock::now();
  uint32_t spins = 0;
  // (doorbells of a unit are answered in order: `done` may already be past `seq` when a later batch of the same unit was rung and
  // collected first -- sequence numbers compare modulo 2^32)
  auto reached = [&]() { return (int32_t)(__atomic_load_n(&u.h_ctl->done, __ATOMIC_ACQUIRE) - seq) >= 0; };
  while (!reached()) {
    if ((++spins & 1023u) == 0u) {
      if (__atomic_load_n(&u.h_ctl->state, __ATOMIC_ACQUIRE) == eppk::kResExited && !reached()) {
        // the workgroup left (idle timeout) just as the doorbell rang: start it again; it sees this doorbell at once
        HIPCHK(c, hipStreamSynchronize(u.stream));
        u.running = false;
        const int rc = resident_start(c, unit, true);
        if (rc) return rc;
      }
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 5.0)
        return fail(c, EPPK_ERR_DEVICE, std::string(who) + ": the resident pick kernel did not answer within 5 s");
    }
  }
  if (u.pending && u.pending_seq == seq) u.pending = false;
  if (g_res_log && g_res_logs.size() < 65536u)
    g_res_logs.push_back({unit, seq, (float)(1e6 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count()), {u.h_ctl->pad1[3], u.h_ctl->pad1[0], u.h_ctl->pad1[1], u.h_ctl->pad1[2]}});
  RES_DBG("answered %u (unit %u) after %u spins; wait %.2f us; device stamps (10 ns ticks, measurement builds only): rows copied %u; bell seen -> caches invalidated %u, -> body done %u, -> released %u",
          seq, unit, spins, 1e6 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), u.h_ctl->pad1[3], u.h_ctl->pad1[0], u.h_ctl->pad1[1], u.h_ctl->pad1[2]);
  return EPPK_OK;
}
// Every doorbell that has been rung and not collected yet (eppk_pick_stage_begin rings, _end collects) is answered: in front of
// whatever CHANGES what the resident workgroups read -- the argument blocks, the index, the snapshot -- and of parking them.
int resident_drain(eppk_ctx* c) {
  for (uint32_t unit = 0; unit < eppk_ctx::kResUnits; ++unit) {
    eppk_ctx::ResidentUnit& u = c->res[unit];
    if (u.pending) {
      const int rc = resident_wait(c, unit, u.pending_seq, "resident path");
      u.pending = false;
      if (rc) return rc;
    }
    if (u.updating) {               // a LEARN unit: the index update behind its last answer (microseconds; the workgroup cannot leave in between)
      const auto t0 = std::chrono::steady_clock::now();
      uint32_t spins = 0;
      while ((int32_t)(__atomic_load_n(&u.h_ctl->updated, __ATOMIC_ACQUIRE) - u.update_seq) < 0) {
        if ((++spins & 4095u) == 0u && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 5.0)
          return fail(c, EPPK_ERR_DEVICE, "resident path: the index update of a small LEARN batch did not finish within 5 s");
      }
      u.updating = false;
    }
  }
  return EPPK_OK;
}
// The unit a call shape rings.
uint32_t resident_unit_of(const eppk_ctx* c, uint32_t n_reqs, bool masked, uint32_t k, bool learn) {
  if (learn) return masked ? kResLearnMasked : kResLearn;
  if (!masked && k == 1u) return (n_reqs >= c->resident_quad_from && resident_quad(c)) ? kResQuad : kResFast;   // (EPPK_RESIDENT_QUAD_FROM; measured crossover: see eppk_ctx)
  return k > 1u ? (masked ? kResTopkMasked : kResTopk) : kResMasked;
}
// More call shapes than stream slots: a shape whose unit is not resident while every slot is taken by units in use takes the LAUNCHED path
// (false) until it has been asked for kResAdmitAfter times in a row -- only then does the least recently rung unit make room.  (Five
// shapes in rotation over four slots parked and started a unit per call: ~35 us each, worse than a launch; profiles/r05_resident_latency.txt.)
constexpr uint32_t kResAdmitAfter = 8u;
bool resident_admit(eppk_ctx* c, uint32_t unit) {
  eppk_ctx::ResidentUnit& u = c->res[unit];
  if (u.running || u.slot >= 0) { u.misses = 0u; return true; }
  for (uint32_t sl = 0; sl < eppk_ctx::kResSlots; ++sl) {
    const int32_t other = c->res_slot_unit[sl];
    if (other < 0) { u.misses = 0u; return true; }                                  // a free slot
    const eppk_ctx::ResidentUnit& o = c->res[other];
    if (o.h_ctl && !o.pending && __atomic_load_n(&o.h_ctl->state, __ATOMIC_ACQUIRE) == eppk::kResExited) { u.misses = 0u; return true; }   // ... or one whose unit has left by itself
  }
  if (++u.misses >= kResAdmitAfter) { u.misses = 0u; return true; }                 // asked for often enough: the LRU unit goes
  return false;
}
// Ring a small batch in: its rows (and mask rows) are in buffer set `bufset` (0 = the context's pinned staging buffers, 1 + s = staging
// set s) already and have been validated; the results land in that set's pinned result buffers.  *unit_out / *seq_out: what to wait for.
int resident_ring(eppk_ctx* c, uint32_t n_reqs, bool masked, uint32_t k, uint32_t bufset, uint32_t* unit_out, uint32_t* seq_out, bool learn = false) {
  { const int rcf = learn_fence(c, c->stream); if (rcf) return rcf; }
  // The resident kernels are OUTSIDE stream order: whatever this context has queued that changes the index or the snapshot must be over
  // before the doorbell rings -- a LEARN update behind a staging set (the `learned` event), and anything on the context's own stream
  // (eppk_index_insert_picks_device / eppk_pick_learn_device / eppk_index_evict_older_device with stream = NULL on a context that never
  // used the staging sets: no event).  One hipStreamQuery when the stream is idle.  Work on a CALLER's stream is the caller's to order
  // (include/eppk.h: "streams").
  if (c->learn_pending || hipStreamQuery(c->stream) != hipSuccess) HIPCHK(c, hipStreamSynchronize(c->stream));
  (void)hipGetLastError();
  int rc = resident_ensure(c);
  if (rc) return rc;
  if (c->res_args_dirty) {        // a publish, new staging buffers (or the first use): the argument blocks again -- with no doorbell outstanding
    rc = resident_drain(c);
    if (rc) return rc;
    eppk::ResidentArgs a{};
    a.sn = make_ksnap(c); a.ix = make_kindex(c); a.tl = c->tail; a.chain = c->kchain;
    a.buf[0] = {(const uint8_t*)c->h_reqs_dev, c->h_mask_dev, c->h_pick_dev, c->h_score_dev};
    for (uint32_t sset = 0; sset < EPPK_STAGE_SETS; ++sset) {
      const eppk_ctx::StageSet& ss = c->stage[sset];
      a.buf[1u + sset] = {(const uint8_t*)ss.h_reqs_dev, ss.h_mask_dev, ss.h_pick_dev, ss.h_score_dev};
    }
    a.stride = c->stride; a.pwn = (c->cfg.max_blocks + 1u) * c->pterm_ld;
    if (++c->res_gen == 0u) c->res_gen = 1u;
    a.gen = c->res_gen;
    bool hist_fits = false;
    a.lds_bytes = (uint32_t)resident_lds(c, &hist_fits);
    if (!hist_fits) a.ix.lists = nullptr;                 // (no room for the list routes' histogram in the 160 KB: dense rows only)
    if (c->slots) {               // what the LEARN units' own index update needs (eppk.hip: learn_picks passes the same to its launches)
      a.keys_w = c->keys; a.bitmaps_w = c->bitmaps; a.lists_w = c->lists; a.rstamps = c->rstamps; a.ixc = c->ixc; a.status = c->d_status + 1;   // (rows are validated on the host: the quiet word)
      a.act = c->have_snapshot ? c->snap[c->cur].act_t : nullptr;
      a.limit = c->limit; a.epoch = c->index_epoch; a.max_blocks = c->cfg.max_blocks; a.max_pods = c->cfg.max_pods;
      a.set_ctl = c->set_ctl;
    }
    eppk::ResidentArgs all[eppk_ctx::kResUnits];
    const size_t wl_words = 32u + 16u + 16u * (size_t)c->res_wl_cap * 34u;
    const size_t rm = c->resident_max ? c->resident_max : 1u, sort_words = 4u + rm * (c->cfg.max_blocks ? c->cfg.max_blocks : 1u);
    for (uint32_t unit = 0; unit < eppk_ctx::kResUnits; ++unit) {
      all[unit] = a;
      uint32_t* wl = c->d_res_wl + unit * wl_words;
      all[unit].defer_total = wl; all[unit].defer_cnt = wl + 32; all[unit].defer_list = wl + 48; all[unit].defer_cap = c->res_wl_cap;
      if (unit >= kResLearn && c->d_res_sortwl) {
        const uint32_t lu = unit - kResLearn;
        all[unit].rows_copy = c->d_res_rows + lu * rm * c->stride; all[unit].learn = c->d_res_learn + lu * rm;
        all[unit].sort_wl = c->d_res_sortwl + lu * sort_words; all[unit].sort_cap = (uint32_t)(sort_words - 4u);
      }
    }
    HIPCHK(c, hipMemcpy(c->d_res_args, all, sizeof all, hipMemcpyHostToDevice));
    c->res_args_dirty = false;
  }
  const uint32_t unit = resident_unit_of(c, n_reqs, masked, k, learn);
  eppk_ctx::ResidentUnit& u = c->res[unit];
  if (u.pending) {                  // one doorbell per unit at a time (the other staging set's batch of the same kind): answered first
    rc = resident_wait(c, unit, u.pending_seq, "resident path");
    if (rc) return rc;
  }
  rc = resident_start(c, unit);
  if (rc) return rc;
  if (++u.seq == eppk::kResQuit || u.seq == 0u) u.seq = 1u;
  RES_DBG("ring %u (unit %u, n = %u, k = %u, buffers %u)", u.seq, unit, n_reqs, k, bufset);
  static_assert(offsetof(eppk::ResidentCtl, n_reqs) == offsetof(eppk::ResidentCtl, bell) + 4u && offsetof(eppk::ResidentCtl, bell) % 8u == 0u, "doorbell + count: one aligned 8-byte word");
  __atomic_store_n((uint64_t*)&u.h_ctl->bell, ((uint64_t)eppk::res_bell_hi(n_reqs, k, bufset) << 32) | u.seq, __ATOMIC_RELEASE);      // count and doorbell in one store
  u.pending = true; u.pending_seq = u.seq; u.last_rung = ++c->res_clock;
  if (learn) { u.updating = true; u.update_seq = u.seq; }
  *unit_out = unit; *seq_out = u.seq;
  ++c->res_batches;
  return EPPK_OK;
}
// One small batch through a resident kernel, synchronously: rows (and mask) are in c->h_reqs / c->h_mask (pinned) already; results land in
// c->h_pick / c->h_score (n * k entries with ordered fallbacks).
int resident_pick(eppk_ctx* c, uint32_t n_reqs, int32_t* out_pick, double* out_score, const char* who, bool masked = false, uint32_t k = 1u) {
  int rc = validate_rows(c, who, c->h_reqs, n_reqs, 0u);
  if (rc) return rc;
  uint32_t unit = 0, seq = 0;
  rc = resident_ring(c, n_reqs, masked, k, 0u, &unit, &seq);
  if (rc) return rc;
  rc = resident_wait(c, unit, seq, who);
  if (rc) return rc;
  std::memcpy(out_pick, c->h_pick, (size_t)n_reqs * k * 4u);
  if (out_score) std::memcpy(out_score, c->h_score, (size_t)n_reqs * k * 8u);
  return EPPK_OK;
}

// One batch through the picker the caller asked for, in assumed-load epochs when those are on (SEMANTICS.md §2b):
//   k == 1, !random   the pick                         (d_pick / d_score: n entries)
//   k  > 1, !random   ordered fallbacks                (n * k entries; the request's pick is entry 0 of its list)
//   random            picker "random-top-k" (§3b)      (n entries), r0 = batch index of the first request (the rule hashes it)
int run_pick(eppk_ctx* c, const uint8_t* d_reqs, uint32_t n_reqs, const uint64_t* d_mask, int32_t* d_pick, double* d_score, hipStream_t st,
             uint32_t k, bool random, uint64_t seed, uint32_t r0, uint32_t* d_learn = nullptr, bool* wrote_learn = nullptr) {
  bool all_wrote = d_learn != nullptr;       // (learn words: valid only if EVERY launch of the batch wrote its part)
  { const int rcf = learn_fence(c, st); if (rcf) return rcf; }     // (the index an earlier EPPK_PICK_LEARN batch leaves behind)
  const uint32_t E = c->assumed_epochs;
  const uint32_t per = E ? (n_reqs + E - 1u) / E : n_reqs;
  const size_t J = (c->n_pods + 63u) / 64u;
  const uint32_t ok = random ? 1u : k;                        // entries per request in the caller's arrays
  if (random && (size_t)per * k > c->rs_cap) {                // fallback lists of one epoch
    { const int rcp = resident_park(c); if (rcp) return rcp; }
    HIPCHK(c, hipStreamSynchronize(st));
    (void)hipFree(c->d_rs_pick); (void)hipFree(c->d_rs_score);
    c->d_rs_pick = nullptr; c->d_rs_score = nullptr; c->rs_cap = 0;
    HIPCHK(c, hipMalloc((void**)&c->d_rs_pick, (size_t)per * k * 4u));
    HIPCHK(c, hipMalloc((void**)&c->d_rs_score, (size_t)per * k * 8u));
    c->rs_cap = (size_t)per * k;
  }
  for (uint32_t lo = 0; lo < n_reqs; lo += per) {
    const uint32_t cnt = n_reqs - lo < per ? n_reqs - lo : per;
    const uint8_t* reqs = d_reqs + (size_t)lo * c->stride;
    const uint64_t* mask = d_mask ? d_mask + (size_t)lo * J : nullptr;
    int32_t* pick = d_pick + (size_t)lo * ok;
    double* score = d_score ? d_score + (size_t)lo * ok : nullptr;
    int rc;
    if (random) {
      rc = launch_pick(c, reqs, cnt, mask, c->d_rs_pick, c->d_rs_score, st, k);
      if (rc) return rc;
      hipLaunchKernelGGL(random_select_kernel, dim3((cnt + 255u) / 256u), dim3(256), 0, st, (const int32_t*)c->d_rs_pick, (const double*)c->d_rs_score,
                         cnt, k, seed, r0 + lo, pick, score);
      HIPCHK(c, hipGetLastError());
    } else {
      bool wrote = false;
      rc = launch_pick(c, reqs, cnt, mask, pick, score, st, k, d_learn ? d_learn + lo : nullptr, &wrote);
      if (rc) return rc;
      all_wrote = all_wrote && wrote;
    }
    if (E) {    // the assumed load of what this epoch routed, then everything derived from the queue gauge again
      hipLaunchKernelGGL(assumed_bump_kernel, dim3((cnt + 255u) / 256u), dim3(256), 0, st, c->d_rows, (const int32_t*)pick, cnt, ok, c->n_pods);
      HIPCHK(c, hipGetLastError());
      rc = rebuild_snapshot(c, c->n_pods, st);
      if (rc) return rc;
    }
  }
  if (wrote_learn) *wrote_learn = all_wrote && !random;
  return EPPK_OK;
}

template <typename F>
int by_lane_word(const eppk_ctx* c, F&& f) {
  switch (c->lw_bytes) {
    case 2: return f((uint16_t)0);
    case 4: return f((uint32_t)0);
    default: return f((uint64_t)0);
  }
}

// The work list of index_lists_sort_kernel (eppk_kernels.hip.h: SortWl): room for one entry per pair of the insert launch that is
// about to be issued (capped at 16 Mi entries: beyond that a launch that overflows it makes the sort pass walk the table), and the
// cursor this launch appends through.  Index updates of one context are ordered with respect to each other (one stream, or
// events: include/eppk.h), so one work list serves them all.
int sortwl_begin(eppk_ctx* c, uint64_t n_pairs, eppk::SortWl* sw) {
  const uint32_t want = (uint32_t)(n_pairs < (1ull << 24) ? n_pairs : (1ull << 24));
  if (!c->sortwl || want > c->sortwl_cap) {
    { const int rcs_ = device_sync(c); if (rcs_) return rcs_; }                 // (rare: the first insert, or a larger launch than ever before)
    if (c->sortwl) HIPCHK(c, hipFree(c->sortwl));
    c->sortwl = nullptr; c->sortwl_cap = 0;
    const uint32_t cap = want < 4096u ? 4096u : want;
    HIPCHK(c, hipMalloc((void**)&c->sortwl, (4u + (size_t)cap) * 4u));
    HIPCHK(c, hipMemset(c->sortwl, 0, 16));
    { const int rcs_ = device_sync(c); if (rcs_) return rcs_; }
    c->sortwl_cap = cap; c->sort_uses = 0;
  }
  sw->wl = c->sortwl; sw->cap = c->sortwl_cap; sw->which = c->sort_uses & 1u;
  ++c->sort_uses;
  return EPPK_OK;
}
eppk::SetTab make_settab(const eppk_ctx* c) {
  eppk::SetTab st{};
  st.lines = c->lists + ((size_t)c->slots + 4u) * eppk::kListDwords; st.mask = c->sets_cap - 1u; st.ctl = c->set_ctl;
  return st;
}
// ... and the canon pass behind the update launch, on its stream: the lists it changed go back to ascending order and get their set ids
// again (eppk_kernels.hip.h: index_canon_kernel).  The host does not know how many lists the launch touched: a grid of 64 workgroups
// walks whatever the cursor says (a closed-loop step of a 64k batch touches a few thousand lists at most).
// The set table fills with the lines of sets that no key names any more (a prefix whose pod set grew leaves its earlier sets behind, an
// evicted key its own): when the pass last reported it half full -- or a set that found no line since the last rebuild -- the table is
// cleared and the pass walks the whole index instead, interning every listed set again (`force_all`).  The report is a pinned word the
// pass before this one wrote: read without synchronising, late by one launch at most; a set without a line is only slower to pick from.
int sortwl_finish(eppk_ctx* c, const eppk::SortWl& sw, hipStream_t st) {
  const uint32_t used = __atomic_load_n(&c->h_set_report[0], __ATOMIC_RELAXED), failed = __atomic_load_n(&c->h_set_report[1], __ATOMIC_RELAXED);
  const bool rebuild = (used > c->sets_cap / 2u || failed != c->set_fail_seen) && ++c->set_passes >= 16u;
  if (rebuild) {
    c->set_passes = 0u;
    HIPCHK(c, hipMemsetAsync(c->lists + ((size_t)c->slots + 4u) * eppk::kListDwords, 0, (size_t)c->sets_cap * eppk::kListDwords * 4u, st));
    HIPCHK(c, hipMemsetAsync(c->set_ctl, 0, 4u, st));          // (lines in use; the failure count keeps counting)
    c->h_set_report[0] = 0u;
    c->set_fail_seen = failed;
  }
  const uint32_t grid = rebuild ? (uint32_t)std::min<size_t>(((size_t)c->slots + 2u + 255u) / 256u, 4096u) : 64u;
  hipLaunchKernelGGL(eppk::index_canon_kernel, dim3(grid), dim3(256), 0, st, c->keys, c->lists, c->slots, make_settab(c), sw.wl, sw.cap, sw.which, rebuild ? 1u : 0u,
                     c->h_set_report_dev);
  HIPCHK(c, hipGetLastError());
  return EPPK_OK;
}

// Remove every pod of the lane-transposed set `holes` from every index row (one pass; rows that become empty are tombstoned).
int index_scrub(eppk_ctx* c, const uint64_t* holes) {
  if (!c->d_rm) HIPCHK(c, hipMalloc(&c->d_rm, 64u * 8u));
  uint8_t packed[64 * 8];
  for (uint32_t l = 0; l < 64u; ++l) {           // narrow the u64 lane words to the context's lane-word type
    if (c->lw_bytes == 8) std::memcpy(packed + l * 8u, &holes[l], 8u);
    else if (c->lw_bytes == 4) { const uint32_t v = (uint32_t)holes[l]; std::memcpy(packed + l * 4u, &v, 4u); }
    else { const uint16_t v = (uint16_t)holes[l]; std::memcpy(packed + l * 2u, &v, 2u); }
  }
  HIPCHK(c, hipMemcpyAsync(c->d_rm, packed, 64u * (size_t)c->lw_bytes, hipMemcpyHostToDevice, c->stream));
  const uint32_t rows = c->slots + 2u, threads = 256;
  uint32_t grid = (rows * 64u + threads - 1) / threads;
  if (grid > 4096u) grid = 4096u;
  eppk::SortWl sw{};
  int rc = sortwl_begin(c, 1u << 16, &sw);        // (the sets the pass edits lose their ids: the canon pass behind it gives them new ones)
  if (rc) return rc;
  rc = by_lane_word(c, [&](auto tag) {
    using LW = decltype(tag);
    hipLaunchKernelGGL((index_remove_pod_kernel<LW>), dim3(grid), dim3(threads), 0, c->stream, c->keys, c->bitmaps, c->lists, c->slots, 0u, c->ixc,
                       (const LW*)c->d_rm, sw);
    return EPPK_OK;
  });
  HIPCHK(c, hipGetLastError());
  if (rc == EPPK_OK) rc = sortwl_finish(c, sw, c->stream);
  HIPCHK(c, hipStreamSynchronize(c->stream));    // (`packed` is a stack buffer)
  return rc;
}

// (declared ahead of run_pick)
// Build the idle snapshot buffer from the raw rows in c->d_rows (asynchronous on `st`) and make it the current one: the
// publish path, and the rebuild after every assumed-load epoch (SEMANTICS.md §2b) -- same kernels, same binary64 operations.
int rebuild_snapshot(eppk_ctx* c, uint32_t n_pods, hipStream_t st) {
  const uint32_t J = (n_pods + 63u) / 64u;
  const size_t np64 = (size_t)J * 64u;
  const int nxt = c->cur ^ 1;
  SnapBuf& s = c->snap[nxt];
  KChain lead{};
  lead.n = c->n_lead;
  for (uint32_t k = 0; k < c->n_lead; ++k) { lead.kind[k] = c->cfg.chain[k].kind; lead.w[k] = (double)c->cfg.chain[k].weight; }
  const uint32_t n64 = (uint32_t)np64;
  hipLaunchKernelGGL(snap_qrange_kernel, dim3(1), dim3(1024), 0, st, (const eppk_pod_row*)c->d_rows, n_pods, s.qrange);
  if (n64)
    hipLaunchKernelGGL(snap_terms_kernel, dim3((n64 + 255u) / 256u), dim3(256), 0, st, (const eppk_pod_row*)c->d_rows, n_pods, n64,
                       (const uint32_t*)s.qrange, lead, c->postc, s.base, s.post[0], s.post[1], s.queue, s.kv);
  int rc = by_lane_word(c, [&](auto tag) {
    using LW = decltype(tag);
    hipLaunchKernelGGL((snap_planes_kernel<LW>), dim3((131u * 64u + 255u) / 256u), dim3(256), 0, st, (const eppk_pod_row*)c->d_rows,
                       n_pods, J, (const uint32_t*)s.qrange, (LW*)s.thi_t, (LW*)s.tlo_t, (LW*)s.qmin_t, (LW*)s.qmax_t, (LW*)s.act_t, s.nat);
    if (c->canonical)
      hipLaunchKernelGGL((snap_top_kernel<LW>), dim3(129), dim3(256), np64 * 8u, st, (const double*)s.base, (const LW*)s.thi_t,
                         (const LW*)s.tlo_t, n_pods, n64, c->has_l ? 1u : 0u, c->tail, (const double*)s.post[0], (const double*)s.post[1],
                         (const LW*)s.act_t, s.topv, s.topi);
    return EPPK_OK;
  });
  HIPCHK(c, hipGetLastError());
  if (rc) return rc;
  c->cur = nxt;
  c->n_pods = n_pods;
  c->res_args_dirty = true;                  // (the resident kernel's argument block names the snapshot buffer)
  return EPPK_OK;
}

// Sum one field of the sharded index counters (synchronises the context's stream).
int ixc_sum(eppk_ctx* c, uint32_t field, unsigned long long* out) {
  unsigned long long h[eppk::kIxShards * 8u];
  HIPCHK(c, hipMemcpyAsync(h, c->ixc, sizeof h, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  unsigned long long t = 0;
  for (uint32_t sh = 0; sh < eppk::kIxShards; ++sh) t += h[sh * 8u + field];
  *out = t;
  return EPPK_OK;
}

int ensure_tmp(eppk_ctx* c, size_t bytes) {
  if (bytes <= c->d_tmp_bytes) return EPPK_OK;
  { const int rcp = resident_park(c); if (rcp) return rcp; }
  if (c->d_tmp) HIPCHK(c, hipFree(c->d_tmp));
  c->d_tmp = nullptr; c->d_tmp_bytes = 0;
  HIPCHK(c, hipMalloc(&c->d_tmp, bytes));
  c->d_tmp_bytes = bytes;
  return EPPK_OK;
}

}  // namespace

// ================================================================================================
extern "C" {

uint32_t eppk_abi_version(void) { return EPPK_ABI_VERSION; }

const char* eppk_last_error(const eppk_ctx* ctx) {
  if (ctx) return ctx->err.c_str();
  std::lock_guard<std::mutex> g(g_err_mu);
  static thread_local std::string copy;
  copy = g_create_err;
  return copy.c_str();
}

int eppk_create(const eppk_cfg* cfg, eppk_ctx** out) {
  if (!cfg || !out) return fail(nullptr, EPPK_ERR_ARG, "eppk_create: null argument");
  *out = nullptr;
  if (cfg->struct_size != sizeof(eppk_cfg)) return fail(nullptr, EPPK_ERR_ARG, "eppk_create: struct_size mismatch");
  if (cfg->max_pods == 0 || cfg->max_pods > EPPK_MAX_PODS) return fail(nullptr, EPPK_ERR_LIMIT, "eppk_create: max_pods out of range (1..4096)");
  if (cfg->max_blocks > EPPK_MAX_BLOCKS) return fail(nullptr, EPPK_ERR_LIMIT, "eppk_create: max_blocks > 256");
  if (cfg->n_scorers > EPPK_MAX_SCORERS) return fail(nullptr, EPPK_ERR_LIMIT, "eppk_create: n_scorers > 8");
  if (cfg->index_slots && ((cfg->index_slots & (cfg->index_slots - 1)) || cfg->index_slots < 64u || cfg->index_slots > (1u << 28)))
    return fail(nullptr, EPPK_ERR_ARG, "eppk_create: index_slots must be a power of two in [64, 2^28]");      // (2^29 physical slots: 32-bit slot numbers to spare)
  for (uint32_t k = 0; k < cfg->n_scorers; ++k)
    if (cfg->chain[k].kind < EPPK_SCORER_QUEUE || cfg->chain[k].kind > EPPK_SCORER_PREFIX)
      return fail(nullptr, EPPK_ERR_ARG, "eppk_create: unknown scorer kind");

  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0)
    return fail(nullptr, EPPK_ERR_DEVICE, std::string("eppk_create: no HIP device (") + hipGetErrorString(e) + "); libeppk has no CPU path");
  if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, EPPK_ERR_ARG, "eppk_create: device ordinal out of range");

  eppk_ctx* c = new (std::nothrow) eppk_ctx();
  if (!c) return fail(nullptr, EPPK_ERR_NOMEM, "eppk_create: out of memory");
  c->cfg = *cfg;
  auto bail = [&](int code) { std::string m = c->err; eppk_destroy(c); fail(nullptr, code, m); return code; };
#define CHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { c->err = std::string(#expr) + ": " + hipGetErrorString(e_); return bail(EPPK_ERR_DEVICE); } } while (0)
  CHK(hipSetDevice(cfg->device));
  hipDeviceProp_t prop;
  CHK(hipGetDeviceProperties(&prop, cfg->device));
  c->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if (prop.maxSharedMemoryPerMultiProcessor > c->max_lds) c->max_lds = prop.maxSharedMemoryPerMultiProcessor;
  CHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));

  if (const char* ft = getenv("EPPK_FAST_THREADS")) {
    const int v = atoi(ft);
    if (v >= 64 && v <= 1024 && v % 64 == 0) c->fast_threads = (uint32_t)v;
  }
  if (const char* mw = getenv("EPPK_MAX_WG_PER_CU")) c->max_wg_per_cu = atoi(mw) > 0 ? atoi(mw) : 0;
  if (const char* rs = getenv("EPPK_RESIDENT")) c->resident_on = atoi(rs) != 0;
  if (const char* qf = getenv("EPPK_RESIDENT_QUAD_FROM")) c->resident_quad_from = atoi(qf) > 0 ? (uint32_t)atoi(qf) : 1u;
  bool resident_max_set = false;
  if (const char* rm = getenv("EPPK_RESIDENT_MAX")) { c->resident_max = atoi(rm) > 0 ? (uint32_t)atoi(rm) : 0u; resident_max_set = true; }
  if (c->resident_max > 255u) c->resident_max = 255u;      // (the doorbell carries the request count in eight bits: res_bell_hi)
  if (c->resident_on && c->num_cu > (int)eppk_ctx::kResSlots) c->num_cu -= (int)eppk_ctx::kResSlots;    // a resident workgroup holds one CU (kResSlots of them at most): the persistent pick kernels are sized for the rest
  if (const char* qd = getenv("EPPK_QUAD")) c->quad_on = atoi(qd) != 0;
  if (const char* qm = getenv("EPPK_QUAD_MIN")) c->quad_min = atoi(qm) >= 4 ? (uint32_t)atoi(qm) : 4u;
  if (const char* qt = getenv("EPPK_QUAD_TAIL")) c->quad_tail_on = atoi(qt) != 0;
  if (const char* qp = getenv("EPPK_QUAD_PAUSE")) c->quad_pause_on = atoi(qp) != 0;
  if (const char* zc = getenv("EPPK_ZERO_COPY_MAX")) c->zero_copy_max = atoi(zc) > 0 ? (uint32_t)atoi(zc) : 0u;
  if (const char* hc = getenv("EPPK_HOST_CHECK_MAX")) c->host_check_max = atoi(hc) > 0 ? (uint32_t)atoi(hc) : 0u;
  if (const char* qt = getenv("EPPK_QUAD_THREADS")) {
    const int v = atoi(qt);
    if (v >= 64 && v <= EPPK_QUAD_MAX_THREADS && v % 64 == 0) c->quad_threads = (uint32_t)v;
  }
  CHK(hipHostMalloc((void**)&c->h_reports, (kReportRing + 1u) * sizeof(uint32_t), hipHostMallocDefault));   // (+ 1: the word the warm-up launch reports to)
  for (uint32_t i = 0; i <= kReportRing; ++i) c->h_reports[i] = 0u;
  c->lw_bytes = cfg->max_pods <= 1024 ? 2 : cfg->max_pods <= 2048 ? 4 : 8;
  c->npl = cfg->max_blocks <= 63 ? 6 : 9;
  c->pwn = (cfg->max_blocks + 2u) & ~1u;
  c->stride = 8u + 8u * cfg->max_blocks;
  c->jmax = (cfg->max_pods + 63u) / 64u;

  // chain analysis: leading pod-only scorers fuse into base[]
  c->kchain.n = cfg->n_scorers;
  for (uint32_t k = 0; k < cfg->n_scorers; ++k) { c->kchain.kind[k] = cfg->chain[k].kind; c->kchain.w[k] = (double)cfg->chain[k].weight; }
  uint32_t lead = 0;
  while (lead < cfg->n_scorers && (cfg->chain[lead].kind == EPPK_SCORER_QUEUE || cfg->chain[lead].kind == EPPK_SCORER_KV)) ++lead;
  c->n_lead = lead;
  for (uint32_t k = 0; k < lead; ++k) c->lead_queue |= cfg->chain[k].kind == EPPK_SCORER_QUEUE;
  // Fast path: at most one LORA and one PREFIX scorer, at most two pod-only scorers behind the first of them (each of those
  // needs its own per-pod product array: they are added one by one).  Everything else -> generic kernel.
  c->canonical = true;
  int nl = 0, np = 0;
  c->tail.n_tail = 0;
  for (uint32_t k = lead; k < cfg->n_scorers; ++k) {
    const uint32_t kind = cfg->chain[k].kind;
    uint32_t code;
    if (kind == EPPK_SCORER_LORA) { if (nl++ == 0 && np) c->p_first = true; code = 0u; }
    else if (kind == EPPK_SCORER_PREFIX) { ++np; code = 1u; }
    else {
      if (c->postc.n >= 2u) { c->canonical = false; break; }
      code = 2u + c->postc.n;
      c->postc.kind[c->postc.n] = kind;
      c->postc.w[c->postc.n] = (double)cfg->chain[k].weight;
      c->postc.n++;
      c->lead_queue |= kind == EPPK_SCORER_QUEUE;   // (any fused QUEUE term embeds the snapshot-wide normalisers)
    }
    if (c->tail.n_tail >= 4u) { c->canonical = false; break; }
    c->tail.kind[c->tail.n_tail++] = code;
  }
  if (nl > 1 || np > 1) c->canonical = false;
  c->has_l = nl > 0; c->has_p = np > 0;
  c->gen = c->canonical && c->postc.n > 0;
  if (!(c->has_l && c->has_p) || c->gen) c->p_first = false;
  if (c->canonical) {
    static const double tier_score[4] = {0.0, 0.6, 0.8, 1.0};
    double wl = 0.0, wp = 0.0;
    for (uint32_t k = lead; k < cfg->n_scorers; ++k) {
      if (cfg->chain[k].kind == EPPK_SCORER_LORA) wl = (double)cfg->chain[k].weight;
      if (cfg->chain[k].kind == EPPK_SCORER_PREFIX) wp = (double)cfg->chain[k].weight;
    }
    for (int t = 0; t < 4; ++t) c->tail.lw[t] = h_clamp01(tier_score[t]) * wl;
    c->tail.wp = wp;
  }

  const size_t np64 = (size_t)c->jmax * 64u;
  const size_t lora_bytes = ((size_t)EPPK_MAX_ADAPTERS + 1u) * 64u * (size_t)c->lw_bytes;   // + the base-model row
  {
    SnapLayout& L = c->lay;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off = (off + bytes + 255u) & ~(size_t)255u; return o; };
    // per-adapter tables first, back to back: the fast kernel addresses them with the compile-time offsets SnapOff<LW>
    L.topv = off; off += 129u * 64u * 8u;
    L.topi = off; off += 129u * 64u * 4u;
    L.thi = off; off += lora_bytes;
    L.tlo = off; off += lora_bytes;
    off += 2u * lora_bytes;                    // the interleaved {hi, lo} copy of the two planes (SnapOff<LW>::thl; snap_planes_kernel writes it)
    off += 129u * 16u * 16u;                   // the first 16 entries of every top table as {T, pod} pairs (SnapOff<LW>::top16; snap_top_kernel)
    off = (off + 255u) & ~(size_t)255u;
    L.base = take(np64 * 8u); L.post0 = take(np64 * 8u); L.post1 = take(np64 * 8u); L.queue = take(np64 * 4u); L.kv = take(np64 * 8u);
    L.qmin = take(64u * (size_t)c->lw_bytes); L.qmax = take(64u * (size_t)c->lw_bytes); L.act = take(64u * (size_t)c->lw_bytes); L.nat = take(3u * 64u * 8u); L.qrange = take(8u);
    L.bytes = off + 256u;                      // the LAST dword is the fast kernel's launch-status word (kBlobStatusTail)
    for (int b = 0; b < 2; ++b) {
      SnapBuf& s = c->snap[b];
      CHK(hipMalloc((void**)&s.blob, L.bytes));
      CHK(hipMemset(s.blob + L.bytes - 256u, 0, 256u));
      s.base = (double*)(s.blob + L.base); s.post[0] = (double*)(s.blob + L.post0); s.post[1] = (double*)(s.blob + L.post1);
      s.queue = (uint32_t*)(s.blob + L.queue); s.kv = (double*)(s.blob + L.kv);
      s.thi_t = s.blob + L.thi; s.tlo_t = s.blob + L.tlo;
      s.qmin_t = s.blob + L.qmin; s.qmax_t = s.blob + L.qmax; s.act_t = s.blob + L.act; s.nat = (uint64_t*)(s.blob + L.nat); s.qrange = (uint32_t*)(s.blob + L.qrange);
      s.topv = (double*)(s.blob + L.topv); s.topi = (uint32_t*)(s.blob + L.topi);
    }
  }
  if (c->canonical && c->has_p && cfg->max_blocks >= 1 && c->npl == 6) {   // the NPL == 6 fast kernels read it unconditionally
    // pterm[n][cnt] = clamp01((double)cnt / (double)n) * (double)w_prefix — the oracle's operations, done once here
    const uint32_t ld = cfg->max_blocks + 1u;
    std::vector<double> tab((size_t)ld * ld, 0.0);
    for (uint32_t n = 1; n <= cfg->max_blocks; ++n)
      for (uint32_t cnt = 0; cnt <= n; ++cnt) tab[(size_t)n * ld + cnt] = h_clamp01((double)cnt / (double)n) * c->tail.wp;
    CHK(hipMalloc((void**)&c->pterm, tab.size() * 8u));
    CHK(hipMemcpy(c->pterm, tab.data(), tab.size() * 8u, hipMemcpyHostToDevice));
    c->pterm_ld = ld;
  }
  CHK(hipHostMalloc((void**)&c->h_rows, (size_t)cfg->max_pods * sizeof(eppk_pod_row), hipHostMallocDefault));
  CHK(hipMalloc((void**)&c->d_rows, (size_t)cfg->max_pods * sizeof(eppk_pod_row)));
  CHK(hipMalloc((void**)&c->stats, (4 + 2 * (size_t)kStatSlots * kStatBanks) * sizeof(unsigned long long)));
  CHK(hipMemset(c->stats, 0, (4 + 2 * (size_t)kStatSlots * kStatBanks) * sizeof(unsigned long long)));
  CHK(hipMalloc((void**)&c->ixc, eppk::kIxShards * 8u * sizeof(unsigned long long)));
  CHK(hipMemset(c->ixc, 0, eppk::kIxShards * 8u * sizeof(unsigned long long)));
  CHK(hipMalloc((void**)&c->d_ixl, sizeof(eppk::IxLaunch)));
  CHK(hipMalloc((void**)&c->d_status, 2 * sizeof(uint32_t)));
  CHK(hipMemset(c->d_status, 0, 2 * sizeof(uint32_t)));
  if (cfg->index_slots) {
    // PHYSICAL slots = words of the key table: twice the API's index_slots, because five of a bucket's eight words hold keys (the other
    // three: flags and the keys' meta dwords -- tag + set id, eppk_kernels.hip.h: kBucket).  The capacity the API promises stays
    // index_slots / 2 live hashes: 40 % of the key words, about one key per bucket at the recommended sizing (a quarter of index_slots).
    c->slots = cfg->index_slots * 2u;
    uint32_t lg = 0;
    while ((1u << lg) < c->slots / kBucket) ++lg;   // 8-word buckets (eppk_kernels.hip.h: KIndex)
    c->shift = 32u - lg;
    c->limit = cfg->index_slots / 2u;  // load factor <= 0.5 of the API's slots
    // ONE allocation: pod-set rows first, the key table behind them (the fast kernel reads both through one descriptor)
    c->rows_bytes = (((size_t)c->slots + 3u) * 64u * (size_t)c->lw_bytes + 255u) & ~(size_t)255u;
    c->index_bytes = c->rows_bytes + ((size_t)c->slots + 2u) * 8u;
    CHK(hipMalloc(&c->bitmaps, c->index_bytes));
    c->keys = (uint64_t*)((uint8_t*)c->bitmaps + c->rows_bytes);
    CHK(hipMemset(c->bitmaps, 0, c->index_bytes));
    CHK(hipMalloc((void**)&c->rstamps, 2u * 4u));
    CHK(hipMemset(c->rstamps, 0, 2u * 4u));
    const char* le = getenv("EPPK_LISTS");
    c->list_routes = !(le && atoi(le) == 0);
    {
      // the set table: one line per 16 physical slots, 1024 at least (a line per distinct listed pod SET, not per key: the blocks of a
      // prefix share one); all-zero = every line free
      c->sets_cap = c->slots / 16u < 1024u ? 1024u : c->slots / 16u;
      const size_t nd = ((size_t)c->slots + 4u) * eppk::kListDwords, ns = (size_t)c->sets_cap * eppk::kListDwords;
      CHK(hipMalloc((void**)&c->lists, (nd + ns) * 4u));
      hipLaunchKernelGGL(eppk::lists_fill_kernel, dim3(1024), dim3(256), 0, c->stream, c->lists, nd);
      CHK(hipGetLastError());
      CHK(hipMemsetAsync(c->lists + nd, 0, ns * 4u, c->stream));
      CHK(hipMalloc((void**)&c->set_ctl, 2u * 4u));
      CHK(hipMemsetAsync(c->set_ctl, 0, 2u * 4u, c->stream));
      CHK(hipHostMalloc((void**)&c->h_set_report, 2u * 4u, hipHostMallocDefault));
      c->h_set_report[0] = c->h_set_report[1] = 0u;
      CHK(hipHostGetDevicePointer((void**)&c->h_set_report_dev, c->h_set_report, 0));
      CHK(hipStreamSynchronize(c->stream));
    }
  }
  CHK(hipDeviceSynchronize());
  // the resident path's default limit: with pick_quad_kernel's body behind the doorbell 64 requests take 16.6 us (21.1 launched), with
  // pick_fast_kernel's body alone 24.5 -- 32 stays the limit there
  if (c->resident_on && !resident_max_set && resident_quad(c)) c->resident_max = 64;
#undef CHK
  *out = c;
  return EPPK_OK;
}

void eppk_destroy(eppk_ctx* c) {
  if (!c) return;
  if (g_res_log && !g_res_logs.empty()) {
    const size_t from = g_res_logs.size() > 24u ? g_res_logs.size() - 24u : 0u;
    for (size_t i = from; i < g_res_logs.size(); ++i) {
      const ResLog& l = g_res_logs[i];
      std::fprintf(stderr, "[eppk resident] unit %u seq %u: wait %.2f us; stamps (10 ns ticks, measurement builds): rows copied %u, bell seen -> caches invalidated %u, -> body done %u, -> released %u\n",
                   l.unit, l.seq, l.wait_us, l.st[0], l.st[1], l.st[2], l.st[3]);
    }
    g_res_logs.clear();
  }
  (void)hipSetDevice(c->cfg.device);
  (void)resident_park(c);
  for (eppk_ctx::ResidentUnit& u : c->res) if (u.h_ctl) (void)hipHostFree(u.h_ctl);
  for (hipStream_t st : c->res_slot_stream) if (st) (void)hipStreamDestroy(st);
  (void)hipFree(c->d_res_args);
  (void)hipFree(c->d_res_wl);
  (void)hipFree(c->d_res_rows); (void)hipFree(c->d_res_learn); (void)hipFree(c->d_res_sortwl);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  for (int b = 0; b < 2; ++b) (void)hipFree(c->snap[b].blob);
  (void)hipFree(c->bitmaps); (void)hipFree(c->rstamps); (void)hipFree(c->lists); (void)hipFree(c->set_ctl); (void)hipHostFree(c->h_set_report); (void)hipFree(c->sortwl); (void)hipFree(c->d_ixl);
  (void)hipFree(c->d_at); (void)hipFree(c->d_av); (void)hipFree(c->d_sk); (void)hipFree(c->d_so);
  if (c->wait_ev) (void)hipEventDestroy(c->wait_ev);
  (void)hipFree(c->stats); (void)hipFree(c->pterm); (void)hipFree(c->d_status); (void)hipFree(c->ixc);
  (void)hipFree(c->d_tk_reqs); (void)hipFree(c->d_tk_mask); (void)hipFree(c->d_tk_pick); (void)hipFree(c->d_tk_score);
  (void)hipFree(c->d_reqs); (void)hipFree(c->d_mask); (void)hipFree(c->d_pick); (void)hipFree(c->d_score); (void)hipFree(c->d_tmp);
  for (uint32_t b = 0; b < 8; ++b) (void)hipFree(c->dsets[b].d);
  for (auto& s : c->stage) {
    if (s.st) { (void)hipStreamSynchronize(s.st); (void)hipStreamDestroy(s.st); }
    if (s.picked) (void)hipEventDestroy(s.picked);
    if (s.st_copy) { (void)hipStreamSynchronize(s.st_copy); (void)hipStreamDestroy(s.st_copy); }
    if (s.copied) (void)hipEventDestroy(s.copied);
    if (s.st_check) { (void)hipStreamSynchronize(s.st_check); (void)hipStreamDestroy(s.st_check); }
    if (s.checked) (void)hipEventDestroy(s.checked);
    if (s.h_bad) (void)hipHostFree(s.h_bad);
    (void)hipFree(s.d_reqs); (void)hipFree(s.d_mask); (void)hipFree(s.d_pick); (void)hipFree(s.d_score);
    if (s.h_reqs) (void)hipHostFree(s.h_reqs);
    if (s.h_mask) (void)hipHostFree(s.h_mask);
    if (s.h_pick) (void)hipHostFree(s.h_pick);
    if (s.h_score) (void)hipHostFree(s.h_score);
  }
  if (c->learned) (void)hipEventDestroy(c->learned);
  if (c->learn_words_free) (void)hipEventDestroy(c->learn_words_free);
  if (c->check.st) { (void)hipStreamSynchronize(c->check.st); (void)hipStreamDestroy(c->check.st); }
  if (c->check.done) (void)hipEventDestroy(c->check.done);
  if (c->check.h_bad) (void)hipHostFree(c->check.h_bad);
  if (c->h_reports) (void)hipHostFree((void*)c->h_reports);
  if (c->h_rows) (void)hipHostFree(c->h_rows);
  (void)hipFree(c->d_rows); (void)hipFree(c->d_rm); (void)hipFree(c->d_rs_pick); (void)hipFree(c->d_rs_score); (void)hipFree(c->d_learn);
  if (c->h_reqs) (void)hipHostFree(c->h_reqs);
  if (c->h_mask) (void)hipHostFree(c->h_mask);
  if (c->h_pick) (void)hipHostFree(c->h_pick);
  if (c->h_score) (void)hipHostFree(c->h_score);
  for (hipEvent_t e : c->ev) (void)hipEventDestroy(e);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

// ---- snapshot ------------------------------------------------------------------------------------

int eppk_snapshot_publish(eppk_ctx* c, const eppk_pod_row* rows, uint32_t n_pods, uint64_t epoch) {
  if (!c || (!rows && n_pods)) return fail(c, EPPK_ERR_ARG, "eppk_snapshot_publish: null argument");
  if (n_pods > c->cfg.max_pods) return fail(c, EPPK_ERR_LIMIT, "eppk_snapshot_publish: n_pods > max_pods");
  HIPCHK(c, hipSetDevice(c->cfg.device));
  { const int rcf = learn_fence(c, c->stream); if (rcf) return rcf; }      // (the index scrub below; a LEARN update reads the active-pods row of the current snapshot)

  // holes (flags & EPPK_POD_INACTIVE) are never candidates; the QUEUE normalisers range over the ACTIVE pods (snap_qrange_kernel)
  uint64_t act[64] = {0};                      // lane-transposed active set (u64 words serve every lane-word width)
  for (uint32_t p = 0; p < n_pods; ++p)
    if (!(rows[p].flags & EPPK_POD_INACTIVE)) act[p & 63u] |= 1ull << (p >> 6);
  // A slot that turned into a hole takes its cache knowledge with it: the index forgets it (SEMANTICS.md §6b), in ONE pass over
  // the table for all holes of this snapshot -- only when some slot became a hole since the last publish.
  bool new_hole = false;
  for (uint32_t l = 0; l < 64u; ++l) {
    const uint64_t exist = ((l < n_pods) ? (n_pods - l + 63u) / 64u : 0u) >= 64u ? ~0ull : ((1ull << ((l < n_pods) ? (n_pods - l + 63u) / 64u : 0u)) - 1ull);
    const uint64_t holes = exist & ~act[l];
    if (holes & ~c->h_holes[l]) new_hole = true;
    c->h_holes[l] = holes;
  }
  if (new_hole && c->slots) {
    int rcs = index_scrub(c, c->h_holes);
    if (rcs) return rcs;
  }

  // Snapshot producer on the device (eppk_kernels.hip.h "snapshot producer"): one H2D copy of the raw rows, then the queue
  // range, fused terms, tier planes and the per-adapter top-64 tables are built by four launches into the idle buffer.
  if (n_pods) {
    std::memcpy(c->h_rows, rows, (size_t)n_pods * sizeof(eppk_pod_row));
    HIPCHK(c, hipMemcpyAsync(c->d_rows, c->h_rows, (size_t)n_pods * sizeof(eppk_pod_row), hipMemcpyHostToDevice, c->stream));
  }
  int rc = rebuild_snapshot(c, n_pods, c->stream);
  if (rc) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->epoch = epoch;
  c->have_snapshot = true;
  c->have_addrs = false;     // the slot -> address table belongs to the snapshot it was set for: a publish may have re-mapped slots at the
                             // same pod count, and a filter resolved against the old table could admit a pod outside the subset
                             // (request.go:128-131 wants it strictly respected) -- eppk_snapshot_set_addresses again, or EPPK_ERR_NO_SNAPSHOT
  return EPPK_OK;
}

int eppk_snapshot_info(const eppk_ctx* c, uint32_t* n_pods, uint64_t* epoch) {
  if (!c) return EPPK_ERR_ARG;
  if (!c->have_snapshot) return EPPK_ERR_NO_SNAPSHOT;
  if (n_pods) *n_pods = c->n_pods;
  if (epoch) *epoch = c->epoch;
  return EPPK_OK;
}

// ---- prefix index --------------------------------------------------------------------------------

int eppk_index_clear(eppk_ctx* c) {
  if (!c) return EPPK_ERR_ARG;
  if (!c->slots) return EPPK_OK;
  HIPCHK(c, hipSetDevice(c->cfg.device));
  { const int rcf = learn_fence(c, c->stream); if (rcf) return rcf; }
  HIPCHK(c, hipMemsetAsync(c->bitmaps, 0, c->index_bytes, c->stream));
  HIPCHK(c, hipMemsetAsync(c->rstamps, 0, 2u * 4u, c->stream));
  c->min_live = c->index_epoch;            // (nothing is left that could be older)
  hipLaunchKernelGGL(eppk::lists_fill_kernel, dim3(1024), dim3(256), 0, c->stream, c->lists, ((size_t)c->slots + 4u) * eppk::kListDwords);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemsetAsync(c->lists + ((size_t)c->slots + 4u) * eppk::kListDwords, 0, (size_t)c->sets_cap * eppk::kListDwords * 4u, c->stream));   // the set table
  HIPCHK(c, hipMemsetAsync(c->set_ctl, 0, 2u * 4u, c->stream));
  c->h_set_report[0] = c->h_set_report[1] = 0u; c->set_fail_seen = 0u;
  HIPCHK(c, hipMemsetAsync(c->ixc, 0, eppk::kIxShards * 8u * sizeof(unsigned long long), c->stream));  // key / drop counters
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return EPPK_OK;
}

int eppk_index_insert(eppk_ctx* c, const uint64_t* hashes, const uint32_t* pods, uint32_t n) {
  if (!c || ((!hashes || !pods) && n)) return fail(c, EPPK_ERR_ARG, "eppk_index_insert: null argument");
  if (!c->slots) return fail(c, EPPK_ERR_ARG, "eppk_index_insert: context was created with index_slots = 0");
  for (uint32_t i = 0; i < n; ++i)
    if (pods[i] >= c->cfg.max_pods) return fail(c, EPPK_ERR_LIMIT, "eppk_index_insert: pod >= max_pods");
  if (n == 0) return EPPK_OK;
  HIPCHK(c, hipSetDevice(c->cfg.device));
  { const int rcf = learn_fence(c, c->stream); if (rcf) return rcf; }
  int rc = ensure_tmp(c, (size_t)n * 12u);
  if (rc) return rc;
  uint64_t* d_h = (uint64_t*)c->d_tmp;
  uint32_t* d_p = (uint32_t*)((uint8_t*)c->d_tmp + (size_t)n * 8u);
  unsigned long long before = 0, after = 0;
  int rcs = ixc_sum(c, eppk::kIxDropped, &before);     // (synchronises the stream: earlier asynchronous inserts are counted)
  if (rcs) return rcs;
  HIPCHK(c, hipMemcpyAsync(d_h, hashes, (size_t)n * 8u, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d_p, pods, (size_t)n * 4u, hipMemcpyHostToDevice, c->stream));
  const uint32_t threads = 256, grid = (n + threads - 1) / threads;
  eppk::SortWl sw{};
  rc = sortwl_begin(c, n, &sw);
  if (rc) return rc;
  hipLaunchKernelGGL(eppk::index_budget_kernel, dim3(1), dim3(64), 0, c->stream, c->ixc, c->limit, c->slots, (unsigned long long)n, c->d_ixl);
  rc = by_lane_word(c, [&](auto tag) {
    using LW = decltype(tag);
    hipLaunchKernelGGL((index_insert_kernel<LW>), dim3(grid), dim3(threads), 0, c->stream, c->keys, c->bitmaps, c->lists, c->rstamps, c->slots, c->shift,
                       c->limit, c->index_epoch, c->ixc, (const uint64_t*)d_h, (const uint32_t*)d_p, n,
                       c->have_snapshot ? (const LW*)c->snap[c->cur].act_t : (const LW*)nullptr, sw, c->d_status + (c->quiet_rows ? 1 : 0), (const eppk::IxLaunch*)c->d_ixl);
    return EPPK_OK;
  });
  HIPCHK(c, hipGetLastError());
  rc = sortwl_finish(c, sw, c->stream);
  if (rc) return rc;
  rcs = ixc_sum(c, eppk::kIxDropped, &after);
  if (rcs) return rcs;
  if (after != before) return fail(c, EPPK_ERR_INDEX_FULL, "eppk_index_insert: table at load limit, inserts dropped");
  return rc;
}

namespace {
// The post-route update "index[hash[r][i]] U= {pick[r]}" (0602-…/README.md:101-108) for a batch whose rows and picks are on the device:
// capacity verdict, update, re-sort of the lists it touched -- three launches on `st`.
int learn_picks(eppk_ctx* c, const void* d_reqs, const int32_t* d_picks, uint32_t n_reqs, hipStream_t st, const uint32_t* d_learn = nullptr) {
  if (n_reqs == 0 || c->cfg.max_blocks == 0) return EPPK_OK;
  { const int rcf = learn_fence(c, st); if (rcf) return rcf; }
  const uint64_t total = (uint64_t)n_reqs * c->cfg.max_blocks;
  const uint32_t threads = 256;
  const uint64_t grid64 = (total + threads - 1) / threads;
  if (grid64 > 0x7FFFFFFFull) return fail(c, EPPK_ERR_LIMIT, "eppk_index_insert_picks_device: batch too large");
  eppk::SortWl sw{};
  int rc = sortwl_begin(c, total, &sw);
  if (rc) return rc;
  hipLaunchKernelGGL(eppk::index_budget_kernel, dim3(1), dim3(64), 0, st, c->ixc, c->limit, c->slots, (unsigned long long)total, c->d_ixl);
  rc = by_lane_word(c, [&](auto tag) {
    using LW = decltype(tag);
    hipLaunchKernelGGL((index_insert_picks_kernel<LW>), dim3((uint32_t)grid64), dim3(threads), 0, st, c->keys, c->bitmaps, c->lists, c->rstamps, c->slots,
                       c->shift, c->limit, c->index_epoch, c->ixc, (const uint8_t*)d_reqs, c->stride, c->cfg.max_blocks, d_picks, n_reqs,
                       c->cfg.max_pods, c->d_status + (c->quiet_rows ? 1 : 0), c->have_snapshot ? (const LW*)c->snap[c->cur].act_t : (const LW*)nullptr, sw, (const eppk::IxLaunch*)c->d_ixl,
                       d_learn);
    return EPPK_OK;
  });
  HIPCHK(c, hipGetLastError());
  if (rc) return rc;
  return sortwl_finish(c, sw, st);
}
}  // namespace

int eppk_index_insert_picks_device(eppk_ctx* c, const void* d_reqs, const int32_t* d_picks, uint32_t n_reqs, void* stream) {
  if (!c || ((!d_reqs || !d_picks) && n_reqs)) return fail(c, EPPK_ERR_ARG, "eppk_index_insert_picks_device: null argument");
  if (!c->slots) return fail(c, EPPK_ERR_ARG, "eppk_index_insert_picks_device: no index");
  HIPCHK(c, hipSetDevice(c->cfg.device));
  return learn_picks(c, d_reqs, d_picks, n_reqs, stream ? (hipStream_t)stream : c->stream);
}

int eppk_index_remove_pod(eppk_ctx* c, uint32_t pod) {
  if (!c) return EPPK_ERR_ARG;
  if (pod >= c->cfg.max_pods) return fail(c, EPPK_ERR_LIMIT, "eppk_index_remove_pod: pod >= max_pods");
  if (!c->slots) return EPPK_OK;
  HIPCHK(c, hipSetDevice(c->cfg.device));
  { const int rcf = learn_fence(c, c->stream); if (rcf) return rcf; }
  const uint32_t rows = c->slots + 2u, threads = 256;
  uint32_t grid = (rows * 64u + threads - 1) / threads;
  if (grid > 4096u) grid = 4096u;
  eppk::SortWl sw{};
  int rc = sortwl_begin(c, 1u << 16, &sw);
  if (rc) return rc;
  rc = by_lane_word(c, [&](auto tag) {
    using LW = decltype(tag);
    hipLaunchKernelGGL((index_remove_pod_kernel<LW>), dim3(grid), dim3(threads), 0, c->stream, c->keys, c->bitmaps, c->lists, c->slots, pod, c->ixc,
                       (const LW*)nullptr, sw);
    return EPPK_OK;
  });
  HIPCHK(c, hipGetLastError());
  if (rc == EPPK_OK) rc = sortwl_finish(c, sw, c->stream);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return rc;
}

int eppk_index_size(eppk_ctx* c, uint32_t* n_entries) {
  if (!c || !n_entries) return EPPK_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->cfg.device));
  unsigned long long live = 0;
  { const int rcs_ = device_sync(c); if (rcs_) return rcs_; }      // (asynchronous updates on the caller's streams included)
  int rc = ixc_sum(c, eppk::kIxLive, &live);   // live keys (kIxWords counts non-empty words, tombstones included)
  if (rc) return rc;
  *n_entries = (uint32_t)live;
  return EPPK_OK;
}

int eppk_index_dropped(eppk_ctx* c, uint64_t* n_dropped) {
  if (!c || !n_dropped) return EPPK_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->cfg.device));
  unsigned long long d = 0;
  { const int rcs_ = device_sync(c); if (rcs_) return rcs_; }
  int rc = ixc_sum(c, eppk::kIxDropped, &d);
  if (rc) return rc;
  *n_dropped = (uint64_t)d;
  return EPPK_OK;
}

int eppk_index_selfcheck(eppk_ctx* c, uint64_t* n_bad) {
  if (!c || !n_bad) return EPPK_ERR_ARG;
  *n_bad = 0;
  if (!c->slots) return EPPK_OK;
  HIPCHK(c, hipSetDevice(c->cfg.device));
  { const int rcf = learn_fence(c, c->stream); if (rcf) return rcf; }
  int rct = ensure_tmp(c, 256u * 8u);                                          // bad count | records | eight offenders in full (index_selfcheck_kernel)
  if (rct) return rct;
  unsigned long long* d_bad = (unsigned long long*)c->d_tmp;
  HIPCHK(c, hipMemsetAsync(d_bad, 0, 256u * 8u, c->stream));
  const uint32_t rows = c->slots + 3u, threads = 256;
  uint32_t grid = (rows * 64u + threads - 1) / threads;
  if (grid > 4096u) grid = 4096u;
  int rc = by_lane_word(c, [&](auto tag) {
    using LW = decltype(tag);
    hipLaunchKernelGGL((index_selfcheck_kernel<LW>), dim3(grid), dim3(threads), 0, c->stream, (const uint64_t*)c->keys, (const void*)c->bitmaps,
                       (const uint32_t*)c->lists, c->slots, c->sets_cap - 1u, d_bad);
    return EPPK_OK;
  });
  HIPCHK(c, hipGetLastError());
  unsigned long long h_bad[256];
  HIPCHK(c, hipMemcpyAsync(h_bad, d_bad, sizeof h_bad, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  *n_bad = (uint64_t)h_bad[0];
  if (h_bad[0] && getenv("EPPK_SELFCHECK_VERBOSE")) {
    for (unsigned long long k = 0; k < h_bad[1] && k < 8ull; ++k) {
      const unsigned long long* r = h_bad + 2 + 24 * k;
      std::fprintf(stderr, "[eppk selfcheck] slot %llu why 0x%llx key %016llx row members %llu bad-entry lanes %llx list:", r[0], r[1], r[2], r[3], r[4]);
      for (int i = 0; i < 16; ++i) std::fprintf(stderr, " %08llx", r[5 + i]);
      std::fprintf(stderr, "\n");
    }
  }
  return rc;
}

int eppk_index_advance_epoch(eppk_ctx* c, uint32_t* new_epoch) {
  if (!c) return EPPK_ERR_ARG;
  if (c->index_epoch == 0xFFFFFFFFu) return fail(c, EPPK_ERR_LIMIT, "eppk_index_advance_epoch: epoch counter exhausted (clear the index)");
  // The window of the 8-bit stamp tags (SEMANTICS.md 6a): after the tick to epoch e no live hash may be stamped before e - 254.  The
  // eviction runs BEFORE the tick, at epoch e - 1: a hash stamped at e - 255 is 254 epochs old there -- the largest age a tag can
  // express -- and goes (keep = 253).  Behind the tick it would be 255 epochs old, its tag would equal the new epoch's and it would
  // read as age 0 for ever (round-4 defect: the scan ran behind the tick with keep = 254, which no tag age exceeds).
  // A shim that ages its index (evict_older(epoch - keep) with a keep of a few epochs) never gets here; one that never evicts, or
  // keeps 254 epochs and more, pays a scan per tick from the 255th.
  const uint32_t next = c->index_epoch + 1u;
  if (c->slots && next > kEpochWindow && c->min_live < next - kEpochWindow) {
    HIPCHK(c, hipSetDevice(c->cfg.device));
    { const int rcs_ = device_sync(c); if (rcs_) return rcs_; }               // (index updates the caller may have in flight on streams of its own)
    uint32_t gone = 0;
    const int rc = eppk_index_evict_older(c, next - kEpochWindow, &gone);
    if (rc) return rc;
  }
  c->index_epoch = next;
  c->res_args_dirty = true;                  // (the LEARN resident units stamp with the epoch of their argument block)
  if (new_epoch) *new_epoch = c->index_epoch;
  return EPPK_OK;
}

int eppk_index_evict_older(eppk_ctx* c, uint32_t min_epoch, uint32_t* n_evicted) {
  if (!c) return EPPK_ERR_ARG;
  if (n_evicted) *n_evicted = 0;
  if (!c->slots) return EPPK_OK;
  HIPCHK(c, hipSetDevice(c->cfg.device));
  { const int rcf = learn_fence(c, c->stream); if (rcf) return rcf; }
  for (uint32_t sh = 0; sh < eppk::kIxShards; ++sh)     // "evicted by this launch" (sharded like the other index counters)
    HIPCHK(c, hipMemsetAsync(c->ixc + sh * 8u + eppk::kIxEvicted, 0, sizeof(unsigned long long), c->stream));
  const uint32_t rows = c->slots + 2u, threads = 256;
  uint32_t grid = (rows * 64u + threads - 1) / threads;
  if (grid > 4096u) grid = 4096u;
  int rc = by_lane_word(c, [&](auto tag) {
    using LW = decltype(tag);
    hipLaunchKernelGGL((index_evict_kernel<LW>), dim3(grid), dim3(threads), 0, c->stream, c->keys, c->bitmaps, c->lists, (const uint32_t*)c->rstamps, c->slots,
                       c->index_epoch, min_epoch, c->ixc);
    return EPPK_OK;
  });
  HIPCHK(c, hipGetLastError());
  { const uint32_t horizon = min_epoch < c->index_epoch ? min_epoch : c->index_epoch; if (horizon > c->min_live) c->min_live = horizon; }
  unsigned long long ev = 0;
  int rcs = ixc_sum(c, eppk::kIxEvicted, &ev);
  if (rcs) return rcs;
  if (n_evicted) *n_evicted = (uint32_t)ev;
  return rc;
}

int eppk_index_trim_pods(eppk_ctx* c, uint32_t cap, uint64_t* n_removed) {
  if (!c) return EPPK_ERR_ARG;
  if (n_removed) *n_removed = 0;
  if (!c->slots) return EPPK_OK;
  HIPCHK(c, hipSetDevice(c->cfg.device));
  { const int rcf = learn_fence(c, c->stream); if (rcf) return rcf; }
  // scratch: hist[4096][64] u32 | cutage[4096] u32 | over_t[64] u64 | removed u64
  const size_t hist_b = 4096u * (size_t)eppk::kTrimBins * 4u, cut_b = 4096u * 4u, over_b = 64u * 8u;
  int rc = ensure_tmp(c, hist_b + cut_b + over_b + 8u);
  if (rc) return rc;
  uint32_t* hist = (uint32_t*)c->d_tmp;
  uint32_t* cutage = (uint32_t*)((uint8_t*)c->d_tmp + hist_b);
  uint64_t* over_t = (uint64_t*)((uint8_t*)c->d_tmp + hist_b + cut_b);
  unsigned long long* removed = (unsigned long long*)((uint8_t*)c->d_tmp + hist_b + cut_b + over_b);
  HIPCHK(c, hipMemsetAsync(c->d_tmp, 0, hist_b + cut_b + over_b + 8u, c->stream));
  const uint32_t rows = c->slots + 2u, threads = 256;
  uint32_t grid = (rows * 64u + threads - 1) / threads;
  if (grid > 4096u) grid = 4096u;
  eppk::SortWl sw{};
  rc = sortwl_begin(c, 1u << 16, &sw);
  if (rc) return rc;
  rc = by_lane_word(c, [&](auto tag) {
    using LW = decltype(tag);
    hipLaunchKernelGGL((index_pod_hist_kernel<LW>), dim3(grid), dim3(threads), 0, c->stream, (const uint64_t*)c->keys, (const void*)c->bitmaps,
                       (const uint32_t*)c->lists, (const uint32_t*)c->rstamps, c->slots, c->index_epoch, hist);
    hipLaunchKernelGGL(index_pod_cut_kernel, dim3(1), dim3(1024), 0, c->stream, (const uint32_t*)hist, c->cfg.max_pods, cap, cutage, over_t);
    hipLaunchKernelGGL((index_pod_trim_kernel<LW>), dim3(grid), dim3(threads), 0, c->stream, c->keys, c->bitmaps, c->lists, (const uint32_t*)c->rstamps,
                       c->slots, c->index_epoch, (const uint32_t*)cutage, (const uint64_t*)over_t, c->ixc, removed, sw);
    return EPPK_OK;
  });
  HIPCHK(c, hipGetLastError());
  if (rc == EPPK_OK) rc = sortwl_finish(c, sw, c->stream);
  unsigned long long rm = 0;
  HIPCHK(c, hipMemcpyAsync(&rm, removed, sizeof rm, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (n_removed) *n_removed = (uint64_t)rm;
  return rc;
}

int eppk_index_evict_older_device(eppk_ctx* c, uint32_t min_epoch, void* stream) {
  if (!c) return EPPK_ERR_ARG;
  if (!c->slots) return EPPK_OK;
  HIPCHK(c, hipSetDevice(c->cfg.device));
  hipStream_t st = stream ? (hipStream_t)stream : c->stream;
  { const int rcf = learn_fence(c, st); if (rcf) return rcf; }
  // The pipelined host path may have a staging set between begin and end: the eviction queues behind that set's pick (which reads the
  // index; its LEARN update, if any, is behind the fence above) and every later begin queues behind the eviction (the `learned` event
  // again: run_pick's fence) -- a shim ages its index without draining the two-set pipeline.
  for (uint32_t i = 0; i < EPPK_STAGE_SETS; ++i) {
    eppk_ctx::StageSet& s = c->stage[i];
    if (s.busy && s.n && s.picked && st != s.st) HIPCHK(c, hipStreamWaitEvent(st, s.picked, 0));
  }
  const uint32_t rows = c->slots + 2u, threads = 256;
  uint32_t grid = (rows * 64u + threads - 1) / threads;
  if (grid > 4096u) grid = 4096u;
  int rc = by_lane_word(c, [&](auto tag) {
    using LW = decltype(tag);
    hipLaunchKernelGGL((index_evict_kernel<LW>), dim3(grid), dim3(threads), 0, st, c->keys, c->bitmaps, c->lists, (const uint32_t*)c->rstamps, c->slots,
                       c->index_epoch, min_epoch, c->ixc);     // (stats[0], the per-launch count of the synchronous form, just accumulates here)
    return EPPK_OK;
  });
  HIPCHK(c, hipGetLastError());
  if (c->learned) {                        // (only a context that uses the staging sets has the event: a single-stream closed loop pays nothing)
    HIPCHK(c, hipEventRecord(c->learned, st));
    c->learn_pending = true;
  }
  { const uint32_t horizon = min_epoch < c->index_epoch ? min_epoch : c->index_epoch; if (horizon > c->min_live) c->min_live = horizon; }
  return rc;
}

// ---- the hot path ----------------------------------------------------------------------------------

int eppk_pick_batch_device(eppk_ctx* c, const void* d_reqs, uint32_t n_reqs, const uint64_t* d_cand_mask, int32_t* d_out_pick,
                           double* d_out_score, void* stream) {
  if (!c || ((!d_reqs || !d_out_pick) && n_reqs)) return fail(c, EPPK_ERR_ARG, "eppk_pick_batch_device: null argument");
  if (!c->have_snapshot) return fail(c, EPPK_ERR_NO_SNAPSHOT, "eppk_pick_batch_device: no snapshot published");
  if (n_reqs == 0) return EPPK_OK;
  HIPCHK(c, hipSetDevice(c->cfg.device));
  hipStream_t st = stream ? (hipStream_t)stream : c->stream;
  // the kernels address request rows with 32-bit byte offsets: split batches of 2 GiB and more
  const uint32_t per = (uint32_t)((1ull << 31) / c->stride);
  const size_t J = (c->n_pods + 63u) / 64u;
  for (uint32_t r0 = 0; r0 < n_reqs; r0 += per) {
    const uint32_t n = (n_reqs - r0 < per) ? n_reqs - r0 : per;
    int rc = run_pick(c, (const uint8_t*)d_reqs + (size_t)r0 * c->stride, n, d_cand_mask ? d_cand_mask + (size_t)r0 * J : nullptr,
                      d_out_pick + r0, d_out_score ? d_out_score + r0 : nullptr, st, 1u, false, 0ull, 0u);
    if (rc) return rc;
  }
  return EPPK_OK;
}

namespace {
// The learn words live in ONE buffer per context (d_learn): a LEARN pick on stream `st` may only overwrite them once the update
// behind the previous LEARN pick -- possibly on another stream of the caller's -- has read them.  Free in the single-stream case: the
// event is recorded (on the PREVIOUS stream, behind everything queued there) only when the stream changes.
int learn_words_fence(eppk_ctx* c, hipStream_t st) {
  if (c->learn_words_stream && c->learn_words_stream != st) {
    if (!c->learn_words_free) HIPCHK(c, hipEventCreateWithFlags(&c->learn_words_free, hipEventDisableTiming));
    if (hipEventRecord(c->learn_words_free, c->learn_words_stream) == hipSuccess) {
      HIPCHK(c, hipStreamWaitEvent(st, c->learn_words_free, 0));
    } else {                                  // (the caller has destroyed that stream meanwhile: whatever ran on it is waited for wholesale)
      (void)hipGetLastError();
      const int rcs_ = device_sync(c); if (rcs_) return rcs_;
    }
  }
  c->learn_words_stream = st;
  return EPPK_OK;
}
// room for the learn words of n requests (grown behind a device synchronise: rare)
int learn_ensure(eppk_ctx* c, size_t n) {
  if (n <= c->learn_cap) return EPPK_OK;
  { const int rcs_ = device_sync(c); if (rcs_) return rcs_; }
  if (c->d_learn) HIPCHK(c, hipFree(c->d_learn));
  c->d_learn = nullptr; c->learn_cap = 0;
  const size_t cap = n < c->cfg.max_batch ? c->cfg.max_batch : n;
  HIPCHK(c, hipMalloc((void**)&c->d_learn, cap * 4u));
  c->learn_cap = cap;
  return EPPK_OK;
}
}  // namespace

int eppk_pick_learn_device(eppk_ctx* c, const void* d_reqs, uint32_t n_reqs, const uint64_t* d_cand_mask, int32_t* d_out_pick,
                           double* d_out_score, void* stream) {
  if (!c || ((!d_reqs || !d_out_pick) && n_reqs)) return fail(c, EPPK_ERR_ARG, "eppk_pick_learn_device: null argument");
  if (!c->have_snapshot) return fail(c, EPPK_ERR_NO_SNAPSHOT, "eppk_pick_learn_device: no snapshot published");
  if (!c->slots) return fail(c, EPPK_ERR_ARG, "eppk_pick_learn_device: no index");
  if (n_reqs == 0) return EPPK_OK;
  HIPCHK(c, hipSetDevice(c->cfg.device));
  hipStream_t st = stream ? (hipStream_t)stream : c->stream;
  int rc = learn_ensure(c, n_reqs);
  if (rc) return rc;
  rc = learn_words_fence(c, st);
  if (rc) return rc;
  // (the kernels address request rows with 32-bit byte offsets: batches of 2 GiB and more in pieces, each picked and learned in turn)
  const uint32_t per = (uint32_t)((1ull << 31) / c->stride);
  const size_t J = (c->n_pods + 63u) / 64u;
  for (uint32_t r0 = 0; r0 < n_reqs; r0 += per) {
    const uint32_t n = (n_reqs - r0 < per) ? n_reqs - r0 : per;
    const uint8_t* reqs = (const uint8_t*)d_reqs + (size_t)r0 * c->stride;
    bool words = false;
    rc = run_pick(c, reqs, n, d_cand_mask ? d_cand_mask + (size_t)r0 * J : nullptr, d_out_pick + r0, d_out_score ? d_out_score + r0 : nullptr, st, 1u, false, 0ull, 0u,
                  c->d_learn, &words);
    if (rc) return rc;
    rc = learn_picks(c, reqs, d_out_pick + r0, n, st, words ? c->d_learn : nullptr);
    if (rc) return rc;
  }
  return EPPK_OK;
}

int eppk_stream_wait_pick(eppk_ctx* c, void* waiting_stream) {
  if (!c || !waiting_stream) return fail(c, EPPK_ERR_ARG, "eppk_stream_wait_pick: null argument");
  HIPCHK(c, hipSetDevice(c->cfg.device));
  hipEvent_t ev = c->last_done;
  if (!ev) {                                  // the launch carried no completion event: record one behind it
    if (!c->wait_ev) HIPCHK(c, hipEventCreateWithFlags(&c->wait_ev, hipEventDisableTiming));
    ev = c->wait_ev;
    HIPCHK(c, hipEventRecord(ev, c->last_stream ? c->last_stream : c->stream));
  }
  HIPCHK(c, hipStreamWaitEvent((hipStream_t)waiting_stream, ev, 0));
  return EPPK_OK;
}

// Host-buffer pick in two halves, so that a group (eppk_group_pick_batch) can overlap the devices: `begin` validates, stages and
// enqueues H2D + kernel + D2H on the context's stream; `end` waits and copies out.  `h_src` (optional) = a pinned, portable copy of
// the rows that the H2D may read directly (the group stages a batch ONCE for all devices); `full_n` > n_reqs uploads rows
// [0, full_n) of `h_src` / reqs_base instead of only this context's shard (a group that learns prefixes needs the whole batch on
// every device) and the shard starts at row `lo` of them.
namespace {

// While one of these lives, the kernels this context enqueues raise "request row out of range" on a word of their own instead of the
// sticky flag of the *_device entry points: a host-buffer call reports the row itself (EPPK_ERR_ARG naming it), and a flag that an
// earlier or concurrent *_device launch of the same context raised legitimately must survive that (eppk_launch_status).
struct QuietRows {
  eppk_ctx* c;
  explicit QuietRows(eppk_ctx* c_) : c(c_) { c->quiet_rows = true; }
  ~QuietRows() { c->quiet_rows = false; }
};

// Range check of request-row headers on the device: the lowest row out of range lands in *bad (pinned host word, 0xFFFFFFFF = none).
__global__ void rows_check_kernel(const uint8_t* __restrict__ reqs, uint32_t stride, uint32_t n, uint32_t max_blocks, uint32_t* bad) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const eppk_req_hdr h = *(const eppk_req_hdr*)(reqs + (size_t)r * stride);
  if (h.n_blocks > max_blocks || h.adapter < -1 || h.adapter >= (int32_t)EPPK_MAX_ADAPTERS)
    __hip_atomic_fetch_min(bad, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

int rows_check_launch(eppk_ctx* c, const void* rows_dev, uint32_t n, uint32_t* h_bad, uint32_t* h_bad_dev, hipStream_t st) {
  *h_bad = 0xFFFFFFFFu;
  hipLaunchKernelGGL(rows_check_kernel, dim3((n + 255u) / 256u), dim3(256), 0, st, (const uint8_t*)rows_dev, c->stride, n, c->cfg.max_blocks, h_bad_dev);
  HIPCHK(c, hipGetLastError());
  return EPPK_OK;
}

// A batch with a row out of range has been scored (that row: EPPK_NO_PICK; its kernels ran under QuietRows, so the sticky flag of the
// *_device entry points is untouched): fail the call, naming the row.
int rows_check_fail(eppk_ctx* c, const char* who, uint32_t row) {
  return fail(c, EPPK_ERR_ARG, std::string(who ? who : "eppk_pick_batch") + ": request row " + std::to_string(row) + " out of range");
}

int row_check_ensure(eppk_ctx* c, uint32_t** h_bad, uint32_t** h_bad_dev, hipStream_t* st, hipEvent_t* ev) {
  if (!*h_bad) {
    HIPCHK(c, hipHostMalloc((void**)h_bad, 64, hipHostMallocDefault));
    HIPCHK(c, hipHostGetDevicePointer((void**)h_bad_dev, *h_bad, 0));
    **h_bad = 0xFFFFFFFFu;
    HIPCHK(c, hipStreamCreateWithFlags(st, hipStreamNonBlocking));
    HIPCHK(c, hipEventCreateWithFlags(ev, hipEventDisableTiming));
  }
  return EPPK_OK;
}

int validate_rows(eppk_ctx* c, const char* who, const void* reqs, uint32_t n_reqs, uint32_t first_row) {
  for (uint32_t r = 0; r < n_reqs; ++r) {
    eppk_req_hdr h;
    std::memcpy(&h, (const uint8_t*)reqs + (size_t)r * c->stride, sizeof h);
    if (h.n_blocks > c->cfg.max_blocks || h.adapter < -1 || h.adapter >= (int32_t)EPPK_MAX_ADAPTERS)
      return fail(c, EPPK_ERR_ARG, std::string(who) + ": request row " + std::to_string(first_row + r) + " out of range");
  }
  return EPPK_OK;
}

int ensure_host_staging(eppk_ctx* c, bool need_mask) {
  const size_t mb = c->cfg.max_batch;
  if (!c->d_reqs) {
    HIPCHK(c, hipMalloc(&c->d_reqs, mb * c->stride));
    HIPCHK(c, hipMalloc((void**)&c->d_pick, (mb + EPPK_GROUP_MAX_DEVICES) * 4u));   // (+ the padding of a group's in-place all-gather)
    HIPCHK(c, hipMalloc((void**)&c->d_score, mb * 8u));
    HIPCHK(c, hipHostMalloc(&c->h_reqs, mb * c->stride, hipHostMallocDefault));
    HIPCHK(c, hipHostMalloc((void**)&c->h_pick, mb * 4u, hipHostMallocDefault));
    HIPCHK(c, hipHostMalloc((void**)&c->h_score, mb * 8u, hipHostMallocDefault));
    // (what the device calls the same memory: the zero-copy path of small batches hands these to the kernel)
    HIPCHK(c, hipHostGetDevicePointer(&c->h_reqs_dev, c->h_reqs, 0));
    HIPCHK(c, hipHostGetDevicePointer((void**)&c->h_pick_dev, c->h_pick, 0));
    HIPCHK(c, hipHostGetDevicePointer((void**)&c->h_score_dev, c->h_score, 0));
    c->res_args_dirty = true;                // (the resident kernels' argument blocks name these buffers)
  }
  if (need_mask && !c->d_mask) {
    HIPCHK(c, hipMalloc((void**)&c->d_mask, mb * c->jmax * 8u));
    HIPCHK(c, hipHostMalloc((void**)&c->h_mask, mb * c->jmax * 8u, hipHostMallocDefault));
    HIPCHK(c, hipHostGetDevicePointer((void**)&c->h_mask_dev, c->h_mask, 0));
    c->res_args_dirty = true;
  }
  return EPPK_OK;
}

// rows [lo, lo + n) of a batch of `full_n` rows starting at `base` (host memory; `pinned` = the device may DMA from it directly).
// upload_all: rows [0, full_n) go to d_reqs (the shard is then d_reqs + lo * stride), else only the shard (at d_reqs).
// mask_on_device: c->d_mask already holds the shard's mask rows (built by subset_masks_kernel on the context stream)
// validate_as != nullptr: the rows are validated on the way (chunk by chunk, while the previous chunk is on the PCIe link), under that name
int pick_host_begin(eppk_ctx* c, const uint8_t* base, bool pinned, uint32_t full_n, uint32_t lo, uint32_t n, bool upload_all,
                    const uint64_t* cand_mask_shard, bool mask_on_device = false, const char* validate_as = nullptr, bool allow_zero_copy = false) {
  HIPCHK(c, hipSetDevice(c->cfg.device));
  const size_t J = (c->n_pods + 63u) / 64u;
  int rc = ensure_host_staging(c, cand_mask_shard != nullptr || mask_on_device);
  if (rc) return rc;
  c->check.pending = false;
  // rows that the call validates itself (on the host, or by rows_check_kernel) report through the call, not through the sticky flag
  struct MaybeQuiet { eppk_ctx* c; bool on; MaybeQuiet(eppk_ctx* c_, bool on_) : c(c_), on(on_) { if (on) c->quiet_rows = true; } ~MaybeQuiet() { if (on) c->quiet_rows = false; } } quiet(c, validate_as != nullptr);
  if (allow_zero_copy && n != 0 && n <= c->zero_copy_max && !upload_all && lo == 0 && full_n == n && !mask_on_device) {
    // ZERO-COPY (a small batch of one context): rows (and mask) into the pinned staging buffers unless they are there already, then
    // ONE launch that reads them over PCIe and writes the results into the pinned result buffers; pick_host_end waits for the stream.
    if (base != (const uint8_t*)c->h_reqs) std::memcpy(c->h_reqs, base, (size_t)n * c->stride);
    const bool dev_check = validate_as && n > c->host_check_max;
    if (validate_as && !dev_check) { rc = validate_rows(c, validate_as, c->h_reqs, n, 0u); if (rc) return rc; }
    const bool use_mask = cand_mask_shard != nullptr && J != 0;
    if (use_mask && cand_mask_shard != c->h_mask) std::memcpy(c->h_mask, cand_mask_shard, (size_t)n * J * 8u);
    rc = run_pick(c, (const uint8_t*)c->h_reqs_dev, n, use_mask ? c->h_mask_dev : nullptr, c->h_pick_dev, c->h_score_dev, c->stream, 1u, false, 0ull, 0u);
    if (rc) return rc;
    if (dev_check) {        // beside the pick, on a stream of its own (it reads the row headers over PCIe as the pick does the rows)
      rc = row_check_ensure(c, &c->check.h_bad, &c->check.h_bad_dev, &c->check.st, &c->check.done);
      if (rc) return rc;
      rc = rows_check_launch(c, c->h_reqs_dev, n, c->check.h_bad, c->check.h_bad_dev, c->check.st);
      if (rc) return rc;
      HIPCHK(c, hipEventRecord(c->check.done, c->check.st));
      c->check.pending = true; c->check.side = true; c->check.who = validate_as;
    }
    return EPPK_OK;
  }
  const uint32_t up_lo = upload_all ? 0u : lo, up_n = upload_all ? full_n : n;
  const uint8_t* src = base + (size_t)up_lo * c->stride;
  if (up_n) {
    if (pinned) {
      // ONE DMA for all rows.  (In 2 MB chunks, each validated by the host before its own copy, the copy engine paid a hand-off between
      // dependent copies nine times per 64k-request batch and the host loop itself took longer than the link: p50 0.52 ms against a
      // PCIe time of 0.31.)
      HIPCHK(c, hipMemcpyAsync(c->d_reqs, src, (size_t)up_n * c->stride, hipMemcpyHostToDevice, c->stream));
      if (validate_as) {    // on the device, behind the upload and in front of the pick (4 MB of headers out of HBM: microseconds)
        rc = row_check_ensure(c, &c->check.h_bad, &c->check.h_bad_dev, &c->check.st, &c->check.done);
        if (rc) return rc;
        rc = rows_check_launch(c, c->d_reqs, up_n, c->check.h_bad, c->check.h_bad_dev, c->stream);
        if (rc) return rc;
        c->check.pending = true; c->check.side = false; c->check.who = validate_as;
      }
    } else {
      // In chunks of whole rows: pageable caller memory goes through the pinned staging buffer, and the copy of chunk i + 1 runs while
      // chunk i is on its way over PCIe (one pass over 17 MB followed by one DMA of 17 MB was 1.2 ms per 64k-request batch).
      // (Helper threads for that copy -- 1, 3 or 7, chunks handed out by an atomic cursor, a third of the batch per DMA -- changed
      // nothing: 0.63-0.67 ms per 64k-request batch with or without them, gpurun_out/r3dma/copy_threads.txt; not kept.)
      const uint32_t rows_per_chunk = (uint32_t)(((size_t)2 << 20) / c->stride) ? (uint32_t)(((size_t)2 << 20) / c->stride) : 1u;
      for (uint32_t r0 = 0; r0 < up_n; r0 += rows_per_chunk) {
        const uint32_t nr = up_n - r0 < rows_per_chunk ? up_n - r0 : rows_per_chunk;
        const size_t off = (size_t)r0 * c->stride, len = (size_t)nr * c->stride;
        const uint8_t* from = src + off;
        if (!pinned) {
          if (from != (const uint8_t*)c->h_reqs + off) std::memcpy((uint8_t*)c->h_reqs + off, from, len);   // (a caller may hand the staging buffer itself)
          from = (const uint8_t*)c->h_reqs + off;
        }
        HIPCHK(c, hipMemcpyAsync((uint8_t*)c->d_reqs + off, from, len, hipMemcpyHostToDevice, c->stream));
      }
      if (validate_as) {    // the row headers: on the device, behind the last chunk (the host loop was a third of this path's time)
        rc = row_check_ensure(c, &c->check.h_bad, &c->check.h_bad_dev, &c->check.st, &c->check.done);
        if (rc) return rc;
        rc = rows_check_launch(c, c->d_reqs, up_n, c->check.h_bad, c->check.h_bad_dev, c->stream);
        if (rc) return rc;
        c->check.pending = true; c->check.side = false; c->check.who = validate_as;
      }
    }
  }
  if (n == 0) return EPPK_OK;
  if (cand_mask_shard && J) {
    const size_t total = (size_t)n * J * 8u, chunk = (size_t)2 << 20;
    if (cand_mask_shard == c->h_mask) {            // (eppk_pick_batch_staged: the rows are in the staging buffer already)
      HIPCHK(c, hipMemcpyAsync(c->d_mask, c->h_mask, total, hipMemcpyHostToDevice, c->stream));
    } else {
      for (size_t off = 0; off < total; off += chunk) {
        const size_t len = total - off < chunk ? total - off : chunk;
        std::memcpy((uint8_t*)c->h_mask + off, (const uint8_t*)cand_mask_shard + off, len);
        HIPCHK(c, hipMemcpyAsync((uint8_t*)c->d_mask + off, (const uint8_t*)c->h_mask + off, len, hipMemcpyHostToDevice, c->stream));
      }
    }
  }
  const uint8_t* d_shard = (const uint8_t*)c->d_reqs + (upload_all ? (size_t)lo * c->stride : 0u);
  // A single context's host entry points let the kernel write picks and scores straight into the pinned result buffers (posted PCIe
  // writes: 12 bytes per request) -- no download copies, two enqueues and two engine hand-offs less per batch; a group member keeps
  // its picks on the device (the gather reads them there).
  const bool host_out = allow_zero_copy && !upload_all;
  rc = run_pick(c, d_shard, n, ((cand_mask_shard || mask_on_device) && J) ? c->d_mask : nullptr, host_out ? c->h_pick_dev : c->d_pick + (upload_all ? lo : 0u),
                host_out ? c->h_score_dev : c->d_score + (upload_all ? lo : 0u), c->stream, 1u, false, 0ull, 0u);
  if (rc) return rc;
  if (!host_out) {
    HIPCHK(c, hipMemcpyAsync(c->h_pick, c->d_pick + (upload_all ? lo : 0u), (size_t)n * 4u, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->h_score, c->d_score + (upload_all ? lo : 0u), (size_t)n * 8u, hipMemcpyDeviceToHost, c->stream));
  }
  return EPPK_OK;
}

int pick_host_end(eppk_ctx* c, uint32_t n, bool had_mask, int32_t* out_pick, double* out_score) {
  HIPCHK(c, hipSetDevice(c->cfg.device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (c->check.pending) {
    c->check.pending = false;
    if (c->check.side) HIPCHK(c, hipEventSynchronize(c->check.done));
    if (*c->check.h_bad != 0xFFFFFFFFu) return rows_check_fail(c, c->check.who, *c->check.h_bad);
  }
  if (n == 0) return EPPK_OK;
  const size_t J = (c->n_pods + 63u) / 64u;
  std::memcpy(out_pick, c->h_pick, (size_t)n * 4u);
  if (out_score) std::memcpy(out_score, c->h_score, (size_t)n * 8u);
  if (had_mask && !J) for (uint32_t r = 0; r < n; ++r) { out_pick[r] = EPPK_NO_PICK; if (out_score) out_score[r] = 0.0; }
  return EPPK_OK;
}

}  // namespace

int eppk_pick_batch(eppk_ctx* c, const void* reqs, uint32_t n_reqs, const uint64_t* cand_mask, int32_t* out_pick, double* out_score) {
  if (!c || ((!reqs || !out_pick) && n_reqs)) return fail(c, EPPK_ERR_ARG, "eppk_pick_batch: null argument");
  if (!c->have_snapshot) return fail(c, EPPK_ERR_NO_SNAPSHOT, "eppk_pick_batch: no snapshot published");
  if (n_reqs > c->cfg.max_batch) return fail(c, EPPK_ERR_LIMIT, "eppk_pick_batch: n_reqs > max_batch");
  if (n_reqs == 0) return EPPK_OK;
  if (c->n_pods != 0u && resident_takes(c, n_reqs, cand_mask != nullptr)) {      // the latency path of a small batch: a resident workgroup, no launch
    HIPCHK(c, hipSetDevice(c->cfg.device));
    int rcr = ensure_host_staging(c, cand_mask != nullptr);
    if (rcr) return rcr;
    std::memcpy(c->h_reqs, reqs, (size_t)n_reqs * c->stride);
    if (cand_mask) std::memcpy(c->h_mask, cand_mask, (size_t)n_reqs * ((c->n_pods + 63u) / 64u) * 8u);
    return resident_pick(c, n_reqs, out_pick, out_score, "eppk_pick_batch", cand_mask != nullptr);
  }
  // (rows are validated on the host, chunk by chunk on their way to the device: never hand the kernel an out-of-range adapter / block count)
  int rc = pick_host_begin(c, (const uint8_t*)reqs, false, n_reqs, 0u, n_reqs, false, cand_mask, false, "eppk_pick_batch", true);
  if (rc) return rc;
  return pick_host_end(c, n_reqs, cand_mask != nullptr, out_pick, out_score);
}

int eppk_host_staging(eppk_ctx* c, void** reqs, uint64_t** cand_mask) {
  if (!c) return EPPK_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->cfg.device));
  int rc = ensure_host_staging(c, cand_mask != nullptr);
  if (rc) return rc;
  if (reqs) *reqs = c->h_reqs;
  if (cand_mask) *cand_mask = c->h_mask;
  return EPPK_OK;
}

int eppk_pick_batch_staged(eppk_ctx* c, uint32_t n_reqs, int use_mask, int32_t* out_pick, double* out_score) {
  if (!c || (!out_pick && n_reqs)) return fail(c, EPPK_ERR_ARG, "eppk_pick_batch_staged: null argument");
  if (!c->have_snapshot) return fail(c, EPPK_ERR_NO_SNAPSHOT, "eppk_pick_batch_staged: no snapshot published");
  if (n_reqs > c->cfg.max_batch) return fail(c, EPPK_ERR_LIMIT, "eppk_pick_batch_staged: n_reqs > max_batch");
  if (!c->h_reqs || (use_mask && !c->h_mask)) return fail(c, EPPK_ERR_ARG, "eppk_pick_batch_staged: eppk_host_staging was not called for these buffers");
  if (n_reqs == 0) return EPPK_OK;
  if (c->n_pods != 0u && resident_takes(c, n_reqs, use_mask != 0)) {
    HIPCHK(c, hipSetDevice(c->cfg.device));
    return resident_pick(c, n_reqs, out_pick, out_score, "eppk_pick_batch_staged", use_mask != 0);
  }
  int rc = pick_host_begin(c, (const uint8_t*)c->h_reqs, true, n_reqs, 0u, n_reqs, false, use_mask ? c->h_mask : nullptr, false, "eppk_pick_batch_staged", true);
  if (rc) return rc;
  return pick_host_end(c, n_reqs, use_mask != 0, out_pick, out_score);
}

// ---- the pipelined host path (include/eppk.h: eppk_pick_stage_*) ---------------------------------------------------------------

namespace {
int stage_ensure(eppk_ctx* c, uint32_t set, bool need_mask) {
  eppk_ctx::StageSet& s = c->stage[set];
  const size_t mb = c->cfg.max_batch;
  if (!s.st) {
    HIPCHK(c, hipStreamCreateWithFlags(&s.st, hipStreamNonBlocking));
    HIPCHK(c, hipEventCreateWithFlags(&s.picked, hipEventDisableTiming));
    HIPCHK(c, hipMalloc(&s.d_reqs, mb * c->stride));
    HIPCHK(c, hipMalloc((void**)&s.d_pick, (mb + EPPK_GROUP_MAX_DEVICES) * 4u));   // (+ the padding of a group's in-place all-gather)
    HIPCHK(c, hipMalloc((void**)&s.d_score, mb * 8u));
    HIPCHK(c, hipHostMalloc(&s.h_reqs, mb * c->stride, hipHostMallocDefault));
    HIPCHK(c, hipHostMalloc((void**)&s.h_pick, mb * 4u, hipHostMallocDefault));
    HIPCHK(c, hipHostMalloc((void**)&s.h_score, mb * 8u, hipHostMallocDefault));
    HIPCHK(c, hipHostGetDevicePointer(&s.h_reqs_dev, s.h_reqs, 0));
    HIPCHK(c, hipHostGetDevicePointer((void**)&s.h_pick_dev, s.h_pick, 0));
    HIPCHK(c, hipHostGetDevicePointer((void**)&s.h_score_dev, s.h_score, 0));
    c->res_args_dirty = true;                // (the resident kernels' argument blocks name these buffers)
  }
  if (need_mask && !s.d_mask) {
    HIPCHK(c, hipMalloc((void**)&s.d_mask, mb * c->jmax * 8u));
    HIPCHK(c, hipHostMalloc((void**)&s.h_mask, mb * c->jmax * 8u, hipHostMallocDefault));
    HIPCHK(c, hipHostGetDevicePointer((void**)&s.h_mask_dev, s.h_mask, 0));
    c->res_args_dirty = true;
  }
  if (!c->learned) HIPCHK(c, hipEventCreateWithFlags(&c->learned, hipEventDisableTiming));
  return EPPK_OK;
}
}  // namespace

int eppk_pick_stage_buffers(eppk_ctx* c, uint32_t set, void** reqs, uint64_t** cand_mask) {
  if (!c || set >= EPPK_STAGE_SETS) return fail(c, EPPK_ERR_ARG, "eppk_pick_stage_buffers: no such set");
  HIPCHK(c, hipSetDevice(c->cfg.device));
  int rc = stage_ensure(c, set, cand_mask != nullptr);
  if (rc) return rc;
  if (reqs) *reqs = c->stage[set].h_reqs;
  if (cand_mask) *cand_mask = c->stage[set].h_mask;
  return EPPK_OK;
}

int eppk_pick_stage_begin(eppk_ctx* c, uint32_t set, uint32_t n_reqs, int use_mask, uint32_t flags) {
  if (!c || set >= EPPK_STAGE_SETS) return fail(c, EPPK_ERR_ARG, "eppk_pick_stage_begin: no such set");
  if (!c->have_snapshot) return fail(c, EPPK_ERR_NO_SNAPSHOT, "eppk_pick_stage_begin: no snapshot published");
  if (n_reqs > c->cfg.max_batch) return fail(c, EPPK_ERR_LIMIT, "eppk_pick_stage_begin: n_reqs > max_batch");
  eppk_ctx::StageSet& s = c->stage[set];
  if (!s.st || (use_mask && !s.h_mask)) return fail(c, EPPK_ERR_ARG, "eppk_pick_stage_begin: eppk_pick_stage_buffers was not called for these buffers");
  if (s.busy) return fail(c, EPPK_ERR_ARG, "eppk_pick_stage_begin: the set is in flight (end it first)");
  if ((flags & EPPK_PICK_LEARN) && !c->slots) return fail(c, EPPK_ERR_ARG, "eppk_pick_stage_begin: EPPK_PICK_LEARN without a prefix index");
  HIPCHK(c, hipSetDevice(c->cfg.device));
  s.n = n_reqs; s.had_mask = use_mask != 0; s.busy = true; s.check_pending = false; s.resident = false;
  if (n_reqs == 0) return EPPK_OK;
  // A begin that fails has NOT begun: whatever it had enqueued on the set's stream is waited for and the set is idle again (both
  // shims treat a failed begin that way and never call end for it; a set left busy would fail every later begin).  One exception: the
  // picks were launched and only the chained LEARN update could not be -- the call then succeeds, end() delivers the picks, and
  // the missed update is reported through eppk_launch_status (EPPK_LAUNCH_LEARN_FAILED) and eppk_last_error.
  struct Abort {
    eppk_ctx::StageSet& s; bool armed = true;
    ~Abort() { if (armed) { (void)hipStreamSynchronize(s.st); if (s.st_copy) (void)hipStreamSynchronize(s.st_copy); if (s.st_check) (void)hipStreamSynchronize(s.st_check);
                            s.busy = false; s.copy_pending = false; s.check_pending = false; } }
  } abort_guard{s};
  QuietRows quiet(c);                        // rows out of range are reported by end(), naming the row (eppk_ctx::quiet_rows)
  const size_t J = (c->n_pods + 63u) / 64u;
  const bool learn = (flags & EPPK_PICK_LEARN) != 0u;
  const bool zero_copy = n_reqs <= c->zero_copy_max;
  int rc;
  if (c->n_pods != 0u && resident_takes(c, n_reqs, use_mask != 0, 1u, learn)) {
    // the latency path of a small batch (EPPK_RESIDENT=1): rung into a resident workgroup, which reads the set's pinned rows (and mask
    // rows) and writes its pinned results; end() polls the completion word -- no launch, no event.  With LEARN the workgroup applies
    // the post-route index update itself, right behind the answer (it has copied the rows: the caller may refill the set after end()).
    rc = validate_rows(c, "eppk_pick_stage_begin", s.h_reqs, n_reqs, 0u);
    if (rc) return rc;
    rc = resident_ring(c, n_reqs, use_mask != 0, 1u, 1u + set, &s.res_unit, &s.res_seq, learn);
    if (rc) return rc;
    s.resident = true;
    abort_guard.armed = false;
    return EPPK_OK;
  }
  bool words = false;                        // the pick kernel left learn words for the update (pick_quad_kernel<..., LEARN>): known pairs are
  if (learn) { rc = learn_ensure(c, n_reqs); if (rc) return rc; }     // skipped, and the picks come out of those words instead of pinned host memory
  // (learn_words_fence goes in right in front of the pick: NOT ahead of the upload, which must not wait for the other set's update --
  // the point of the two sets is that the rows of batch k + 1 cross PCIe while batch k is scored and learned)
  if (zero_copy) {
    // ZERO-COPY (a small batch, as eppk_pick_batch_staged does it): one launch that reads the pinned set and writes its result buffers.
    // With LEARN the index update runs on behind `picked`, and the caller may refill the set as soon as _end has returned: the update
    // reads a DEVICE copy of the rows, uploaded on a stream of its own beside the pick (_end waits for that upload too), and the picks
    // out of the pinned result buffer (which only this set's next pick writes, and that one is ordered behind the update).
    const bool dev_check = n_reqs > c->host_check_max;
    rc = dev_check ? row_check_ensure(c, &s.h_bad, &s.h_bad_dev, &s.st_check, &s.checked) : validate_rows(c, "eppk_pick_stage_begin", s.h_reqs, n_reqs, 0u);
    if (rc) return rc;
    if (dev_check) {        // beside the pick, on a stream of its own (eppk_ctx::RowCheck)
      rc = rows_check_launch(c, s.h_reqs_dev, n_reqs, s.h_bad, s.h_bad_dev, s.st_check);
      if (rc) return rc;
      HIPCHK(c, hipEventRecord(s.checked, s.st_check));
      s.check_pending = true; s.check_side = true;
    }
    if (learn) {
      if (!s.st_copy) {
        HIPCHK(c, hipStreamCreateWithFlags(&s.st_copy, hipStreamNonBlocking));
        HIPCHK(c, hipEventCreateWithFlags(&s.copied, hipEventDisableTiming));
      }
      rc = learn_fence(c, s.st_copy);          // (an earlier update may still read s.d_reqs)
      if (rc) return rc;
      HIPCHK(c, hipMemcpyAsync(s.d_reqs, s.h_reqs, (size_t)n_reqs * c->stride, hipMemcpyHostToDevice, s.st_copy));
      HIPCHK(c, hipEventRecord(s.copied, s.st_copy));
      s.copy_pending = true;
    }
    if (learn) { rc = learn_words_fence(c, s.st); if (rc) return rc; }
    rc = run_pick(c, (const uint8_t*)s.h_reqs_dev, n_reqs, (use_mask && J) ? s.h_mask_dev : nullptr, s.h_pick_dev, s.h_score_dev, s.st, 1u, false, 0ull, 0u,
                  learn ? c->d_learn : nullptr, &words);
    if (rc) return rc;
    HIPCHK(c, hipEventRecord(s.picked, s.st));
    if (learn) HIPCHK(c, hipStreamWaitEvent(s.st, s.copied, 0));
  } else {
    // one upload, its row headers checked on the device (as eppk_pick_batch_staged does: eppk_ctx::RowCheck)
    HIPCHK(c, hipMemcpyAsync(s.d_reqs, s.h_reqs, (size_t)n_reqs * c->stride, hipMemcpyHostToDevice, s.st));
    rc = row_check_ensure(c, &s.h_bad, &s.h_bad_dev, &s.st_check, &s.checked);
    if (!rc) rc = rows_check_launch(c, s.d_reqs, n_reqs, s.h_bad, s.h_bad_dev, s.st);      // behind the upload, in front of the pick
    if (rc) return rc;
    s.check_pending = true; s.check_side = false;
    if (use_mask && J) HIPCHK(c, hipMemcpyAsync(s.d_mask, s.h_mask, (size_t)n_reqs * J * 8u, hipMemcpyHostToDevice, s.st));
    // the pick sees the index every earlier LEARN left behind (run_pick: learn_fence; the upload above did not have to wait for it).
    // Picks and scores: written by the kernel straight into the set's pinned result buffers -- no download copies; a LEARN update reads
    // the picks from there, and only this set's next pick, ordered behind that update, writes them again.
    if (learn) { rc = learn_words_fence(c, s.st); if (rc) return rc; }
    rc = run_pick(c, (const uint8_t*)s.d_reqs, n_reqs, (use_mask && J) ? s.d_mask : nullptr, s.h_pick_dev, s.h_score_dev, s.st, 1u, false, 0ull, 0u,
                  learn ? c->d_learn : nullptr, &words);
    if (rc) return rc;
    HIPCHK(c, hipEventRecord(s.picked, s.st));
  }
  abort_guard.armed = false;                 // the picks are on their way: from here on the call succeeds and end() delivers them
  if (learn) {
    // the post-route update, chained on the device: rows and picks are there already (eppk_pick_learn_device's second half)
    rc = learn_picks(c, s.d_reqs, s.h_pick_dev, n_reqs, s.st, words ? c->d_learn : nullptr);
    if (rc == EPPK_OK && hipEventRecord(c->learned, s.st) == hipSuccess) c->learn_pending = true;
    else c->host_flags |= EPPK_LAUNCH_LEARN_FAILED;        // (eppk_last_error holds the reason; the picks stand)
  }
  return EPPK_OK;
}

int eppk_pick_stage_end(eppk_ctx* c, uint32_t set, int32_t* out_pick, double* out_score) {
  if (!c || set >= EPPK_STAGE_SETS) return fail(c, EPPK_ERR_ARG, "eppk_pick_stage_end: no such set");
  eppk_ctx::StageSet& s = c->stage[set];
  if (!s.busy) return fail(c, EPPK_ERR_ARG, "eppk_pick_stage_end: the set is not in flight");
  if (!out_pick && s.n) return fail(c, EPPK_ERR_ARG, "eppk_pick_stage_end: null argument");
  HIPCHK(c, hipSetDevice(c->cfg.device));
  s.busy = false;
  if (s.n == 0) return EPPK_OK;
  if (s.resident) {
    s.resident = false;
    const int rcw = resident_wait(c, s.res_unit, s.res_seq, "eppk_pick_stage_end");
    if (rcw) return rcw;
    std::memcpy(out_pick, s.h_pick, (size_t)s.n * 4u);
    if (out_score) std::memcpy(out_score, s.h_score, (size_t)s.n * 8u);
    return EPPK_OK;
  }
  HIPCHK(c, hipEventSynchronize(s.picked));
  if (s.copy_pending) { HIPCHK(c, hipEventSynchronize(s.copied)); s.copy_pending = false; }   // (the caller may refill the rows now)
  if (s.check_pending) {
    s.check_pending = false;
    if (s.check_side) HIPCHK(c, hipEventSynchronize(s.checked));
    if (*s.h_bad != 0xFFFFFFFFu) return rows_check_fail(c, "eppk_pick_stage_end", *s.h_bad);
  }
  const size_t J = (c->n_pods + 63u) / 64u;
  std::memcpy(out_pick, s.h_pick, (size_t)s.n * 4u);
  if (out_score) std::memcpy(out_score, s.h_score, (size_t)s.n * 8u);
  if (s.had_mask && !J) for (uint32_t r = 0; r < s.n; ++r) { out_pick[r] = EPPK_NO_PICK; if (out_score) out_score[r] = 0.0; }
  return EPPK_OK;
}

// ---- candidate-major pick (masked batches with few candidates) --------------------------------------------

namespace {
// SINGLE picks on masks of a few candidates: from 8192 requests on the general masked route is the faster one wherever pick_quad_kernel
// serves the batch -- it parks every such row and scores four at a time (quad_exact_rows_par): 16k x 8 candidates 27 us against 55,
// 64k 71 against 210; below, the candidate-major kernel (4096: 18 against 20, 1024: 13 against 26; scripts/subset_route_probe.py).
// Ordered fallbacks (k > 1, parked and scored in k rounds): from 4096 requests on -- top-3 of 8 candidates, 4096: 35 us against 43, 16k: 45
// against 141, 64k: 122 against 559.
bool general_route_is_faster(eppk_ctx* c, uint32_t n_reqs, uint32_t k = 1u) {
  return n_reqs >= (k == 1u ? 8192u : 4096u) && n_reqs >= c->quad_min && c->canonical && c->quad_on && c->quad_backoff == 0u && c->has_p && c->npl == 6 && !c->gen && c->pterm &&
         c->slots != 0u && c->cfg.max_blocks >= 1 && make_kindex(c).lists != nullptr;
}
int launch_pick_cands(eppk_ctx* c, const void* d_reqs, uint32_t n_reqs, const uint64_t* d_mask, uint32_t k, int32_t* d_pick, double* d_score, hipStream_t st) {
  KSnap sn = make_ksnap(c);
  KIndex ix = make_kindex(c);
  KChain ch = c->kchain;
  bool any_prefix = false;                                         // (c->has_p is only meaningful for chains the fast kernel serves)
  for (uint32_t i = 0; i < ch.n; ++i) any_prefix |= ch.kind[i] == 4u;
  if (!any_prefix) ix.slots = 0u;                                  // no PREFIX scorer: nothing to look up
  const uint8_t* reqs8 = (const uint8_t*)d_reqs;
  uint32_t stride = c->stride;
  uint32_t grid = (n_reqs + 3u) / 4u;
  const uint32_t cap = (uint32_t)c->num_cu * 8u;
  if (grid > cap) grid = cap;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (c->prof) {
    if (c->ev_used + 2 > c->ev.size()) {
      hipEvent_t a, b;
      HIPCHK(c, hipEventCreate(&a));
      HIPCHK(c, hipEventCreate(&b));
      c->ev.push_back(a);
      c->ev.push_back(b);
    }
    e0 = c->ev[c->ev_used];
    e1 = c->ev[c->ev_used + 1];
    c->ev_used += 2;
  }
  const void* fn = nullptr;
  (void)by_lane_word(c, [&](auto tag) { using LW = decltype(tag); fn = (const void*)eppk::pick_cands_kernel<LW>; return EPPK_OK; });
  void* args[] = {&sn, &ix, &ch, &reqs8, &stride, &n_reqs, &d_mask, &d_pick, &d_score, &k};
  HIPCHK(c, hipExtLaunchKernel(fn, dim3(grid), dim3(256), args, 0, st, e0, e1, 0));
  c->last_done = e1;
  c->last_stream = st;
  if (c->prof) c->launches++;
  return EPPK_OK;
}
}  // namespace

int eppk_pick_batch_candidates_device(eppk_ctx* c, const void* d_reqs, uint32_t n_reqs, const uint64_t* d_cand_mask, uint32_t k,
                                      int32_t* d_out_pick, double* d_out_score, void* stream) {
  if (!c || ((!d_reqs || !d_out_pick || !d_cand_mask) && n_reqs)) return fail(c, EPPK_ERR_ARG, "eppk_pick_batch_candidates_device: null argument");
  if (k < 1 || k > EPPK_MAX_TOPK) return fail(c, EPPK_ERR_ARG, "eppk_pick_batch_candidates_device: k out of range (1..8)");
  if (!c->have_snapshot) return fail(c, EPPK_ERR_NO_SNAPSHOT, "eppk_pick_batch_candidates_device: no snapshot published");
  if (n_reqs == 0) return EPPK_OK;
  HIPCHK(c, hipSetDevice(c->cfg.device));
  hipStream_t st = stream ? (hipStream_t)stream : c->stream;
  { const int rcf = learn_fence(c, st); if (rcf) return rcf; }
  if (c->assumed_epochs)                                           // assumed load works in epochs of the general path (SEMANTICS.md §2b)
    return k == 1 ? eppk_pick_batch_device(c, d_reqs, n_reqs, d_cand_mask, d_out_pick, d_out_score, stream)
                  : eppk_pick_topk_device(c, d_reqs, n_reqs, d_cand_mask, k, d_out_pick, d_out_score, stream);
  if (general_route_is_faster(c, n_reqs, k))
    return k == 1 ? eppk_pick_batch_device(c, d_reqs, n_reqs, d_cand_mask, d_out_pick, d_out_score, stream)
                  : eppk_pick_topk_device(c, d_reqs, n_reqs, d_cand_mask, k, d_out_pick, d_out_score, stream);
  return launch_pick_cands(c, d_reqs, n_reqs, d_cand_mask, k, d_out_pick, d_out_score, st);
}

// ---- subset filter on the device ------------------------------------------------------------------------

int eppk_snapshot_set_addresses(eppk_ctx* c, const char* const* addrs, const char* const* ports, uint32_t n_pods) {
  if (!c || ((!addrs || !ports) && n_pods)) return fail(c, EPPK_ERR_ARG, "eppk_snapshot_set_addresses: null argument");
  if (!c->have_snapshot) return fail(c, EPPK_ERR_NO_SNAPSHOT, "eppk_snapshot_set_addresses: no snapshot published");
  if (n_pods != c->n_pods) return fail(c, EPPK_ERR_ARG, "eppk_snapshot_set_addresses: n_pods differs from the published snapshot");
  HIPCHK(c, hipSetDevice(c->cfg.device));
  if (!c->d_at) {
    uint32_t t = 64;
    while (t < 8u * c->cfg.max_pods) t *= 2u;                      // two entries per pod, at most a quarter full
    c->at_slots = t;
    HIPCHK(c, hipMalloc((void**)&c->d_at, (size_t)t * 16u));
    HIPCHK(c, hipMalloc((void**)&c->d_av, (size_t)t * 4u));
  }
  const uint32_t T = c->at_slots;
  std::vector<uint64_t> at((size_t)T * 2u, 0ull);
  std::vector<uint32_t> av(T, 0u);
  auto put = [&](const uint64_t fp[2], uint32_t pod) {
    uint32_t t = (uint32_t)fp[0] & (T - 1u);
    while (av[t] != 0u) t = (t + 1u) & (T - 1u);
    at[2 * (size_t)t] = fp[0]; at[2 * (size_t)t + 1] = fp[1]; av[t] = pod + 1u;
  };
  for (uint32_t p = 0; p < n_pods; ++p) {
    if (!addrs[p]) continue;                                       // a hole of the snapshot: matches nothing
    if (!ports[p]) return fail(c, EPPK_ERR_ARG, "eppk_snapshot_set_addresses: pod " + std::to_string(p) + " has an address but no port");
    uint64_t fp[2];
    eppk_addr_fingerprint(addrs[p], std::strlen(addrs[p]), nullptr, 0, fp);
    put(fp, p);
    eppk_addr_fingerprint(addrs[p], std::strlen(addrs[p]), ports[p], std::strlen(ports[p]), fp);
    put(fp, p);
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));                      // (a subset kernel of an earlier batch may still read the table)
  HIPCHK(c, hipMemcpy(c->d_at, at.data(), at.size() * 8u, hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(c->d_av, av.data(), av.size() * 4u, hipMemcpyHostToDevice));
  c->have_addrs = true;
  c->addr_n = n_pods;
  return EPPK_OK;
}

int eppk_subset_masks_device(eppk_ctx* c, const uint64_t* d_keys, const uint32_t* d_off, uint32_t n_reqs, uint64_t* d_mask_out, void* stream) {
  if (!c || ((!d_off || !d_mask_out) && n_reqs)) return fail(c, EPPK_ERR_ARG, "eppk_subset_masks_device: null argument");
  if (!c->have_snapshot) return fail(c, EPPK_ERR_NO_SNAPSHOT, "eppk_subset_masks_device: no snapshot published");
  if (!c->have_addrs || c->addr_n != c->n_pods)
    return fail(c, EPPK_ERR_NO_SNAPSHOT, "eppk_subset_masks_device: eppk_snapshot_set_addresses has not been called for the current snapshot");
  if (n_reqs == 0 || c->n_pods == 0) return EPPK_OK;
  HIPCHK(c, hipSetDevice(c->cfg.device));
  hipStream_t st = stream ? (hipStream_t)stream : c->stream;
  uint32_t grid = (n_reqs + 3u) / 4u;
  const uint32_t cap = (uint32_t)c->num_cu * 8u;
  if (grid > cap) grid = cap;
  hipLaunchKernelGGL(eppk::subset_masks_kernel, dim3(grid), dim3(256), 0, st, (const uint64_t*)c->d_at, (const uint32_t*)c->d_av, c->at_slots - 1u,
                     d_keys, d_off, n_reqs, c->n_pods, d_mask_out);
  HIPCHK(c, hipGetLastError());
  return EPPK_OK;
}

namespace {
// entries of a batch (host, CSR) -> device staging; returns the device pointers
int stage_subset_entries(eppk_ctx* c, const char* who, const uint64_t* keys, const uint32_t* off, uint32_t n_reqs) {
  for (uint32_t r = 0; r < n_reqs; ++r)
    if (off[r + 1] < off[r]) return fail(c, EPPK_ERR_ARG, std::string(who) + ": off[] must be non-decreasing");
  const size_t n_keys = off[n_reqs];
  if (n_keys && !keys) return fail(c, EPPK_ERR_ARG, std::string(who) + ": null keys");
  if (n_keys > c->sk_cap) {
    (void)hipFree(c->d_sk); c->d_sk = nullptr; c->sk_cap = 0;
    size_t cap = 1024; while (cap < n_keys) cap *= 2;
    HIPCHK(c, hipMalloc((void**)&c->d_sk, cap * 16u));
    c->sk_cap = cap;
  }
  if ((size_t)n_reqs + 1u > c->so_cap) {
    (void)hipFree(c->d_so); c->d_so = nullptr; c->so_cap = 0;
    HIPCHK(c, hipMalloc((void**)&c->d_so, ((size_t)c->cfg.max_batch + 1u) * 4u));
    c->so_cap = (size_t)c->cfg.max_batch + 1u;
  }
  if (n_keys) HIPCHK(c, hipMemcpyAsync(c->d_sk, keys, n_keys * 16u, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->d_so, off, ((size_t)n_reqs + 1u) * 4u, hipMemcpyHostToDevice, c->stream));
  return EPPK_OK;
}
}  // namespace

int eppk_subset_masks(eppk_ctx* c, const uint64_t* keys, const uint32_t* off, uint32_t n_reqs, uint64_t* out_mask) {
  if (!c || ((!off || !out_mask) && n_reqs)) return fail(c, EPPK_ERR_ARG, "eppk_subset_masks: null argument");
  if (n_reqs > c->cfg.max_batch) return fail(c, EPPK_ERR_LIMIT, "eppk_subset_masks: n_reqs > max_batch");
  if (!c->have_snapshot) return fail(c, EPPK_ERR_NO_SNAPSHOT, "eppk_subset_masks: no snapshot published");
  if (n_reqs == 0 || c->n_pods == 0) return EPPK_OK;
  HIPCHK(c, hipSetDevice(c->cfg.device));
  int rc = ensure_host_staging(c, true);
  if (rc) return rc;
  rc = stage_subset_entries(c, "eppk_subset_masks", keys, off, n_reqs);
  if (rc) return rc;
  rc = eppk_subset_masks_device(c, c->d_sk, c->d_so, n_reqs, c->d_mask, c->stream);
  if (rc) return rc;
  const size_t J = (c->n_pods + 63u) / 64u;
  HIPCHK(c, hipMemcpyAsync(out_mask, c->d_mask, (size_t)n_reqs * J * 8u, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return EPPK_OK;
}

int eppk_pick_batch_subset(eppk_ctx* c, const void* reqs, uint32_t n_reqs, const uint64_t* keys, const uint32_t* off,
                           int32_t* out_pick, double* out_score) {
  if (!c || ((!reqs || !out_pick || !off) && n_reqs)) return fail(c, EPPK_ERR_ARG, "eppk_pick_batch_subset: null argument");
  if (!c->have_snapshot) return fail(c, EPPK_ERR_NO_SNAPSHOT, "eppk_pick_batch_subset: no snapshot published");
  if (n_reqs > c->cfg.max_batch) return fail(c, EPPK_ERR_LIMIT, "eppk_pick_batch_subset: n_reqs > max_batch");
  if (n_reqs == 0) return EPPK_OK;
  int rc = validate_rows(c, "eppk_pick_batch_subset", reqs, n_reqs);
  if (rc) return rc;
  HIPCHK(c, hipSetDevice(c->cfg.device));
  rc = ensure_host_staging(c, true);
  if (rc) return rc;
  if (c->n_pods) {
    rc = stage_subset_entries(c, "eppk_pick_batch_subset", keys, off, n_reqs);
    if (rc) return rc;
    rc = eppk_subset_masks_device(c, c->d_sk, c->d_so, n_reqs, c->d_mask, c->stream);
    if (rc) return rc;
  }
  // few entries per request (what a subset hint looks like): the candidate-major kernel, O(candidates) per request
  // (large batches: the general route parks and scores such rows itself, faster -- general_route_is_faster)
  const bool few = c->n_pods != 0u && c->assumed_epochs == 0u && (uint64_t)off[n_reqs] <= 32ull * n_reqs && !general_route_is_faster(c, n_reqs);
  if (!few) {
    rc = pick_host_begin(c, (const uint8_t*)reqs, false, n_reqs, 0u, n_reqs, false, nullptr, true);
    if (rc) return rc;
    return pick_host_end(c, n_reqs, true, out_pick, out_score);
  }
  std::memcpy(c->h_reqs, reqs, (size_t)n_reqs * c->stride);
  HIPCHK(c, hipMemcpyAsync(c->d_reqs, c->h_reqs, (size_t)n_reqs * c->stride, hipMemcpyHostToDevice, c->stream));
  rc = launch_pick_cands(c, c->d_reqs, n_reqs, c->d_mask, 1u, c->d_pick, c->d_score, c->stream);
  if (rc) return rc;
  HIPCHK(c, hipMemcpyAsync(c->h_pick, c->d_pick, (size_t)n_reqs * 4u, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->h_score, c->d_score, (size_t)n_reqs * 8u, hipMemcpyDeviceToHost, c->stream));
  return pick_host_end(c, n_reqs, false, out_pick, out_score);
}

// ---- ordered fallbacks ---------------------------------------------------------------------------------

namespace {
int topk_ensure(eppk_ctx* c, bool need_mask) {
  const size_t mb = c->cfg.max_batch;
  if (!c->d_tk_reqs) {
    HIPCHK(c, hipMalloc(&c->d_tk_reqs, mb * c->stride));
    HIPCHK(c, hipMalloc((void**)&c->d_tk_pick, mb * EPPK_MAX_TOPK * 4u));
    HIPCHK(c, hipMalloc((void**)&c->d_tk_score, mb * EPPK_MAX_TOPK * 8u));
  }
  if (need_mask && !c->d_tk_mask) HIPCHK(c, hipMalloc((void**)&c->d_tk_mask, mb * c->jmax * 8u));
  return EPPK_OK;
}
}  // namespace

int eppk_pick_topk_device(eppk_ctx* c, const void* d_reqs, uint32_t n_reqs, const uint64_t* d_cand_mask, uint32_t k,
                          int32_t* d_out_pick, double* d_out_score, void* stream) {
  if (!c || ((!d_reqs || !d_out_pick) && n_reqs)) return fail(c, EPPK_ERR_ARG, "eppk_pick_topk_device: null argument");
  if (k < 1 || k > EPPK_MAX_TOPK) return fail(c, EPPK_ERR_ARG, "eppk_pick_topk_device: k out of range (1..8)");
  if (!c->have_snapshot) return fail(c, EPPK_ERR_NO_SNAPSHOT, "eppk_pick_topk_device: no snapshot published");
  if (n_reqs == 0) return EPPK_OK;
  HIPCHK(c, hipSetDevice(c->cfg.device));
  hipStream_t st = stream ? (hipStream_t)stream : c->stream;
  const uint32_t per = (uint32_t)((1ull << 31) / c->stride);
  const size_t J = (c->n_pods + 63u) / 64u;
  for (uint32_t r0 = 0; r0 < n_reqs; r0 += per) {
    const uint32_t n = (n_reqs - r0 < per) ? n_reqs - r0 : per;
    int rc = run_pick(c, (const uint8_t*)d_reqs + (size_t)r0 * c->stride, n, d_cand_mask ? d_cand_mask + (size_t)r0 * J : nullptr,
                      d_out_pick + (size_t)r0 * k, d_out_score ? d_out_score + (size_t)r0 * k : nullptr, st, k, false, 0ull, 0u);
    if (rc) return rc;
  }
  return EPPK_OK;
}

int eppk_pick_topk(eppk_ctx* c, const void* reqs, uint32_t n_reqs, const uint64_t* cand_mask, uint32_t k, int32_t* out_pick,
                   double* out_score) {
  if (!c || ((!reqs || !out_pick) && n_reqs)) return fail(c, EPPK_ERR_ARG, "eppk_pick_topk: null argument");
  if (k < 1 || k > EPPK_MAX_TOPK) return fail(c, EPPK_ERR_ARG, "eppk_pick_topk: k out of range (1..8)");
  if (!c->have_snapshot) return fail(c, EPPK_ERR_NO_SNAPSHOT, "eppk_pick_topk: no snapshot published");
  if (n_reqs > c->cfg.max_batch) return fail(c, EPPK_ERR_LIMIT, "eppk_pick_topk: n_reqs > max_batch");
  if (n_reqs == 0) return EPPK_OK;
  for (uint32_t r = 0; r < n_reqs; ++r) {
    eppk_req_hdr hd;
    std::memcpy(&hd, (const uint8_t*)reqs + (size_t)r * c->stride, sizeof hd);
    if (hd.n_blocks > c->cfg.max_blocks || hd.adapter < -1 || hd.adapter >= (int32_t)EPPK_MAX_ADAPTERS)
      return fail(c, EPPK_ERR_ARG, "eppk_pick_topk: request row " + std::to_string(r) + " out of range");
  }
  HIPCHK(c, hipSetDevice(c->cfg.device));
  const size_t J = (c->n_pods + 63u) / 64u;
  if (cand_mask && !J) {
    for (size_t i = 0; i < (size_t)n_reqs * k; ++i) { out_pick[i] = EPPK_NO_PICK; if (out_score) out_score[i] = 0.0; }
    return EPPK_OK;
  }
  const size_t mb = c->cfg.max_batch;
  if (n_reqs <= c->zero_copy_max && (size_t)n_reqs * k <= mb) {
    // ZERO-COPY, as the single-pick entry points do it for small batches (a dispatcher that asks for fallback lists sends EVERY batch
    // through here): rows (and mask) into the pinned staging buffers, one launch that reads them and writes the n x k lists into the
    // pinned result buffers (max_batch entries each: n x k fits), no upload, no download.
    int rc = ensure_host_staging(c, cand_mask != nullptr);
    if (rc) return rc;
    if (reqs != c->h_reqs) std::memcpy(c->h_reqs, reqs, (size_t)n_reqs * c->stride);
    if (cand_mask && cand_mask != c->h_mask) std::memcpy(c->h_mask, cand_mask, (size_t)n_reqs * J * 8u);
    // the latency path of a dispatcher that asks for fallback lists (every one of its batches comes through here): a resident workgroup
    if (resident_takes(c, n_reqs, cand_mask != nullptr, k)) return resident_pick(c, n_reqs, out_pick, out_score, "eppk_pick_topk", cand_mask != nullptr, k);
    rc = run_pick(c, (const uint8_t*)c->h_reqs_dev, n_reqs, cand_mask ? c->h_mask_dev : nullptr, c->h_pick_dev, c->h_score_dev, c->stream, k, false, 0ull, 0u);
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    std::memcpy(out_pick, c->h_pick, (size_t)n_reqs * k * 4u);
    if (out_score) std::memcpy(out_score, c->h_score, (size_t)n_reqs * k * 8u);
    return EPPK_OK;
  }
  // device buffers are kept in the context (allocated on first use); transfers are plain pageable copies
  { const int rct = topk_ensure(c, cand_mask != nullptr); if (rct) return rct; }
  HIPCHK(c, hipMemcpyAsync(c->d_tk_reqs, reqs, (size_t)n_reqs * c->stride, hipMemcpyHostToDevice, c->stream));
  if (cand_mask) HIPCHK(c, hipMemcpyAsync(c->d_tk_mask, cand_mask, (size_t)n_reqs * J * 8u, hipMemcpyHostToDevice, c->stream));
  int rc = run_pick(c, (const uint8_t*)c->d_tk_reqs, n_reqs, cand_mask ? c->d_tk_mask : nullptr, c->d_tk_pick, c->d_tk_score, c->stream, k, false, 0ull, 0u);
  if (rc) return rc;
  HIPCHK(c, hipMemcpyAsync(out_pick, c->d_tk_pick, (size_t)n_reqs * k * 4u, hipMemcpyDeviceToHost, c->stream));
  if (out_score) HIPCHK(c, hipMemcpyAsync(out_score, c->d_tk_score, (size_t)n_reqs * k * 8u, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return EPPK_OK;
}

// ---- picker "random-top-k" (SEMANTICS.md §3b) and assumed load (§2b) --------------------------------------------------

int eppk_pick_random_topk_device(eppk_ctx* c, const void* d_reqs, uint32_t n_reqs, const uint64_t* d_cand_mask, uint32_t k, uint64_t seed,
                                 int32_t* d_out_pick, double* d_out_score, void* stream) {
  if (!c || ((!d_reqs || !d_out_pick) && n_reqs)) return fail(c, EPPK_ERR_ARG, "eppk_pick_random_topk_device: null argument");
  if (k < 1 || k > EPPK_MAX_TOPK) return fail(c, EPPK_ERR_ARG, "eppk_pick_random_topk_device: k out of range (1..8)");
  if (!c->have_snapshot) return fail(c, EPPK_ERR_NO_SNAPSHOT, "eppk_pick_random_topk_device: no snapshot published");
  if (n_reqs == 0) return EPPK_OK;
  if ((uint64_t)n_reqs * c->stride >= (1ull << 31)) return fail(c, EPPK_ERR_LIMIT, "eppk_pick_random_topk_device: batch of 2 GiB and more (split it)");
  HIPCHK(c, hipSetDevice(c->cfg.device));
  hipStream_t st = stream ? (hipStream_t)stream : c->stream;
  return run_pick(c, (const uint8_t*)d_reqs, n_reqs, d_cand_mask, d_out_pick, d_out_score, st, k, true, seed, 0u);
}

int eppk_pick_random_topk(eppk_ctx* c, const void* reqs, uint32_t n_reqs, const uint64_t* cand_mask, uint32_t k, uint64_t seed, int32_t* out_pick,
                          double* out_score) {
  if (!c || ((!reqs || !out_pick) && n_reqs)) return fail(c, EPPK_ERR_ARG, "eppk_pick_random_topk: null argument");
  if (k < 1 || k > EPPK_MAX_TOPK) return fail(c, EPPK_ERR_ARG, "eppk_pick_random_topk: k out of range (1..8)");
  if (!c->have_snapshot) return fail(c, EPPK_ERR_NO_SNAPSHOT, "eppk_pick_random_topk: no snapshot published");
  if (n_reqs > c->cfg.max_batch) return fail(c, EPPK_ERR_LIMIT, "eppk_pick_random_topk: n_reqs > max_batch");
  if (n_reqs == 0) return EPPK_OK;
  int rc = validate_rows(c, "eppk_pick_random_topk", reqs, n_reqs);
  if (rc) return rc;
  HIPCHK(c, hipSetDevice(c->cfg.device));
  const size_t J = (c->n_pods + 63u) / 64u;
  if (cand_mask && !J) {
    for (uint32_t r = 0; r < n_reqs; ++r) { out_pick[r] = EPPK_NO_PICK; if (out_score) out_score[r] = 0.0; }
    return EPPK_OK;
  }
  rc = ensure_host_staging(c, cand_mask != nullptr);
  if (rc) return rc;
  if (reqs != c->h_reqs) std::memcpy(c->h_reqs, reqs, (size_t)n_reqs * c->stride);
  if (cand_mask && cand_mask != c->h_mask) std::memcpy(c->h_mask, cand_mask, (size_t)n_reqs * J * 8u);
  if (n_reqs <= c->zero_copy_max) {          // zero-copy (small batch): the kernels read the pinned rows and write the pinned results
    rc = run_pick(c, (const uint8_t*)c->h_reqs_dev, n_reqs, cand_mask ? c->h_mask_dev : nullptr, c->h_pick_dev, c->h_score_dev, c->stream, k, true, seed, 0u);
    if (rc) return rc;
  } else {                                   // one upload; the results still land in the pinned buffers (no download copies)
    HIPCHK(c, hipMemcpyAsync(c->d_reqs, c->h_reqs, (size_t)n_reqs * c->stride, hipMemcpyHostToDevice, c->stream));
    if (cand_mask) HIPCHK(c, hipMemcpyAsync(c->d_mask, c->h_mask, (size_t)n_reqs * J * 8u, hipMemcpyHostToDevice, c->stream));
    rc = run_pick(c, (const uint8_t*)c->d_reqs, n_reqs, cand_mask ? c->d_mask : nullptr, c->h_pick_dev, c->h_score_dev, c->stream, k, true, seed, 0u);
    if (rc) return rc;
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  std::memcpy(out_pick, c->h_pick, (size_t)n_reqs * 4u);
  if (out_score) std::memcpy(out_score, c->h_score, (size_t)n_reqs * 8u);
  return EPPK_OK;
}

int eppk_set_assumed_load(eppk_ctx* c, uint32_t epochs) {
  if (!c) return EPPK_ERR_ARG;
  if (epochs > 65536u) return fail(c, EPPK_ERR_LIMIT, "eppk_set_assumed_load: more than 65536 epochs per batch");
  c->assumed_epochs = epochs;
  return EPPK_OK;
}

// ---- on-device prompt hashing ------------------------------------------------------------------------

int eppk_hash_prompts_device(eppk_ctx* c, const void* d_prompts, uint64_t prompt_stride, const uint32_t* d_prompt_len,
                             const uint64_t* d_seed, const int32_t* d_adapter, uint32_t n_reqs, uint32_t block_chars,
                             void* d_reqs_out, void* stream) {
  if (!c || ((!d_prompts || !d_prompt_len || !d_seed || !d_adapter || !d_reqs_out) && n_reqs))
    return fail(c, EPPK_ERR_ARG, "eppk_hash_prompts_device: null argument");
  if (block_chars == 0 || (block_chars & 7u) || (prompt_stride & 7u) || ((uintptr_t)d_prompts & 7u))
    return fail(c, EPPK_ERR_ARG, "eppk_hash_prompts_device: block_chars, prompt_stride and the prompt base must be multiples of 8 (use eppk_hash_prompt on the host otherwise)");
  if (n_reqs == 0) return EPPK_OK;
  HIPCHK(c, hipSetDevice(c->cfg.device));
  hipStream_t st = stream ? (hipStream_t)stream : c->stream;
  const uint32_t threads = 64, grid = (n_reqs + threads - 1) / threads;
  hipLaunchKernelGGL(hash_prompts_kernel, dim3(grid), dim3(threads), 0, st, (const uint8_t*)d_prompts, prompt_stride, d_prompt_len, d_seed,
                     d_adapter, n_reqs, block_chars, c->cfg.max_blocks, (uint8_t*)d_reqs_out, c->stride);
  HIPCHK(c, hipGetLastError());
  return EPPK_OK;
}

int eppk_launch_status(eppk_ctx* c, uint32_t* flags) {
  if (!c || !flags) return EPPK_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->cfg.device));
  { const int rcs_ = device_sync(c); if (rcs_) return rcs_; }     // every launch of this context, on whatever stream the caller used
  uint32_t f = c->host_flags;
  c->host_flags = 0;
  // sticky words: d_status[0] and the last dword of either snapshot blob.  The words of the host-buffer launches (d_status[1], the
  // dword 8 bytes earlier in the blobs: eppk_ctx::quiet_rows) only ever matter for what is NOT a row error -- those calls name the row
  // themselves -- and are cleared along.
  struct W { uint32_t* p; uint32_t keep; };
  const W words[6] = {{c->d_status, ~0u}, {c->d_status + 1, ~EPPK_LAUNCH_BAD_REQUEST_ROW},
                      {(uint32_t*)(c->snap[0].blob + c->lay.bytes - kBlobStatusTail), ~0u}, {(uint32_t*)(c->snap[1].blob + c->lay.bytes - kBlobStatusTail), ~0u},
                      {(uint32_t*)(c->snap[0].blob + c->lay.bytes - kBlobStatusTail - 8u), 0u}, {(uint32_t*)(c->snap[1].blob + c->lay.bytes - kBlobStatusTail - 8u), 0u}};
  for (const W& w : words) {
    uint32_t v = 0;
    HIPCHK(c, hipMemcpy(&v, w.p, sizeof v, hipMemcpyDeviceToHost));
    if (v) HIPCHK(c, hipMemset(w.p, 0, sizeof v));
    f |= v & w.keep;
  }
  *flags = f;
  return EPPK_OK;
}

// ---- measurement -----------------------------------------------------------------------------------

int eppk_chain_is_fused(const eppk_ctx* c) {
  if (!c) return EPPK_ERR_ARG;
  return c->canonical ? (c->gen ? 2 : 1) : 0;
}

int eppk_resident_stats(const eppk_ctx* c, uint64_t* batches, uint64_t* starts) {
  if (!c) return EPPK_ERR_ARG;
  if (batches) *batches = c->res_batches;
  if (starts) *starts = c->res_starts;
  return c->resident_on ? 1 : 0;
}

int eppk_quad_stats(eppk_ctx* c, uint64_t* launches, uint64_t* deferred) {
  if (!c) return EPPK_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->cfg.device));
  { const int rcs_ = device_sync(c); if (rcs_) return rcs_; }
  quad_consume_reports(c);                 // (every launch has finished: every report is there)
  if (launches) *launches = c->quad_launches;
  if (deferred) *deferred = c->quad_deferred_seen;
  return EPPK_OK;
}

int eppk_profile_enable(eppk_ctx* c, int on) {
  if (!c) return EPPK_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->cfg.device));
  { const int rcs_ = device_sync(c); if (rcs_) return rcs_; }
  HIPCHK(c, hipMemset(c->stats + 4, 0, 2 * (size_t)kStatSlots * kStatBanks * sizeof(unsigned long long)));
  c->prof = on != 0;
  c->prof_every = on > 1 ? (uint32_t)on : 1u;
  c->prof_tick = 0;
  c->ev_used = 0;
  c->fixed_bytes = 0;
  c->launches = 0;
  return EPPK_OK;
}

int eppk_profile_drain(eppk_ctx* c, float* ms, uint32_t cap, uint32_t* n_out) {
  if (!c || (!ms && cap) || !n_out) return EPPK_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->cfg.device));
  uint32_t n = 0;
  for (size_t i = 0; i + 1 < c->ev_used && n < cap; i += 2) {
    HIPCHK(c, hipEventSynchronize(c->ev[i + 1]));
    float t = 0.f;
    HIPCHK(c, hipEventElapsedTime(&t, c->ev[i], c->ev[i + 1]));
    ms[n++] = t;
  }
  *n_out = n;
  c->ev_used = 0;
  return EPPK_OK;
}

int eppk_profile_bytes(eppk_ctx* c, uint64_t* bytes, uint64_t* lookups, uint32_t* launches) {
  if (!c || !bytes) return EPPK_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->cfg.device));
  { const int rcs_ = device_sync(c); if (rcs_) return rcs_; }
  std::vector<unsigned long long> slots(2 * (size_t)kStatSlots * kStatBanks);
  HIPCHK(c, hipMemcpy(slots.data(), c->stats + 4, slots.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  unsigned long long st[2] = {0, 0};
  for (size_t i = 0; i < slots.size(); i += 2) { st[0] += slots[i]; st[1] += slots[i + 1]; }
  // a hit reads key + pod-set row, the terminating miss reads a key
  const uint64_t row = 8u + 64u * (uint64_t)c->lw_bytes;
  *bytes = c->fixed_bytes + (uint64_t)st[0] * row + (uint64_t)(st[1] - st[0]) * 8u;
  if (lookups) *lookups = st[1];
  if (launches) *launches = c->launches;
  return EPPK_OK;
}


// ================================================================================================
// Device groups: ONE picker over several GPUs, behind the C ABI (SURVEY.md §8(b) "device list", §8(e)).
//
// Snapshot and prefix index are replicated on every member device; a batch shards by request -- device g scores rows
// [g*per, (g+1)*per), per = ceil(n / devices used) -- with NO data-path collective: a pick depends only on its own request row
// and the replicated read-only state.  Every device returns its shard of picks to the host over its own PCIe link.  The one
// exchange, an all-gather of the int32 picks so that EVERY device holds all of them, happens only when something on the devices
// needs them: the post-route index update that each device applies to its own replica (EPPK_GROUP_LEARN), or a caller that asks
// for it (EPPK_GROUP_GATHER).  Three ways to do it (eppk_group_create `gather_mode`):
//   PEER  every device pushes its shard into each peer's pick array with hipMemcpyPeerAsync on its own stream (xGMI is
//         point-to-point and fully connected: G-1 one-hop copies of n/G * 4 bytes each, no ring), peers wait on its event;
//   RCCL  ncclAllGather, in place, one communicator per device in ONE process (ncclCommInitAll; librccl is dlopen'ed so that
//         libeppk does not link it): the collective north_star names;
//   HOST  the picks every device already returns to the host are uploaded back to all of them (no device-to-device traffic).
struct eppk_group {
  std::vector<eppk_ctx*> ctx;
  std::vector<int> dev;
  uint32_t mode = 0;
  uint32_t min_shard = 2048;
  uint32_t max_batch = 0;
  void* h_stage = nullptr; size_t h_stage_bytes = 0;   // ONE pinned, portable copy of a batch: every device DMAs from it
  std::vector<hipEvent_t> ev;                           // ev[g]: device g's shard of picks is in every peer's array (PEER)
  // ordered fallbacks over the group (eppk_group_pick_topk / _random_topk): pinned, portable mask and result staging
  uint64_t* h_tk_mask = nullptr; int32_t* h_tk_pick = nullptr; double* h_tk_score = nullptr;
  // The PIPELINED host path over the group (eppk_group_pick_stage_*): per set ONE pinned, portable buffer of rows / masks / results that
  // the caller fills and every member's DMA engine reads; the members work on their own staging sets' streams and device buffers
  // (eppk_ctx::stage[set]).  sev[set][i]: member i's shard of picks has landed in every peer's array (PEER gather of a LEARN batch).
  struct GStage {
    void* h_reqs = nullptr; uint64_t* h_mask = nullptr; int32_t* h_pick = nullptr; double* h_score = nullptr;
    uint32_t n = 0, used = 0, per = 0; bool busy = false, had_mask = false, learn = false, host_learn = false;
    std::vector<hipEvent_t> sev;
    std::vector<hipEvent_t> cev;   // cev[p]: member p's update has read its copy of this set's gathered picks: the peers may push the next batch's into it
  };
  GStage gstage[EPPK_STAGE_SETS];
  // RCCL (dlopen)
  void* rccl = nullptr;
  std::vector<void*> comms;
  int ranks_seen = 0;
  int (*nccl_comm_init_all)(void**, int, const int*) = nullptr;
  int (*nccl_comm_destroy)(void*) = nullptr;
  int (*nccl_comm_count)(void*, int*) = nullptr;
  int (*nccl_group_start)() = nullptr;
  int (*nccl_group_end)() = nullptr;
  int (*nccl_all_gather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  const char* (*nccl_err)(int) = nullptr;
  std::string err;
};

namespace {

std::string g_group_err;

int gfail(eppk_group* g, int code, const std::string& msg) {
  if (g) g->err = msg;
  else { std::lock_guard<std::mutex> l(g_err_mu); g_group_err = msg; }
  return code;
}

int group_load_rccl(eppk_group* g) {
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
  for (const char* n : names) { g->rccl = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (g->rccl) break; }
  if (!g->rccl) return gfail(g, EPPK_ERR_DEVICE, std::string("eppk_group_create: cannot load librccl: ") + dlerror());
  auto sym = [&](const char* n) { return dlsym(g->rccl, n); };
  g->nccl_comm_init_all = (int (*)(void**, int, const int*))sym("ncclCommInitAll");
  g->nccl_comm_destroy = (int (*)(void*))sym("ncclCommDestroy");
  g->nccl_comm_count = (int (*)(void*, int*))sym("ncclCommCount");
  g->nccl_group_start = (int (*)())sym("ncclGroupStart");
  g->nccl_group_end = (int (*)())sym("ncclGroupEnd");
  g->nccl_all_gather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))sym("ncclAllGather");
  g->nccl_err = (const char* (*)(int))sym("ncclGetErrorString");
  if (!g->nccl_comm_init_all || !g->nccl_comm_destroy || !g->nccl_group_start || !g->nccl_group_end || !g->nccl_all_gather)
    return gfail(g, EPPK_ERR_DEVICE, "eppk_group_create: librccl lacks an expected symbol");
  return EPPK_OK;
}

#define GFOR(g, i) for (uint32_t i = 0; i < (uint32_t)(g)->ctx.size(); ++i)
// run `expr` (an int-returning call on member i) on every member; first failure wins
#define GALL(g, expr)                                                                            \
  do {                                                                                           \
    GFOR(g, i) {                                                                                 \
      eppk_ctx* m = (g)->ctx[i]; (void)m;                                                        \
      const int rc_ = (expr);                                                                    \
      if (rc_ != EPPK_OK) return gfail((g), rc_, "device " + std::to_string((g)->dev[i]) + ": " + eppk_last_error(m)); \
    }                                                                                            \
  } while (0)

}  // namespace

const char* eppk_group_last_error(const eppk_group* g) {
  if (g) return g->err.c_str();
  std::lock_guard<std::mutex> l(g_err_mu);
  static thread_local std::string copy;
  copy = g_group_err;
  return copy.c_str();
}

int eppk_group_create(const eppk_cfg* cfg, const int32_t* devices, uint32_t n_devices, uint32_t gather_mode, eppk_group** out) {
  if (!cfg || !devices || !out || n_devices == 0) return gfail(nullptr, EPPK_ERR_ARG, "eppk_group_create: null argument / no devices");
  *out = nullptr;
  if (n_devices > EPPK_GROUP_MAX_DEVICES) return gfail(nullptr, EPPK_ERR_LIMIT, "eppk_group_create: more than EPPK_GROUP_MAX_DEVICES devices");
  if (gather_mode > EPPK_GATHER_HOST) return gfail(nullptr, EPPK_ERR_ARG, "eppk_group_create: unknown gather mode");
  eppk_group* g = new (std::nothrow) eppk_group();
  if (!g) return gfail(nullptr, EPPK_ERR_NOMEM, "eppk_group_create: out of memory");
  g->mode = gather_mode;
  g->max_batch = cfg->max_batch;
  auto bail = [&](int code, const std::string& m) { eppk_group_destroy(g); return gfail(nullptr, code, m); };
  for (uint32_t i = 0; i < n_devices; ++i) {
    eppk_cfg c = *cfg;
    c.device = devices[i];
    eppk_ctx* m = nullptr;
    const int rc = eppk_create(&c, &m);
    if (rc != EPPK_OK) return bail(rc, std::string("eppk_group_create: device ") + std::to_string(devices[i]) + ": " + eppk_last_error(nullptr));
    g->ctx.push_back(m);
    g->dev.push_back(devices[i]);
  }
  // peer access between distinct devices (hipMemcpyPeerAsync works without it, through host memory; with it the copy is one xGMI hop)
  for (uint32_t i = 0; i < n_devices; ++i)
    for (uint32_t j = 0; j < n_devices; ++j) {
      if (devices[i] == devices[j]) continue;
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, devices[i], devices[j]) == hipSuccess && can) {
        (void)hipSetDevice(devices[i]);
        const hipError_t e = hipDeviceEnablePeerAccess(devices[j], 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
      }
    }
  for (uint32_t i = 0; i < n_devices; ++i) {
    hipEvent_t e;
    (void)hipSetDevice(devices[i]);
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return bail(EPPK_ERR_DEVICE, "eppk_group_create: hipEventCreate failed");
    g->ev.push_back(e);
  }
  g->ranks_seen = (int)n_devices;
  if (gather_mode == EPPK_GATHER_RCCL) {
    for (uint32_t i = 0; i < n_devices; ++i)
      for (uint32_t j = i + 1; j < n_devices; ++j)
        if (devices[i] == devices[j]) return bail(EPPK_ERR_ARG, "eppk_group_create: EPPK_GATHER_RCCL needs distinct devices (one communicator rank per GPU)");
    const int rc = group_load_rccl(g);
    if (rc) { const std::string m = g->err; return bail(rc, m); }
    g->comms.assign(n_devices, nullptr);
    std::vector<int> devs(devices, devices + n_devices);
    const int nrc = g->nccl_comm_init_all(g->comms.data(), (int)n_devices, devs.data());
    if (nrc != 0) { g->comms.clear(); return bail(EPPK_ERR_DEVICE, std::string("ncclCommInitAll: ") + (g->nccl_err ? g->nccl_err(nrc) : "error")); }
    int cnt = 0;
    if (g->nccl_comm_count && g->nccl_comm_count(g->comms[0], &cnt) == 0) g->ranks_seen = cnt;
  }
  *out = g;
  return EPPK_OK;
}

void eppk_group_destroy(eppk_group* g) {
  if (!g) return;
  for (void* c : g->comms) if (c && g->nccl_comm_destroy) (void)g->nccl_comm_destroy(c);
  for (size_t i = 0; i < g->ev.size(); ++i) { (void)hipSetDevice(g->dev[i]); (void)hipEventDestroy(g->ev[i]); }
  for (eppk_group::GStage& gs : g->gstage) {
    for (size_t i = 0; i < gs.sev.size(); ++i) { (void)hipSetDevice(g->dev[i]); (void)hipEventDestroy(gs.sev[i]); }
    for (size_t i = 0; i < gs.cev.size(); ++i) { (void)hipSetDevice(g->dev[i]); (void)hipEventDestroy(gs.cev[i]); }
    if (gs.h_reqs) (void)hipHostFree(gs.h_reqs);
    if (gs.h_mask) (void)hipHostFree(gs.h_mask);
    if (gs.h_pick) (void)hipHostFree(gs.h_pick);
    if (gs.h_score) (void)hipHostFree(gs.h_score);
  }
  for (eppk_ctx* m : g->ctx) eppk_destroy(m);
  if (g->h_stage) (void)hipHostFree(g->h_stage);
  if (g->h_tk_mask) (void)hipHostFree(g->h_tk_mask);
  if (g->h_tk_pick) (void)hipHostFree(g->h_tk_pick);
  if (g->h_tk_score) (void)hipHostFree(g->h_tk_score);
  delete g;     // (librccl stays loaded: unloading a library that owns device state is not worth the risk)
}

uint32_t eppk_group_size(const eppk_group* g) { return g ? (uint32_t)g->ctx.size() : 0u; }
eppk_ctx* eppk_group_ctx(eppk_group* g, uint32_t i) { return (g && i < g->ctx.size()) ? g->ctx[i] : nullptr; }
int eppk_group_ranks_seen(const eppk_group* g) { return g ? g->ranks_seen : EPPK_ERR_ARG; }
int eppk_group_set_min_shard(eppk_group* g, uint32_t n) { if (!g || n == 0) return EPPK_ERR_ARG; g->min_shard = n; return EPPK_OK; }

// (EPPK_GATHER_HOST: a staged LEARN batch whose update is still owed is given it before anything else touches the replicas, the epoch
// counter included -- the update is stamped with the epoch of ITS begin's place in the call order)
namespace { void group_host_learn_flush(eppk_group* g, uint32_t set); }
#define GFLUSH(g) do { for (uint32_t set_ = 0; set_ < EPPK_STAGE_SETS; ++set_) group_host_learn_flush((g), set_); } while (0)

int eppk_group_snapshot_publish(eppk_group* g, const eppk_pod_row* rows, uint32_t n_pods, uint64_t epoch) {
  if (!g) return EPPK_ERR_ARG;
  GFLUSH(g);
  GALL(g, eppk_snapshot_publish(m, rows, n_pods, epoch));
  return EPPK_OK;
}
int eppk_group_index_clear(eppk_group* g) { if (!g) return EPPK_ERR_ARG; GFLUSH(g); GALL(g, eppk_index_clear(m)); return EPPK_OK; }
int eppk_group_index_insert(eppk_group* g, const uint64_t* hashes, const uint32_t* pods, uint32_t n) {
  if (!g) return EPPK_ERR_ARG;
  GFLUSH(g);
  GALL(g, eppk_index_insert(m, hashes, pods, n));
  return EPPK_OK;
}
int eppk_group_index_remove_pod(eppk_group* g, uint32_t pod) { if (!g) return EPPK_ERR_ARG; GFLUSH(g); GALL(g, eppk_index_remove_pod(m, pod)); return EPPK_OK; }
int eppk_group_index_advance_epoch(eppk_group* g, uint32_t* new_epoch) {
  if (!g) return EPPK_ERR_ARG;
  GFLUSH(g);
  GALL(g, eppk_index_advance_epoch(m, new_epoch));
  return EPPK_OK;
}
int eppk_group_index_evict_older(eppk_group* g, uint32_t min_epoch, uint32_t* n_evicted) {
  if (!g) return EPPK_ERR_ARG;
  GFLUSH(g);
  GALL(g, eppk_index_evict_older(m, min_epoch, n_evicted));     // (replicas are identical: every member evicts the same hashes)
  return EPPK_OK;
}

int eppk_group_pick_batch(eppk_group* g, const void* reqs, uint32_t n_reqs, const uint64_t* cand_mask, int32_t* out_pick,
                          double* out_score, uint32_t flags) {
  if (!g || ((!reqs || !out_pick) && n_reqs)) return gfail(g, EPPK_ERR_ARG, "eppk_group_pick_batch: null argument");
  if (n_reqs > g->max_batch) return gfail(g, EPPK_ERR_LIMIT, "eppk_group_pick_batch: n_reqs > max_batch");
  if (n_reqs == 0) return EPPK_OK;
  GFLUSH(g);
  const uint32_t G = (uint32_t)g->ctx.size();
  eppk_ctx* c0 = g->ctx[0];
  for (eppk_ctx* m : g->ctx) if (!m->have_snapshot) return gfail(g, EPPK_ERR_NO_SNAPSHOT, "eppk_group_pick_batch: no snapshot published");
  const bool learn = (flags & EPPK_GROUP_LEARN) != 0 && c0->slots != 0 && c0->cfg.max_blocks != 0;
  const bool gather = learn || (flags & EPPK_GROUP_GATHER) != 0;
  int rc = validate_rows(c0, "eppk_group_pick_batch", reqs, n_reqs);
  if (rc) return gfail(g, rc, eppk_last_error(c0));
  // devices used: a small batch stays on one GPU (north_star: shard only when requests x pods outgrow one device)
  uint32_t used = (n_reqs + g->min_shard - 1) / g->min_shard;
  used = used < 1 ? 1 : used > G ? G : used;
  const uint32_t per = (n_reqs + used - 1) / used;
  const size_t J = (c0->n_pods + 63u) / 64u;
  // ONE pinned + portable staging copy of the batch; every device's DMA engine reads it over its own PCIe link
  const size_t need = (size_t)g->max_batch * c0->stride;
  if (g->h_stage_bytes < need) {
    if (g->h_stage) (void)hipHostFree(g->h_stage);
    g->h_stage = nullptr; g->h_stage_bytes = 0;
    if (hipHostMalloc(&g->h_stage, need, hipHostMallocPortable) != hipSuccess) return gfail(g, EPPK_ERR_NOMEM, "eppk_group_pick_batch: pinned staging");
    g->h_stage_bytes = need;
  }
  std::memcpy(g->h_stage, reqs, (size_t)n_reqs * c0->stride);
  auto lo_of = [&](uint32_t i) { const uint64_t l = (uint64_t)i * per; return (uint32_t)(l < n_reqs ? l : n_reqs); };
  auto cnt_of = [&](uint32_t i) { if (i >= used) return 0u; const uint32_t l = lo_of(i), h = lo_of(i + 1); return h - l; };
  // A: every device: H2D (its shard, or the whole batch when the device-side picks will be gathered) -> kernel -> D2H of its picks
  GFOR(g, i) {
    const uint32_t lo = lo_of(i), cnt = cnt_of(i);
    if (cnt == 0 && !gather) continue;
    rc = pick_host_begin(g->ctx[i], (const uint8_t*)g->h_stage, true, n_reqs, lo, cnt, gather,
                         (cand_mask && cnt) ? cand_mask + (size_t)lo * J : nullptr);
    if (rc) return gfail(g, rc, "device " + std::to_string(g->dev[i]) + ": " + eppk_last_error(g->ctx[i]));
  }
  // B: all-gather of the picks on the devices
  if (gather && g->mode == EPPK_GATHER_PEER) {
    GFOR(g, i) {
      const uint32_t lo = lo_of(i), cnt = cnt_of(i);
      eppk_ctx* m = g->ctx[i];
      if (cnt) {
        (void)hipSetDevice(g->dev[i]);
        GFOR(g, p) {
          if (p == i) continue;
          const hipError_t e = g->dev[p] == g->dev[i]
              ? hipMemcpyAsync(g->ctx[p]->d_pick + lo, m->d_pick + lo, (size_t)cnt * 4u, hipMemcpyDeviceToDevice, m->stream)   // (two members on one GPU: tests)
              : hipMemcpyPeerAsync(g->ctx[p]->d_pick + lo, g->dev[p], m->d_pick + lo, g->dev[i], (size_t)cnt * 4u, m->stream);
          if (e != hipSuccess) return gfail(g, EPPK_ERR_DEVICE, std::string("eppk_group_pick_batch: peer copy failed: ") + hipGetErrorString(e));
        }
        if (hipEventRecord(g->ev[i], m->stream) != hipSuccess) return gfail(g, EPPK_ERR_DEVICE, "eppk_group_pick_batch: hipEventRecord failed");
      }
    }
    GFOR(g, p) {
      (void)hipSetDevice(g->dev[p]);
      GFOR(g, i) if (i != p && cnt_of(i)) (void)hipStreamWaitEvent(g->ctx[p]->stream, g->ev[i], 0);
    }
  } else if (gather && g->mode == EPPK_GATHER_RCCL) {
    // in place: rank i's send buffer is its own slice of the receive buffer (per entries per rank; d_pick has room for per * G).
    // With fewer devices used than members the unused ranks contribute padding behind the batch.
    const uint32_t perc = (n_reqs + G - 1) / G;
    if (used != G) return gfail(g, EPPK_ERR_ARG, "eppk_group_pick_batch: EPPK_GATHER_RCCL shards over all members (set min_shard <= n_reqs / devices)");
    (void)perc;
    int nrc = g->nccl_group_start();
    GFOR(g, i) {
      eppk_ctx* m = g->ctx[i];
      (void)hipSetDevice(g->dev[i]);
      if (nrc == 0) nrc = g->nccl_all_gather(m->d_pick + (size_t)i * per, m->d_pick, per, /*ncclInt32*/ 2, g->comms[i], m->stream);
    }
    const int erc = g->nccl_group_end();
    if (nrc != 0 || erc != 0) return gfail(g, EPPK_ERR_DEVICE, std::string("ncclAllGather: ") + (g->nccl_err ? g->nccl_err(nrc ? nrc : erc) : "error"));
  }
  // C (PEER / RCCL): every device applies the SAME post-route index update to its replica, from the gathered picks
  if (learn && g->mode != EPPK_GATHER_HOST) {
    GFOR(g, p) {
      eppk_ctx* m = g->ctx[p];
      rc = eppk_index_insert_picks_device(m, m->d_reqs, m->d_pick, n_reqs, m->stream);
      if (rc) return gfail(g, rc, "device " + std::to_string(g->dev[p]) + ": " + eppk_last_error(m));
    }
  }
  // D: wait, hand the shards to the caller
  GFOR(g, i) {
    const uint32_t lo = lo_of(i), cnt = cnt_of(i);
    if (cnt == 0 && !gather) continue;
    rc = pick_host_end(g->ctx[i], cnt, cand_mask != nullptr, out_pick + lo, out_score ? out_score + lo : nullptr);
    if (rc) return gfail(g, rc, "device " + std::to_string(g->dev[i]) + ": " + eppk_last_error(g->ctx[i]));
  }
  // HOST gather: the picks are on the host now; upload all of them to every device (and learn from them)
  if (gather && g->mode == EPPK_GATHER_HOST) {
    GFOR(g, p) {
      eppk_ctx* m = g->ctx[p];
      (void)hipSetDevice(g->dev[p]);
      if (hipMemcpyAsync(m->d_pick, out_pick, (size_t)n_reqs * 4u, hipMemcpyHostToDevice, m->stream) != hipSuccess)
        return gfail(g, EPPK_ERR_DEVICE, "eppk_group_pick_batch: upload of the gathered picks failed");
      if (learn) {
        rc = eppk_index_insert_picks_device(m, m->d_reqs, m->d_pick, n_reqs, m->stream);
        if (rc) return gfail(g, rc, "device " + std::to_string(g->dev[p]) + ": " + eppk_last_error(m));
      }
    }
    GFOR(g, p) { (void)hipSetDevice(g->dev[p]); if (hipStreamSynchronize(g->ctx[p]->stream) != hipSuccess) return gfail(g, EPPK_ERR_DEVICE, "eppk_group_pick_batch: sync failed"); }
  }
  return EPPK_OK;
}

// Device-resident picks over a group (include/eppk.h): every member scores the rows that already sit in ITS memory, on its own
// stream; nothing crosses the host.  With EPPK_GROUP_GATHER every member then receives every other member's picks (member-major).
int eppk_group_pick_device(eppk_group* g, const void* const* d_reqs, const uint32_t* n_rows, int32_t* const* d_out_pick, double* const* d_out_score,
                           int32_t* const* d_gathered, uint32_t flags) {
  if (!g || !d_reqs || !n_rows || !d_out_pick) return gfail(g, EPPK_ERR_ARG, "eppk_group_pick_device: null argument");
  const bool gather = (flags & EPPK_GROUP_GATHER) != 0;
  if (flags & EPPK_GROUP_LEARN) return gfail(g, EPPK_ERR_ARG, "eppk_group_pick_device: EPPK_GROUP_LEARN needs the whole batch on every member: eppk_group_pick_batch");
  if (gather && !d_gathered) return gfail(g, EPPK_ERR_ARG, "eppk_group_pick_device: EPPK_GROUP_GATHER without d_gathered");
  if (gather && g->mode == EPPK_GATHER_HOST) return gfail(g, EPPK_ERR_ARG, "eppk_group_pick_device: a device-resident gather needs EPPK_GATHER_PEER or EPPK_GATHER_RCCL");
  for (eppk_ctx* m : g->ctx) if (!m->have_snapshot) return gfail(g, EPPK_ERR_NO_SNAPSHOT, "eppk_group_pick_device: no snapshot published");
  GFLUSH(g);
  const uint32_t G = (uint32_t)g->ctx.size();
  std::vector<size_t> off(G + 1, 0);
  GFOR(g, i) off[i + 1] = off[i] + n_rows[i];
  GFOR(g, i) {
    if (n_rows[i] == 0) continue;
    if (!d_reqs[i] || !d_out_pick[i]) return gfail(g, EPPK_ERR_ARG, "eppk_group_pick_device: null shard pointer");
    int rc = eppk_pick_batch_device(g->ctx[i], d_reqs[i], n_rows[i], nullptr, d_out_pick[i], d_out_score ? d_out_score[i] : nullptr, (void*)g->ctx[i]->stream);
    if (rc) return gfail(g, rc, "device " + std::to_string(g->dev[i]) + ": " + eppk_last_error(g->ctx[i]));
  }
  if (!gather) return EPPK_OK;
  if (g->mode == EPPK_GATHER_PEER) {
    GFOR(g, i) {
      if (n_rows[i] == 0) continue;
      eppk_ctx* m = g->ctx[i];
      (void)hipSetDevice(g->dev[i]);
      GFOR(g, p) {
        if (!d_gathered[p]) return gfail(g, EPPK_ERR_ARG, "eppk_group_pick_device: null d_gathered entry");
        const hipError_t e = g->dev[p] == g->dev[i]
            ? hipMemcpyAsync(d_gathered[p] + off[i], d_out_pick[i], (size_t)n_rows[i] * 4u, hipMemcpyDeviceToDevice, m->stream)
            : hipMemcpyPeerAsync(d_gathered[p] + off[i], g->dev[p], d_out_pick[i], g->dev[i], (size_t)n_rows[i] * 4u, m->stream);
        if (e != hipSuccess) return gfail(g, EPPK_ERR_DEVICE, std::string("eppk_group_pick_device: peer copy failed: ") + hipGetErrorString(e));
      }
      if (hipEventRecord(g->ev[i], m->stream) != hipSuccess) return gfail(g, EPPK_ERR_DEVICE, "eppk_group_pick_device: hipEventRecord failed");
    }
    GFOR(g, p) {                                     // member p's stream continues once every shard has landed in ITS array
      (void)hipSetDevice(g->dev[p]);
      GFOR(g, i) if (i != p && n_rows[i]) (void)hipStreamWaitEvent(g->ctx[p]->stream, g->ev[i], 0);
    }
  } else {                                           // RCCL: equal shards, one all-gather
    GFOR(g, i) if (n_rows[i] != n_rows[0] || !d_gathered[i]) return gfail(g, EPPK_ERR_ARG, "eppk_group_pick_device: EPPK_GATHER_RCCL needs equal shards on every member");
    int nrc = g->nccl_group_start();
    GFOR(g, i) {
      (void)hipSetDevice(g->dev[i]);
      if (nrc == 0) nrc = g->nccl_all_gather(d_out_pick[i], d_gathered[i], n_rows[0], /*ncclInt32*/ 2, g->comms[i], g->ctx[i]->stream);
    }
    const int erc = g->nccl_group_end();
    if (nrc != 0 || erc != 0) return gfail(g, EPPK_ERR_DEVICE, std::string("ncclAllGather: ") + (g->nccl_err ? g->nccl_err(nrc ? nrc : erc) : "error"));
  }
  return EPPK_OK;
}

// ---- what a shim calls beside eppk_group_pick_batch: ageing, per-pod capacity, ordered fallbacks, the pipelined host path ------------------

int eppk_group_index_evict_older_device(eppk_group* g, uint32_t min_epoch) {
  if (!g) return EPPK_ERR_ARG;
  // every member: on its own stream, behind the picks (and LEARN updates) of the staging sets begun before, ahead of those begun after
  GFLUSH(g);
  GALL(g, eppk_index_evict_older_device(m, min_epoch, nullptr));
  return EPPK_OK;
}

int eppk_group_index_trim_pods(eppk_group* g, uint32_t cap_per_pod, uint64_t* n_removed) {
  if (!g) return EPPK_ERR_ARG;
  GFLUSH(g);
  GALL(g, eppk_index_trim_pods(m, cap_per_pod, n_removed));     // (replicas are identical: every member removes the same pairs)
  return EPPK_OK;
}

namespace {

int group_pinned(eppk_group* g, void** p, size_t bytes, const char* what) {
  if (*p) return EPPK_OK;
  if (hipHostMalloc(p, bytes ? bytes : 8u, hipHostMallocPortable) != hipSuccess) { *p = nullptr; (void)hipGetLastError(); return gfail(g, EPPK_ERR_NOMEM, std::string(what) + ": pinned staging"); }
  return EPPK_OK;
}

// how a batch of n requests is spread over the members: `used` of them, `per` rows each (the last one what is left)
struct Shards {
  uint32_t n, used, per;
  Shards(const eppk_group* g, uint32_t n_) : n(n_) {
    const uint32_t G = (uint32_t)g->ctx.size();
    used = (n + g->min_shard - 1) / g->min_shard;
    used = used < 1 ? 1 : used > G ? G : used;
    per = (n + used - 1) / used;
  }
  uint32_t lo(uint32_t i) const { const uint64_t l = (uint64_t)i * per; return (uint32_t)(l < n ? l : n); }
  uint32_t cnt(uint32_t i) const { return i >= used ? 0u : lo(i + 1) - lo(i); }
};

// ordered fallbacks / picker "random-top-k" over the group: sharded by request like eppk_group_pick_batch, every member busy at once
int group_topk(eppk_group* g, const char* who, const void* reqs, uint32_t n_reqs, const uint64_t* cand_mask, uint32_t k, bool random, uint64_t seed,
               int32_t* out_pick, double* out_score) {
  if (!g || ((!reqs || !out_pick) && n_reqs)) return gfail(g, EPPK_ERR_ARG, std::string(who) + ": null argument");
  if (k < 1 || k > EPPK_MAX_TOPK) return gfail(g, EPPK_ERR_ARG, std::string(who) + ": k out of range (1..8)");
  if (n_reqs > g->max_batch) return gfail(g, EPPK_ERR_LIMIT, std::string(who) + ": n_reqs > max_batch");
  if (n_reqs == 0) return EPPK_OK;
  GFLUSH(g);
  eppk_ctx* c0 = g->ctx[0];
  for (eppk_ctx* m : g->ctx) if (!m->have_snapshot) return gfail(g, EPPK_ERR_NO_SNAPSHOT, std::string(who) + ": no snapshot published");
  for (eppk_ctx* m : g->ctx) if (m->assumed_epochs) return gfail(g, EPPK_ERR_ARG, std::string(who) + ": device groups do not support assumed load");
  int rc = validate_rows(c0, who, reqs, n_reqs);
  if (rc) return gfail(g, rc, eppk_last_error(c0));
  const size_t J = (c0->n_pods + 63u) / 64u;
  const uint32_t ok = random ? 1u : k;                         // entries per request in the caller's arrays
  if (cand_mask && !J) {
    for (size_t i = 0; i < (size_t)n_reqs * ok; ++i) { out_pick[i] = EPPK_NO_PICK; if (out_score) out_score[i] = 0.0; }
    return EPPK_OK;
  }
  const size_t mb = g->max_batch;
  if ((rc = group_pinned(g, &g->h_stage, mb * c0->stride, who))) return rc;
  g->h_stage_bytes = g->h_stage_bytes < mb * c0->stride ? mb * c0->stride : g->h_stage_bytes;
  if ((rc = group_pinned(g, (void**)&g->h_tk_pick, mb * EPPK_MAX_TOPK * 4u, who))) return rc;
  if ((rc = group_pinned(g, (void**)&g->h_tk_score, mb * EPPK_MAX_TOPK * 8u, who))) return rc;
  if (cand_mask && (rc = group_pinned(g, (void**)&g->h_tk_mask, mb * c0->jmax * 8u, who))) return rc;
  std::memcpy(g->h_stage, reqs, (size_t)n_reqs * c0->stride);
  if (cand_mask) std::memcpy(g->h_tk_mask, cand_mask, (size_t)n_reqs * J * 8u);
  const Shards sh(g, n_reqs);
  GFOR(g, i) {
    const uint32_t lo = sh.lo(i), cnt = sh.cnt(i);
    if (!cnt) continue;
    eppk_ctx* m = g->ctx[i];
    auto mfail = [&](int code) { return gfail(g, code, "device " + std::to_string(g->dev[i]) + ": " + eppk_last_error(m)); };
    if (hipSetDevice(g->dev[i]) != hipSuccess) return gfail(g, EPPK_ERR_DEVICE, std::string(who) + ": hipSetDevice failed");
    if ((rc = topk_ensure(m, cand_mask != nullptr))) return mfail(rc);
    if (hipMemcpyAsync(m->d_tk_reqs, (const uint8_t*)g->h_stage + (size_t)lo * m->stride, (size_t)cnt * m->stride, hipMemcpyHostToDevice, m->stream) != hipSuccess ||
        (cand_mask && hipMemcpyAsync(m->d_tk_mask, g->h_tk_mask + (size_t)lo * J, (size_t)cnt * J * 8u, hipMemcpyHostToDevice, m->stream) != hipSuccess))
      return gfail(g, EPPK_ERR_DEVICE, std::string(who) + ": upload failed");
    // (random-top-k hashes the request's index in the BATCH: r0 = the shard's first row, so that the split does not show)
    rc = run_pick(m, (const uint8_t*)m->d_tk_reqs, cnt, cand_mask ? m->d_tk_mask : nullptr, m->d_tk_pick, m->d_tk_score, m->stream, k, random, seed, lo);
    if (rc) return mfail(rc);
    if (hipMemcpyAsync(g->h_tk_pick + (size_t)lo * ok, m->d_tk_pick, (size_t)cnt * ok * 4u, hipMemcpyDeviceToHost, m->stream) != hipSuccess ||
        hipMemcpyAsync(g->h_tk_score + (size_t)lo * ok, m->d_tk_score, (size_t)cnt * ok * 8u, hipMemcpyDeviceToHost, m->stream) != hipSuccess)
      return gfail(g, EPPK_ERR_DEVICE, std::string(who) + ": download failed");
  }
  GFOR(g, i) {
    if (!sh.cnt(i)) continue;
    (void)hipSetDevice(g->dev[i]);
    if (hipStreamSynchronize(g->ctx[i]->stream) != hipSuccess) return gfail(g, EPPK_ERR_DEVICE, std::string(who) + ": device " + std::to_string(g->dev[i]) + " failed");
  }
  std::memcpy(out_pick, g->h_tk_pick, (size_t)n_reqs * ok * 4u);
  if (out_score) std::memcpy(out_score, g->h_tk_score, (size_t)n_reqs * ok * 8u);
  return EPPK_OK;
}

// EPPK_GATHER_HOST + EPPK_PICK_LEARN: the update of a staged batch cannot be chained on the devices before its picks have reached the
// host.  It is issued as soon as something needs the index it leaves behind: the set's own end, a begin of the other set (its pick
// must see it: this call then waits for the in-flight set's picks first -- the price of the mode), or the ageing step.
void group_host_learn_flush(eppk_group* g, uint32_t set) {
  eppk_group::GStage& gs = g->gstage[set];
  if (!gs.host_learn) return;
  gs.host_learn = false;
  GFOR(g, i) {                               // the picks of every shard must be in the pinned buffer
    eppk_ctx::StageSet& s = g->ctx[i]->stage[set];
    if (!s.busy) continue;
    (void)hipSetDevice(g->dev[i]);
    (void)hipEventSynchronize(s.picked);
  }
  GFOR(g, p) {
    eppk_ctx* m = g->ctx[p];
    eppk_ctx::StageSet& s = m->stage[set];
    (void)hipSetDevice(g->dev[p]);
    QuietRows quiet(m);
    int rc = hipMemcpyAsync(s.d_pick, gs.h_pick, (size_t)gs.n * 4u, hipMemcpyHostToDevice, s.st) == hipSuccess ? EPPK_OK : EPPK_ERR_DEVICE;
    const bool up_ev = rc == EPPK_OK && hipEventRecord(gs.sev[p], s.st) == hipSuccess;
    if (rc == EPPK_OK) rc = learn_picks(m, s.d_reqs, s.d_pick, gs.n, s.st);
    if (rc == EPPK_OK && hipEventRecord(m->learned, s.st) == hipSuccess) m->learn_pending = true;
    else m->host_flags |= EPPK_LAUNCH_LEARN_FAILED;
    // the pinned pick buffer is the caller's again when the set's end returns (the next begin's members write their shards into it):
    // the upload must have read it by then -- a few microseconds; the update itself runs on
    if (up_ev) (void)hipEventSynchronize(gs.sev[p]); else (void)hipStreamSynchronize(s.st);
  }
}

}  // namespace

int eppk_group_pick_topk(eppk_group* g, const void* reqs, uint32_t n_reqs, const uint64_t* cand_mask, uint32_t k, int32_t* out_pick, double* out_score) {
  return group_topk(g, "eppk_group_pick_topk", reqs, n_reqs, cand_mask, k, false, 0ull, out_pick, out_score);
}

int eppk_group_pick_random_topk(eppk_group* g, const void* reqs, uint32_t n_reqs, const uint64_t* cand_mask, uint32_t k, uint64_t seed, int32_t* out_pick,
                                double* out_score) {
  return group_topk(g, "eppk_group_pick_random_topk", reqs, n_reqs, cand_mask, k, true, seed, out_pick, out_score);
}

int eppk_group_pick_stage_buffers(eppk_group* g, uint32_t set, void** reqs, uint64_t** cand_mask) {
  if (!g || set >= EPPK_STAGE_SETS) return gfail(g, EPPK_ERR_ARG, "eppk_group_pick_stage_buffers: no such set");
  eppk_group::GStage& gs = g->gstage[set];
  eppk_ctx* c0 = g->ctx[0];
  const size_t mb = g->max_batch;
  int rc;
  if ((rc = group_pinned(g, &gs.h_reqs, mb * c0->stride, "eppk_group_pick_stage_buffers"))) return rc;
  if ((rc = group_pinned(g, (void**)&gs.h_pick, mb * 4u, "eppk_group_pick_stage_buffers"))) return rc;
  if ((rc = group_pinned(g, (void**)&gs.h_score, mb * 8u, "eppk_group_pick_stage_buffers"))) return rc;
  if (cand_mask && (rc = group_pinned(g, (void**)&gs.h_mask, mb * c0->jmax * 8u, "eppk_group_pick_stage_buffers"))) return rc;
  if (gs.sev.empty()) {
    GFOR(g, i) {
      hipEvent_t e;
      (void)hipSetDevice(g->dev[i]);
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return gfail(g, EPPK_ERR_DEVICE, "eppk_group_pick_stage_buffers: hipEventCreate failed");
      gs.sev.push_back(e);
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return gfail(g, EPPK_ERR_DEVICE, "eppk_group_pick_stage_buffers: hipEventCreate failed");
      gs.cev.push_back(e);
    }
  }
  GFOR(g, i) {       // every member's own set: stream, events, device buffers
    eppk_ctx* m = g->ctx[i];
    (void)hipSetDevice(g->dev[i]);
    if ((rc = stage_ensure(m, set, cand_mask != nullptr))) return gfail(g, rc, "device " + std::to_string(g->dev[i]) + ": " + eppk_last_error(m));
  }
  if (reqs) *reqs = gs.h_reqs;
  if (cand_mask) *cand_mask = gs.h_mask;
  return EPPK_OK;
}

int eppk_group_pick_stage_begin(eppk_group* g, uint32_t set, uint32_t n_reqs, int use_mask, uint32_t flags) {
  if (!g || set >= EPPK_STAGE_SETS) return gfail(g, EPPK_ERR_ARG, "eppk_group_pick_stage_begin: no such set");
  eppk_group::GStage& gs = g->gstage[set];
  if (!gs.h_reqs || gs.sev.empty() || (use_mask && !gs.h_mask)) return gfail(g, EPPK_ERR_ARG, "eppk_group_pick_stage_begin: eppk_group_pick_stage_buffers was not called for these buffers");
  if (gs.busy) return gfail(g, EPPK_ERR_ARG, "eppk_group_pick_stage_begin: the set is in flight (end it first)");
  if (n_reqs > g->max_batch) return gfail(g, EPPK_ERR_LIMIT, "eppk_group_pick_stage_begin: n_reqs > max_batch");
  eppk_ctx* c0 = g->ctx[0];
  for (eppk_ctx* m : g->ctx) if (!m->have_snapshot) return gfail(g, EPPK_ERR_NO_SNAPSHOT, "eppk_group_pick_stage_begin: no snapshot published");
  for (eppk_ctx* m : g->ctx) if (m->assumed_epochs) return gfail(g, EPPK_ERR_ARG, "eppk_group_pick_stage_begin: device groups do not support assumed load");
  if ((flags & EPPK_PICK_LEARN) && !c0->slots) return gfail(g, EPPK_ERR_ARG, "eppk_group_pick_stage_begin: EPPK_PICK_LEARN without a prefix index");
  const bool learn = (flags & EPPK_PICK_LEARN) != 0u && c0->cfg.max_blocks != 0u;
  const uint32_t G = (uint32_t)g->ctx.size();
  const Shards sh(g, n_reqs);
  if (learn && g->mode == EPPK_GATHER_RCCL && sh.used != G && n_reqs)
    return gfail(g, EPPK_ERR_ARG, "eppk_group_pick_stage_begin: EPPK_GATHER_RCCL shards over all members (set min_shard <= n_reqs / devices)");
  group_host_learn_flush(g, set ^ 1u);        // (EPPK_GATHER_HOST: the other set's update, which this set's picks must see)
  gs.n = n_reqs; gs.used = sh.used; gs.per = sh.per; gs.had_mask = use_mask != 0; gs.learn = learn; gs.host_learn = false; gs.busy = true;
  if (n_reqs == 0) return EPPK_OK;
  const size_t J = (c0->n_pods + 63u) / 64u;
  // A begin that fails has not begun (as eppk_pick_stage_begin): whatever the members had enqueued is waited for and every set is idle again.
  struct Abort {
    eppk_group* g; uint32_t set; eppk_group::GStage& gs; bool armed = true;
    ~Abort() {
      if (!armed) return;
      for (uint32_t i = 0; i < (uint32_t)g->ctx.size(); ++i) {
        eppk_ctx::StageSet& s = g->ctx[i]->stage[set];
        if (s.st) { (void)hipSetDevice(g->dev[i]); (void)hipStreamSynchronize(s.st); }
        s.busy = false; s.check_pending = false; s.copy_pending = false;
      }
      gs.busy = false;
    }
  } abort_guard{g, set, gs};
  // A: every member: upload (its shard; the whole batch when the index learns from it), row check, pick of its shard, picks and scores
  //    into the group's pinned result buffers
  GFOR(g, i) {
    const uint32_t lo = sh.lo(i), cnt = sh.cnt(i);
    eppk_ctx* m = g->ctx[i];
    eppk_ctx::StageSet& s = m->stage[set];
    s.busy = false; s.n = 0; s.check_pending = false; s.copy_pending = false;
    if (cnt == 0 && !learn) continue;
    auto mfail = [&](int code) { return gfail(g, code, "device " + std::to_string(g->dev[i]) + ": " + eppk_last_error(m)); };
    if (hipSetDevice(g->dev[i]) != hipSuccess) return gfail(g, EPPK_ERR_DEVICE, "eppk_group_pick_stage_begin: hipSetDevice failed");
    s.busy = true; s.n = learn ? n_reqs : cnt; s.had_mask = use_mask != 0;
    QuietRows quiet(m);                      // rows out of range are reported by end(), naming the row
    const uint32_t up_lo = learn ? 0u : lo, up_n = learn ? n_reqs : cnt;
    if (hipMemcpyAsync(s.d_reqs, (const uint8_t*)gs.h_reqs + (size_t)up_lo * m->stride, (size_t)up_n * m->stride, hipMemcpyHostToDevice, s.st) != hipSuccess)
      return gfail(g, EPPK_ERR_DEVICE, "eppk_group_pick_stage_begin: upload failed");
    int rc = row_check_ensure(m, &s.h_bad, &s.h_bad_dev, &s.st_check, &s.checked);
    if (!rc) rc = rows_check_launch(m, s.d_reqs, up_n, s.h_bad, s.h_bad_dev, s.st);
    if (rc) return mfail(rc);
    s.check_pending = true; s.check_side = false; s.row_base = up_lo;
    if (cnt) {
      if (use_mask && J && hipMemcpyAsync(s.d_mask, gs.h_mask + (size_t)lo * J, (size_t)cnt * J * 8u, hipMemcpyHostToDevice, s.st) != hipSuccess)
        return gfail(g, EPPK_ERR_DEVICE, "eppk_group_pick_stage_begin: mask upload failed");
      const uint8_t* d_shard = (const uint8_t*)s.d_reqs + (learn ? (size_t)lo * m->stride : 0u);
      // picks and scores at their BATCH positions of the member's arrays (a gather fills in the other members' shards around them)
      rc = run_pick(m, d_shard, cnt, (use_mask && J) ? s.d_mask : nullptr, s.d_pick + lo, s.d_score + lo, s.st, 1u, false, 0ull, 0u);
      if (rc) return mfail(rc);
      if (hipMemcpyAsync(gs.h_pick + lo, s.d_pick + lo, (size_t)cnt * 4u, hipMemcpyDeviceToHost, s.st) != hipSuccess ||
          hipMemcpyAsync(gs.h_score + lo, s.d_score + lo, (size_t)cnt * 8u, hipMemcpyDeviceToHost, s.st) != hipSuccess)
        return gfail(g, EPPK_ERR_DEVICE, "eppk_group_pick_stage_begin: download failed");
    }
    if (hipEventRecord(s.picked, s.st) != hipSuccess) return gfail(g, EPPK_ERR_DEVICE, "eppk_group_pick_stage_begin: hipEventRecord failed");
  }
  abort_guard.armed = false;                 // the picks are on their way: from here on the call succeeds and end() delivers them
  if (!learn) return EPPK_OK;
  // B: the picks of every shard onto every member, then C: each member applies the SAME post-route update to its replica.  A failure
  // here leaves the picks standing (EPPK_LAUNCH_LEARN_FAILED on the members, as eppk_pick_stage_begin does).
  auto learn_failed = [&](const std::string& why) {
    for (eppk_ctx* m : g->ctx) m->host_flags |= EPPK_LAUNCH_LEARN_FAILED;
    g->err = why;
    return EPPK_OK;
  };
  if (g->mode == EPPK_GATHER_HOST) { gs.host_learn = true; return EPPK_OK; }      // (the picks reach the host first: the update is chained in end())
  if (g->mode == EPPK_GATHER_PEER) {
    GFOR(g, i) {
      const uint32_t lo = sh.lo(i), cnt = sh.cnt(i);
      if (!cnt) continue;
      eppk_ctx::StageSet& s = g->ctx[i]->stage[set];
      (void)hipSetDevice(g->dev[i]);
      GFOR(g, p) {
        if (p == i) continue;
        // (member p's update of this set's PREVIOUS batch may still be reading the array the push lands in: it runs on p's stream, the
        // push on this member's -- an event that was never recorded does not hold anything up)
        (void)hipStreamWaitEvent(s.st, gs.cev[p], 0);
        int32_t* dst = g->ctx[p]->stage[set].d_pick + lo;
        const hipError_t e = g->dev[p] == g->dev[i] ? hipMemcpyAsync(dst, s.d_pick + lo, (size_t)cnt * 4u, hipMemcpyDeviceToDevice, s.st)
                                                    : hipMemcpyPeerAsync(dst, g->dev[p], s.d_pick + lo, g->dev[i], (size_t)cnt * 4u, s.st);
        if (e != hipSuccess) return learn_failed(std::string("eppk_group_pick_stage_begin: peer copy failed: ") + hipGetErrorString(e));
      }
      if (hipEventRecord(gs.sev[i], s.st) != hipSuccess) return learn_failed("eppk_group_pick_stage_begin: hipEventRecord failed");
    }
    GFOR(g, p) {
      (void)hipSetDevice(g->dev[p]);
      GFOR(g, i) if (i != p && sh.cnt(i)) (void)hipStreamWaitEvent(g->ctx[p]->stage[set].st, gs.sev[i], 0);
    }
  } else {                                   // RCCL: in place, per entries per rank (the arrays have room for per * G)
    int nrc = g->nccl_group_start();
    GFOR(g, i) {
      eppk_ctx::StageSet& s = g->ctx[i]->stage[set];
      (void)hipSetDevice(g->dev[i]);
      if (nrc == 0) nrc = g->nccl_all_gather(s.d_pick + (size_t)i * sh.per, s.d_pick, sh.per, /*ncclInt32*/ 2, g->comms[i], s.st);
    }
    const int erc = g->nccl_group_end();
    if (nrc != 0 || erc != 0) return learn_failed(std::string("ncclAllGather: ") + (g->nccl_err ? g->nccl_err(nrc ? nrc : erc) : "error"));
  }
  GFOR(g, p) {
    eppk_ctx* m = g->ctx[p];
    eppk_ctx::StageSet& s = m->stage[set];
    (void)hipSetDevice(g->dev[p]);
    QuietRows quiet(m);
    const int rc = learn_picks(m, s.d_reqs, s.d_pick, n_reqs, s.st);
    if (rc == EPPK_OK && hipEventRecord(m->learned, s.st) == hipSuccess) m->learn_pending = true;
    else m->host_flags |= EPPK_LAUNCH_LEARN_FAILED;
    (void)hipEventRecord(gs.cev[p], s.st);
  }
  return EPPK_OK;
}

int eppk_group_pick_stage_end(eppk_group* g, uint32_t set, int32_t* out_pick, double* out_score) {
  if (!g || set >= EPPK_STAGE_SETS) return gfail(g, EPPK_ERR_ARG, "eppk_group_pick_stage_end: no such set");
  eppk_group::GStage& gs = g->gstage[set];
  if (!gs.busy) return gfail(g, EPPK_ERR_ARG, "eppk_group_pick_stage_end: the set is not in flight");
  if (!out_pick && gs.n) return gfail(g, EPPK_ERR_ARG, "eppk_group_pick_stage_end: null argument");
  gs.busy = false;
  if (gs.n == 0) return EPPK_OK;
  uint32_t bad = 0xFFFFFFFFu;
  bool dev_failed = false;
  GFOR(g, i) {
    eppk_ctx::StageSet& s = g->ctx[i]->stage[set];
    if (!s.busy) continue;
    (void)hipSetDevice(g->dev[i]);
    if (hipEventSynchronize(s.picked) != hipSuccess) dev_failed = true;
    s.busy = false;
    if (s.check_pending) {
      s.check_pending = false;
      if (*s.h_bad != 0xFFFFFFFFu && s.row_base + *s.h_bad < bad) bad = s.row_base + *s.h_bad;
    }
  }
  // (EPPK_GATHER_HOST + LEARN: the owed update of a batch whose end fails.  A bad row: applied right here, as PEER / RCCL applied it in the
  //  begin and the single-context path behind the pick -- out-of-range rows are skipped by the update kernel itself.  A failed stream:
  //  there are no picks to learn from; the batch is not learned, whatever is called next.)
  if (dev_failed) { gs.host_learn = false; return gfail(g, EPPK_ERR_DEVICE, "eppk_group_pick_stage_end: a member's stream failed"); }
  if (bad != 0xFFFFFFFFu) {
    group_host_learn_flush(g, set);
    return gfail(g, EPPK_ERR_ARG, "eppk_group_pick_stage_end: request row " + std::to_string(bad) + " out of range");
  }
  eppk_ctx* c0 = g->ctx[0];
  const size_t J = (c0->n_pods + 63u) / 64u;
  std::memcpy(out_pick, gs.h_pick, (size_t)gs.n * 4u);
  if (out_score) std::memcpy(out_score, gs.h_score, (size_t)gs.n * 8u);
  if (gs.had_mask && !J) for (uint32_t r = 0; r < gs.n; ++r) { out_pick[r] = EPPK_NO_PICK; if (out_score) out_score[r] = 0.0; }
  group_host_learn_flush(g, set);             // HOST gather: the picks are here now: back onto every member, then its update
  return EPPK_OK;
}

int eppk_group_sync(eppk_group* g) {
  if (!g) return EPPK_ERR_ARG;
  GFOR(g, i) {
    (void)hipSetDevice(g->dev[i]);
    if (hipStreamSynchronize(g->ctx[i]->stream) != hipSuccess) return gfail(g, EPPK_ERR_DEVICE, "eppk_group_sync: device " + std::to_string(g->dev[i]));
  }
  return EPPK_OK;
}

void* eppk_group_stream(eppk_group* g, uint32_t i) { return (g && i < g->ctx.size()) ? (void*)g->ctx[i]->stream : nullptr; }

const int32_t* eppk_group_device_picks(eppk_group* g, uint32_t i) {
  return (g && i < g->ctx.size()) ? g->ctx[i]->d_pick : nullptr;
}

}  // extern "C"
