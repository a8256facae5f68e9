/*
 * eppk.h — C ABI of libeppk, the MI355X-native batched endpoint picker.
 *
 * This is the drop-in boundary for ONE hot path of the reference gateway: the per-request
 * endpoint pick.  Every entry point cites the reference interface it replaces
 * (paths relative to the reference tree):
 *
 *   - the picker seam      pkg/lwepp/handlers/server.go:79-82   (EndpointPicker.Pick)
 *   - its only call site   pkg/lwepp/handlers/request.go:141-163 (pickEndpoint)
 *   - the scheduler shape  docs/proposals/0845-scheduler-architecture-proposal/interfaces/interface.go:55-142
 *                          (Scheduler / SchedulerProfile / Filter / Scorer / WeightedScorer / Picker)
 *   - scorer inputs        docs/proposals/003-model-server-protocol/README.md:28-57
 *   - prefix chain/index   docs/proposals/0602-prefix-cache-aware-routing-proposal/README.md:99-122
 *
 * The reference has no FFI today (pure Go, CGO_ENABLED=0: lwepp.Dockerfile:8); INTEGRATION.md shows
 * the cgo binding a maintainer would add on top of this header.
 *
 * Conventions
 *   - plain C, plain pointers and sizes; no torch / HIP types in signatures (streams travel as void*).
 *   - every function returns EPPK_OK (0) or a negative eppk_status; nothing aborts, nothing calls back.
 *   - the caller owns all host buffers for the duration of the call only; the library owns device memory.
 *   - a context is bound to one GPU; calls on one context must be serialised by the caller
 *     (the Go shim's dispatcher goroutine does that: INTEGRATION.md).  Different contexts are independent.
 *   - arithmetic contract: SEMANTICS.md.  Picks are bit-exact against oracle/ on the same inputs.
 */
#ifndef EPPK_H
#define EPPK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EPPK_ABI_VERSION 4u   /* 2: device groups, launch status, random-top-k, assumed load, holes, per-pod capacity, async eviction; 3: eppk_pick_learn_device,
                               * EPPK_LAUNCH_LEARN_FAILED, the 254-epoch stamp window; 4: groups get fallbacks, the pipelined host path, async ageing and
                               * per-pod capacity (additions only) */

/* Limits of this build (SEMANTICS.md §limits). */
#define EPPK_MAX_PODS      4096u /* candidate endpoints per snapshot                         */
#define EPPK_MAX_ADAPTERS  128u  /* LoRA adapter ids 0..127 (two u64 words per pod row)      */
#define EPPK_MAX_BLOCKS    256u  /* prefix blocks per request (upstream maxPrefixBlocksToMatch) */
#define EPPK_MAX_SCORERS   8u    /* entries in a profile's weighted scorer chain             */

#define EPPK_MAX_TOPK      8u    /* entries of an ordered fallback list (eppk_pick_topk)        */
#define EPPK_INDEX_EPOCH_WINDOW 254u /* largest age, in index epochs, a live hash may reach (eppk_index_advance_epoch) */

#define EPPK_NO_PICK       (-1)  /* out_pick value: request had zero candidates (-> 503, fail closed) */
#define EPPK_ADAPTER_BASE  (-1)  /* req.adapter value: request targets the base model         */

typedef enum eppk_status {
  EPPK_OK              = 0,
  EPPK_ERR_ARG         = -1, /* null pointer / out-of-range argument / bad struct_size       */
  EPPK_ERR_LIMIT       = -2, /* exceeds a limit given at eppk_create                         */
  EPPK_ERR_DEVICE      = -3, /* HIP runtime error (message in eppk_last_error)               */
  EPPK_ERR_NO_SNAPSHOT = -4, /* pick before the first eppk_snapshot_publish                  */
  EPPK_ERR_INDEX_FULL  = -5, /* prefix table load factor exceeded; grow index_slots          */
  EPPK_ERR_NOMEM       = -6
} eppk_status;

/* Scorer plugins of the fused chain.  Replaces Scorer.Score (interface.go:120-126);
 * formulas in SEMANTICS.md §scorers. */
typedef enum eppk_scorer_kind {
  EPPK_SCORER_QUEUE  = 1, /* TotalQueuedRequests, lower is better   (003-…/README.md:30) */
  EPPK_SCORER_KV     = 2, /* KVCacheUtilization, lower is better    (003-…/README.md:32) */
  EPPK_SCORER_LORA   = 3, /* LoRA affinity tiers                    (003-…/README.md:46-57) */
  EPPK_SCORER_PREFIX = 4  /* prefix-cache block-hash match ratio    (0602-…/README.md:99-112) */
} eppk_scorer_kind;

/* One WeightedScorer{Scorer, weight int} (interface.go:132-135). */
typedef struct eppk_weighted_scorer {
  uint32_t kind;   /* eppk_scorer_kind */
  int32_t  weight; /* integer weight applied at profile level */
} eppk_weighted_scorer;

/* Context configuration == one SchedulerProfile (interface.go:70-79) with picker "best-score"
 * (examples/example.yaml:14-15) and a deterministic tie-break (lowest snapshot index). */
typedef struct eppk_cfg {
  uint32_t struct_size;   /* = sizeof(eppk_cfg) */
  int32_t  device;        /* HIP device ordinal */
  uint32_t max_pods;      /* <= EPPK_MAX_PODS */
  uint32_t max_blocks;    /* B: u64 hash slots in every request row (row stride = 8 + 8*B) */
  uint32_t max_batch;     /* largest n_reqs for the host-buffer entry point */
  uint32_t index_slots;   /* prefix table capacity: power of two in [64, 2^28], 0 = no prefix index.  Holds at most
                           * index_slots/2 live hashes (EPPK_ERR_INDEX_FULL beyond; the two reserved hash values 0 and ~0 have
                           * rows of their own behind the table and are always admitted); size it at >= 4x the expected
                           * number (load <= 0.25: about one hash per 64-byte bucket).  Memory: the library allocates TWO physical
                           * slots per index_slot (five of a bucket's eight words hold hashes, the other three their stamps and
                           * pod-set ids): index_slots * 2 * (8 + 64 + 64 * lane-word bytes) + a pod-set table of index_slots * 8
                           * bytes -- 1.2 KiB per index_slot at max_pods = 4096, almost all of it dense pod-set rows that are
                           * touched only by hashes cached on more than 24 pods */
  uint32_t n_scorers;     /* <= EPPK_MAX_SCORERS; order fixes the fp summation order */
  uint32_t reserved;
  eppk_weighted_scorer chain[EPPK_MAX_SCORERS];
} eppk_cfg;

/* One candidate endpoint's metrics: the row the scorers read.  The reference's Endpoint
 * (pkg/lwepp/datastore/datastore.go:40-46) carries identity only; these fields are the
 * model-server-protocol gauges (003-…/README.md:28-57).  64 bytes, natural (per-pod) layout;
 * the library re-lays it out for the device. */
typedef struct eppk_pod_row {
  uint32_t queue;       /* TotalQueuedRequests   */
  uint32_t running;     /* TotalRunningRequests (carried, not scored) */
  double   kv_util;     /* KVCacheUtilization in [0,1] */
  uint32_t max_lora;    /* max_lora label */
  uint32_t flags;       /* EPPK_POD_* bits; every other bit reserved, 0 */
  uint64_t active[2];   /* running_lora_adapters as a bitset over adapter ids 0..127 */
  uint64_t waiting[2];  /* waiting_lora_adapters bitset */
  uint64_t reserved;
} eppk_pod_row;

/* flags bit 0: the slot is a HOLE of this snapshot -- a candidate index that currently names no endpoint (an endpoint left and the
 * host keeps every other endpoint's index stable: datastore churn, pkg/lwepp/datastore/datastore.go:195-255).  A hole is never a
 * candidate, lies outside the QUEUE normalisers, and the prefix index forgets it: publishing a snapshot in which a slot became a
 * hole removes that slot from every pod set, and index inserts that name a hole are ignored (SEMANTICS.md §6b). */
#define EPPK_POD_INACTIVE 1u

#ifdef __cplusplus
static_assert(sizeof(eppk_pod_row) == 64, "eppk_pod_row must be 64 bytes");
#else
_Static_assert(sizeof(eppk_pod_row) == 64, "eppk_pod_row must be 64 bytes");
#endif

/* Request row header; followed in memory by max_blocks u64 chained block hashes
 * (hash[i] = H(block_i || LE64(hash[i-1])), 0602-…/README.md:99).  Row stride = 8 + 8*max_blocks. */
typedef struct eppk_req_hdr {
  int32_t  adapter;   /* adapter id 0..127, or EPPK_ADAPTER_BASE */
  uint32_t n_blocks;  /* number of valid hashes, <= max_blocks */
} eppk_req_hdr;

typedef struct eppk_ctx eppk_ctx;

/* ---- lifecycle ------------------------------------------------------------------------------ */

uint32_t    eppk_abi_version(void);
/* Create a picker bound to cfg->device.  Replaces the hard-wired `picker: &RoundRobinPicker{}`
 * in NewStreamingServer (handlers/server.go:43-48). */
int         eppk_create(const eppk_cfg* cfg, eppk_ctx** out);
void        eppk_destroy(eppk_ctx* ctx);
/* Last error text of this context (or of the last failed eppk_create when ctx == NULL). */
const char* eppk_last_error(const eppk_ctx* ctx);

/* ---- frozen pod-metrics snapshot ------------------------------------------------------------ */

/* Publish a frozen snapshot of n_pods rows; row i is candidate index i (the order of the
 * `endpoints` slice handed to Pick, handlers/server.go:90; the datastore's PodList order,
 * datastore.go:181-193).  Takes effect for every later pick.  epoch is caller bookkeeping.
 * The snapshot is double buffered: a publish builds the idle buffer, so picks already launched through the *_device entry
 * points on the caller's own streams keep reading the rows they started with -- across ONE publish; before a second
 * publish the caller must have synchronised those streams (the host-buffer entry points are synchronous and need nothing). */
int eppk_snapshot_publish(eppk_ctx* ctx, const eppk_pod_row* rows, uint32_t n_pods, uint64_t epoch);
int eppk_snapshot_info(const eppk_ctx* ctx, uint32_t* n_pods, uint64_t* epoch);

/* ---- approximate prefix index (0602-…/README.md:101-112), device resident ------------------- */

int eppk_index_clear(eppk_ctx* ctx);
/* "hash(chunk i): append s" for n (hash, pod) pairs. */
int eppk_index_insert(eppk_ctx* ctx, const uint64_t* hashes, const uint32_t* pods, uint32_t n);
/* Post-pick update on device: for every request r with picks[r] >= 0 append picks[r] to each of
 * its n_blocks hashes.  d_reqs / d_picks are DEVICE pointers (see eppk_pick_batch_device).  Three launches on `stream`: the
 * capacity verdict of the update, the update, and a pass that puts the pod lists it touched back into ascending order (equal pod
 * sets are then equal 64-byte lines again, which the pick kernels' fast routes test bit for bit).  Index updates of one context --
 * this call, eppk_index_evict_older_device, the synchronous entry points -- must be ORDERED with respect to each other (one
 * stream, or events between streams): they share the update's work list and capacity counters. */
int eppk_index_insert_picks_device(eppk_ctx* ctx, const void* d_reqs, const int32_t* d_picks,
                                   uint32_t n_reqs, void* stream);
/* Drop pod from every entry (endpoint deleted / cache flushed). */
int eppk_index_remove_pod(eppk_ctx* ctx, uint32_t pod);
/* Number of hashes with a non-empty pod set (synchronises). */
int eppk_index_size(eppk_ctx* ctx, uint32_t* n_entries);
/* (hash, pod) inserts dropped so far because the table was at its capacity (cumulative since create / clear; synchronises).
 * eppk_index_insert reports them as EPPK_ERR_INDEX_FULL; the asynchronous eppk_index_insert_picks_device cannot, so a shim
 * polls this counter (and grows or ages the index). */
int eppk_index_dropped(eppk_ctx* ctx, uint64_t* n_dropped);
/* Diagnostic: number of index slots that violate an internal invariant: a present hash with an empty pod set, a pod set left behind
 * a removed hash, a pod list that is not strictly ascending / holds an id twice / has entries behind its count, a set of at most
 * 24 pods that is not in its list (or whose dense row is not all-zero), a dense row with fewer than 25 pods, a key whose SET ID (the
 * name of its pod set in its bucket line: the pod itself for a single pod, else a line of the interned set table) disagrees with its
 * list.  0 on a healthy index; synchronous full scan.  EPPK_SELFCHECK_VERBOSE in
 * the environment prints the first eight offenders to stderr. */
int eppk_index_selfcheck(eppk_ctx* ctx, uint64_t* n_bad);
/* Ageing -- "mimicking a similar cache eviction strategy of the model server (e.g., LRU)", 0602-…/README.md:82.
 * Every insert (eppk_index_insert, eppk_index_insert_picks_device) stamps its hashes with the context's index epoch
 * (starts at 1); eppk_index_advance_epoch increments it; eppk_index_evict_older drops every hash whose last stamp is
 * < min_epoch (for all pods) and makes its table word reusable.  A shim ticks the epoch once per interval and evicts
 * `epoch - keep` to bound the index like the model servers' own LRU bounds their caches.
 * Window (SEMANTICS.md 6a): a hash may be at most EPPK_INDEX_EPOCH_WINDOW = 254 epochs old -- the tick to epoch e first evicts every
 * hash stamped before e - 254 (the device keeps a stamp as an 8-bit tag beside the hash, in its bucket line); a caller that never evicts,
 * or whose keep is 254 epochs and more, pays a table scan per tick from the 255th on (a shim clamps its keep to the window). */
int eppk_index_advance_epoch(eppk_ctx* ctx, uint32_t* new_epoch);
int eppk_index_evict_older(eppk_ctx* ctx, uint32_t min_epoch, uint32_t* n_evicted);
/* Per-pod capacity (0602-…/README.md:82: the approximate index mimics the model servers' own LRU-bounded prefix caches; upstream
 * bounds every pod's share of the index, SEMANTICS.md §6c).  A pod listed under more than `cap_per_pod` hashes is removed from
 * its OLDEST ones, at epoch granularity (whole epochs, oldest first; entries stamped in the current epoch always stay), until it
 * fits.  n_removed (nullable) = (hash, pod) pairs removed.  Synchronous; what a shim calls after a few epochs ticks. */
int eppk_index_trim_pods(eppk_ctx* ctx, uint32_t cap_per_pod, uint64_t* n_removed);
/* The same eviction, asynchronous on `stream` (a hipStream_t as void*; NULL = the context's stream) and without the count: what
 * a closed loop (pick -> eppk_index_insert_picks_device -> pick ...) issues on its own stream between two batches, ordered
 * behind the inserts and ahead of the next pick, without draining the pipeline.  eppk_index_size reports the effect.  Also valid while
 * staging sets of the pipelined host path are in flight (eppk_pick_stage_begin, below): ordered behind their picks and updates. */
int eppk_index_evict_older_device(eppk_ctx* ctx, uint32_t min_epoch, void* stream);

/* ---- the hot path --------------------------------------------------------------------------- */

/* Pick one endpoint for each of n_reqs requests against the published snapshot.
 * Batched replacement for EndpointPicker.Pick (handlers/server.go:79-82) as invoked by
 * pickEndpoint (request.go:149), evaluating Filter* -> Score* -> Picker
 * (0845-…/README.md:68-85) for the whole batch in one kernel.
 *   reqs      n_reqs rows of stride 8 + 8*max_blocks bytes (eppk_req_hdr + hashes)
 *   cand_mask nullable; else [n_reqs][ceil(n_pods/64)] u64, bit p%64 of word p/64 set = pod p is a
 *             candidate (the subset filter of request.go:104-133 as a bitmask)
 *   out_pick  [n_reqs] candidate index, or EPPK_NO_PICK when the request has no candidates
 *   out_score nullable; [n_reqs] weighted total of the picked endpoint (NaN-free; 0 for NO_PICK)
 * A request row out of range (n_blocks > max_blocks, adapter outside [-1, EPPK_MAX_ADAPTERS)) fails the whole call with
 * EPPK_ERR_ARG naming the lowest such row; nothing is written to out_pick / out_score.  (Round 3: the check runs on the device for
 * batches of more than EPPK_HOST_CHECK_MAX = 2048 rows -- a host loop over 64k row headers took longer than their PCIe transfer -- so
 * the batch HAS been scored when the error is returned; see the pipelined form below for what that means with EPPK_PICK_LEARN.) */
int eppk_pick_batch(eppk_ctx* ctx, const void* reqs, uint32_t n_reqs, const uint64_t* cand_mask,
                    int32_t* out_pick, double* out_score);

/* The context's pinned staging buffers (allocated on first use, max_batch rows each; *cand_mask only when asked for): a caller that
 * BUILDS its request rows -- and candidate mask rows, [n][ceil(n_pods / 64)] u64 -- right there saves the copy eppk_pick_batch makes of
 * pageable caller memory (that copy, not PCIe, is most of a 64k-request batch's 1.2 ms; INTEGRATION.md: the cgo dispatcher fills
 * C memory anyway).  Valid until eppk_destroy; not to be written while a eppk_pick_batch* call of this context is running. */
int eppk_host_staging(eppk_ctx* ctx, void** reqs, uint64_t** cand_mask);
/* eppk_pick_batch over the first n_reqs rows (and, with use_mask != 0, mask rows) of the staging buffers: same row check, same
 * results, no host copy.  Batches of at most EPPK_ZERO_COPY_MAX requests (environment, default 3072; 0 = never) are scored ZERO-COPY
 * by all host-buffer entry points: the kernel reads the pinned rows and writes the pinned results itself -- one launch, no upload or
 * download (a small batch is all latency: 128 requests 33 -> 22 us host-observed). */
int eppk_pick_batch_staged(eppk_ctx* ctx, uint32_t n_reqs, int use_mask, int32_t* out_pick, double* out_score);

/* The latency path of SMALL batches (opt-in: EPPK_RESIDENT=1 in the environment when the context is created).  What a per-request
 * caller hands over (pkg/lwepp/handlers/request.go:141-163; design point 10-1000 QPS, docs/proposals/006-scheduler/README.md:133) is
 * batches of a few dozen requests at most, and such a batch is all launch and completion latency.  With the switch on, a batch of at
 * most EPPK_RESIDENT_MAX requests (default 64; 32 where the four-requests-per-wavefront kernel does not apply) is handed to a RESIDENT
 * workgroup instead of a kernel launch: it polls a doorbell in pinned host memory, scores the batch with the same code as the launched
 * kernels (same picks, same scores), writes the pinned result buffers and raises a completion word the call is polling.  Every call
 * shape a dispatcher issues has its workgroup, started by the first batch that needs it:
 *     eppk_pick_batch / eppk_pick_batch_staged                 plain and MASKED single picks
 *     eppk_pick_topk                                          ordered fallbacks (k <= 8), with or without masks
 *     eppk_pick_stage_begin / _end                            both staging sets, plain and masked (begin rings, end polls), and with
 *                                                             EPPK_PICK_LEARN: the workgroup copies the rows, answers, and then applies
 *                                                             the post-route index update ITSELF (no launch); the next batch of the
 *                                                             context -- resident or launched -- is ordered behind that update.
 *                                                             On this path the two sets do NOT overlap: a begin first waits (spinning)
 *                                                             until the other set's batch is answered and, with LEARN, updated; and a
 *                                                             bad request row is reported by the BEGIN (host-side row check; the set
 *                                                             stays idle and must not be ended), where the launched path reports it
 *                                                             from the end with the row named
 * Host-observed, 16 requests, C5 snapshot (profiles/r05_resident_latency.txt): plain 12 us, masked 14-15, top-4 13, pick + LEARN 12
 * (23 back to back, the previous update included) against 20 / 28 / 22 / 30-40 us through launches.
 * Costs: a CU per workgroup alive (at most four: each runs on a high-priority stream, i.e. a hardware queue, of its own -- a kernel
 * that never ends would otherwise block whatever shares its queue; a fifth shape takes the launched path until it has been asked for
 * eight times, then the least recently rung workgroup makes room; the persistent pick kernels of the context are sized for four CUs
 * fewer), and a polling host thread for the duration of a call.
 * A workgroup leaves by itself after ~20-50 ms without a doorbell (EPPK_RESIDENT_IDLE_POLLS) and is started again by the next batch of
 * its kind; the library parks all of them in front of every device-wide wait of its own and in eppk_destroy.  Chains the fused kernel
 * does not serve, the random-top-k picker and assumed load take the launched path as before.
 * Streams: the resident workgroups are OUTSIDE stream order.  The library orders them against everything it has queued itself (the
 * context's own stream, the staging sets, LEARN updates, evictions); index or snapshot work a caller has queued on a stream of ITS OWN
 * (eppk_index_insert_picks_device / eppk_pick_learn_device / eppk_index_evict_older_device with a non-NULL stream) must have finished
 * before a small batch is handed over.
 * eppk_resident_stats: returns 1 when the switch is on (0 otherwise); batches = small batches answered by resident workgroups,
 * starts = times one was (re)started. */
int eppk_resident_stats(const eppk_ctx* ctx, uint64_t* batches, uint64_t* starts);

/* The PIPELINED host path: EPPK_STAGE_SETS staging sets, each with its own pinned rows / masks / results, device buffers and
 * stream, so that the rows of batch k + 1 cross PCIe while batch k is being scored (pkg/lwepp/handlers/request.go:141-163 is a
 * per-request caller: what it can hand over is host memory, and a 64k x 32-block batch is 17 MB -- 0.3 ms of PCIe against 20 us of
 * kernel; one synchronous batch at a time leaves the device idle for nine tenths of the time).  The caller alternates:
 *     fill set A; begin(A);  fill set B; begin(B);  end(A) -> results of A;  fill A; begin(A);  end(B); ...
 * eppk_pick_stage_buffers: the set's pinned request rows (max_batch rows) and mask rows (only when asked for), valid until
 *     eppk_destroy, not to be written between the set's begin and end.
 * eppk_pick_stage_begin: enqueues upload + row check + pick on the set's stream and returns (the kernel writes picks and scores into
 *     the set's pinned result buffers; at most EPPK_ZERO_COPY_MAX requests: no upload either, the kernel reads the pinned rows).  Rows
 *     out of range: at most EPPK_HOST_CHECK_MAX rows are checked here, on the host, and refused at once; larger batches are checked ON
 *     THE DEVICE and eppk_pick_stage_end fails with EPPK_ERR_ARG naming the lowest bad row and delivers nothing -- such a row was scored
 *     as EPPK_NO_PICK and a LEARN update has skipped it (the rest of the batch was learned).  flags & EPPK_PICK_LEARN chains the post-route index update (eppk_index_insert_picks_device: index[hash[r][i]] U=
 *     {pick[r]}, 0602-…/README.md:101-108) behind the pick ON THE DEVICE -- rows and picks are there already; the picks are on their
 *     way back to the host before the update starts.  A later begin (either set) scores against the index every earlier LEARN left
 *     behind: its pick waits for that update on the device, its upload does not.
 * eppk_pick_stage_end: waits for the set's picks and copies them out (the index update of a LEARN batch may still be running).
 * Errors: EPPK_ERR_ARG for a set that is busy (begin twice) or idle (end without begin).  A begin that FAILS has not begun: the set is
 * idle again (whatever it had enqueued has been waited for) and end must not be called for it.  One exception: the picks were launched
 * and only the chained LEARN update could not be enqueued -- begin then succeeds, end delivers the picks, and eppk_launch_status
 * reports EPPK_LAUNCH_LEARN_FAILED.  The update of a LEARN batch may still be running when end returns; every later pick, index
 * entry point and publish of the context orders itself behind it (on the device: no host wait), so a caller may issue them at once.
 * While a set is between begin and end, snapshot publishes and index entry points must still not be issued: its pick reads both --
 * with ONE exception, the ageing step of a router's closed loop: eppk_index_advance_epoch (host side only) followed by
 * eppk_index_evict_older_device may be issued at any time; the eviction queues ON THE DEVICE behind the pick (and LEARN update) of
 * every set begun before it and ahead of the pick of every set begun after it, i.e. the index ages in the order of the calls and
 * the two-set pipeline is never drained for it. */
#define EPPK_STAGE_SETS 2u
#define EPPK_PICK_LEARN 1u
int eppk_pick_stage_buffers(eppk_ctx* ctx, uint32_t set, void** reqs, uint64_t** cand_mask);
int eppk_pick_stage_begin(eppk_ctx* ctx, uint32_t set, uint32_t n_reqs, int use_mask, uint32_t flags);
int eppk_pick_stage_end(eppk_ctx* ctx, uint32_t set, int32_t* out_pick, double* out_score);

/* Same, with every buffer already resident in this device's HBM; asynchronous on `stream`
 * (a hipStream_t passed as void*; NULL = the context's own NON-BLOCKING stream, which is not ordered against the
 * legacy default stream — callers that mix this with other GPU work pass their own stream).  No n_reqs limit. */
int eppk_pick_batch_device(eppk_ctx* ctx, const void* d_reqs, uint32_t n_reqs,
                           const uint64_t* d_cand_mask, int32_t* d_out_pick, double* d_out_score,
                           void* stream);
/* Pick + learn in ONE call: eppk_pick_batch_device followed by eppk_index_insert_picks_device on `stream` -- Scheduler.Schedule() and the
 * post-route step of the prefix scorer ("hash(chunk i): append s", docs/proposals/0602-prefix-cache-aware-routing-proposal/README.md:101-108)
 * for a whole batch, same picks, same scores and the same index afterwards as the two calls.  What it adds: the pick kernel has just
 * walked every request's prefix through the index, so it tells the update which (hash, pod) pairs it has already SEEN there -- the
 * blocks of the shared prefix with the picked pod on their list: half of a 64k x 32-block batch's 2 Mi pairs -- and the update only
 * refreshes their stamps.  (Batches that do not take the four-requests-per-wavefront kernel get the plain update.)  The closed loop a
 * router runs: pick -> the index learns the pick -> next batch.  EPPK_PICK_LEARN of the staged host path does the same. */
/* (The learn words live in one buffer per context: calls on DIFFERENT streams are ordered by the library -- the later pick waits, on the
 * device, for the earlier call's update.) */
int eppk_pick_learn_device(eppk_ctx* ctx, const void* d_reqs, uint32_t n_reqs, const uint64_t* d_cand_mask, int32_t* d_out_pick,
                           double* d_out_score, void* stream);
/* Trust contract of the *_device entry points.  The host-buffer entry points check every request row and fail the call with
 * EPPK_ERR_ARG naming the row (above).  The *_device entry points return before the rows are looked at, so the KERNELS check them: a row whose
 * n_blocks exceeds max_blocks or whose adapter lies outside [-1, EPPK_MAX_ADAPTERS) is not scored -- its pick (every entry of its
 * fallback list) is EPPK_NO_PICK, its score 0.0 -- and eppk_index_insert_picks_device ignores a pick >= max_pods (and the picks of
 * such rows); either event sets a sticky flag that eppk_launch_status reports.  Nothing is read or written out of bounds and no
 * input is silently truncated (SEMANTICS.md §7).
 * Index maintenance (eppk_index_insert / _remove_pod / _evict_older / _clear, and a second eppk_snapshot_publish) synchronises the
 * context's own stream only: it must not run while picks launched through a *_device entry point on a CALLER's stream are still
 * in flight -- synchronise those streams first (eppk_index_insert_picks_device is ordered by the stream it is given). */
#define EPPK_LAUNCH_BAD_REQUEST_ROW 1u   /* a request row on a *_device entry point was out of range: it got EPPK_NO_PICK */
#define EPPK_LAUNCH_BAD_PICK        2u   /* eppk_index_insert_picks_device met a pick >= max_pods: ignored */
#define EPPK_LAUNCH_INDEX_STALL     4u   /* an index insert gave up waiting for the first pod of a key that another thread of the same
                                          * launch had just claimed (never observed; the wait is bounded so that a broken index cannot
                                          * hang the device): that (hash, pod) pair was dropped */
#define EPPK_LAUNCH_LEARN_FAILED    8u   /* eppk_pick_stage_begin(EPPK_PICK_LEARN) launched the picks but could not enqueue the post-route index
                                          * update behind them (eppk_last_error says why): the picks were delivered, the index did not learn them */
/* Synchronise the device and return (and clear) the sticky launch-status flags accumulated by every *_device launch of this
 * context since the last call.  0 = every row was in range. */
int eppk_launch_status(eppk_ctx* ctx, uint32_t* flags);

/* Make `waiting_stream` (a hipStream_t) wait for the most recent pick launch of this context -- a cross-stream dependency
 * without a separate event record behind the kernel when the launch already carries a completion event (profiling on).
 * What a caller uses to start a collective or a copy of the picks on another stream (bench.py: the RCCL all-gather). */
int eppk_stream_wait_pick(eppk_ctx* ctx, void* waiting_stream);

/* Ordered fallbacks: the k (1..EPPK_MAX_TOPK) best candidates of every request under the picker's order (weighted total
 * descending, candidate index ascending).  out_pick[r*k + 0] is exactly eppk_pick_batch's pick, out_pick[r*k + i] the
 * i-th fallback; a request with fewer than k candidates is padded with EPPK_NO_PICK / 0.0.
 * Replaces PickResult.Fallbacks (handlers/server.go:72-77); the protocol carries them as an ordered, comma-separated
 * endpoint list (docs/proposals/004-endpoint-picker-protocol/README.md:73).
 * out_pick / out_score (nullable) hold n_reqs * k entries. */
int eppk_pick_topk(eppk_ctx* ctx, const void* reqs, uint32_t n_reqs, const uint64_t* cand_mask, uint32_t k,
                   int32_t* out_pick, double* out_score);
/* Same with device-resident buffers, asynchronous on `stream` (see eppk_pick_batch_device). */
int eppk_pick_topk_device(eppk_ctx* ctx, const void* d_reqs, uint32_t n_reqs, const uint64_t* d_cand_mask, uint32_t k,
                          int32_t* d_out_pick, double* d_out_score, void* stream);

/* Picker "random-top-k" (0845-…/examples/example.yaml:25 `selection: random-top-3`; SEMANTICS.md §3b): the pick of request r is
 * entry (splitmix64(seed + (r+1) * 0x9E3779B97F4A7C15) mod n) of its ordered fallback list of n <= k candidates -- a seeded,
 * reproducible stand-in for "random" (north_star: deterministic picks).  out_pick / out_score hold n_reqs entries. */
int eppk_pick_random_topk(eppk_ctx* ctx, const void* reqs, uint32_t n_reqs, const uint64_t* cand_mask, uint32_t k, uint64_t seed,
                          int32_t* out_pick, double* out_score);
int eppk_pick_random_topk_device(eppk_ctx* ctx, const void* d_reqs, uint32_t n_reqs, const uint64_t* d_cand_mask, uint32_t k, uint64_t seed,
                                 int32_t* d_out_pick, double* d_out_score, void* stream);

/* Assumed load (docs/proposals/006-scheduler/README.md:154-156; SEMANTICS.md §2b).  epochs = E >= 1: every batch handed to a pick
 * entry point of this context is scored in E epochs of ceil(n_reqs / E) consecutive requests; after each epoch the queue gauge of
 * every picked endpoint grows by one per request routed to it and everything derived from the gauge is rebuilt on the device
 * (four small launches, ~0.2 ms at 4096 pods) before the next epoch.  The bumped gauges persist until the next
 * eppk_snapshot_publish.  epochs = 0 (default): off -- every batch sees the published gauges.
 * While it is on, the context's picks must be issued on ONE stream at a time (each epoch flips the double-buffered snapshot);
 * device groups do not support it (every member would bump only its own shard). */
int eppk_set_assumed_load(eppk_ctx* ctx, uint32_t epochs);

/* ---- device groups: one picker over several GPUs (SURVEY.md §8(b) "device list", §8(e)) ------------------------------ */

/* A group replicates the snapshot and the prefix index on every member device and shards each batch BY REQUEST: member g scores
 * rows [g*per, (g+1)*per), per = ceil(n_reqs / devices used).  There is no data-path collective -- a pick depends only on its own
 * request row and the replicated read-only state -- and every device returns its shard of picks to the host over its own PCIe
 * link.  The one exchange of the path, an all-gather of the int32 picks so that every DEVICE holds all of them, runs only when
 * something on the devices needs them: EPPK_GROUP_LEARN (each device applies the same post-route index update to its replica,
 * SEMANTICS.md §6) or EPPK_GROUP_GATHER.  The same calling rules as a context: one caller at a time.
 * What a Go host links instead of torch.distributed: INTEGRATION.md §5. */
typedef struct eppk_group eppk_group;
#define EPPK_GROUP_MAX_DEVICES 16u
/* how the picks are all-gathered on the devices */
#define EPPK_GATHER_PEER 0u   /* every device pushes its shard into each peer's array (hipMemcpyPeerAsync: one xGMI hop per peer) */
#define EPPK_GATHER_RCCL 1u   /* ncclAllGather, in place, one communicator rank per device in this process (librccl is dlopen'ed);
                               * needs distinct devices and shards every batch over ALL members */
#define EPPK_GATHER_HOST 2u   /* the picks the devices return to the host are uploaded back to every device (no peer traffic) */
/* eppk_group_pick_batch flags */
#define EPPK_GROUP_LEARN  1u  /* after the picks: index[hash[r][i]] U= {pick[r]} on EVERY member, from the gathered picks */
#define EPPK_GROUP_GATHER 2u  /* all-gather the picks on the devices even without LEARN (eppk_group_device_picks) */

/* One member context per entry of `devices` (HIP ordinals; cfg->device is ignored; an ordinal may repeat -- two members on one GPU
 * is how a one-GPU box tests the sharding).  cfg->max_batch bounds the WHOLE batch. */
int         eppk_group_create(const eppk_cfg* cfg, const int32_t* devices, uint32_t n_devices, uint32_t gather_mode, eppk_group** out);
void        eppk_group_destroy(eppk_group* g);
const char* eppk_group_last_error(const eppk_group* g);       /* g == NULL: the last failed eppk_group_create */
uint32_t    eppk_group_size(const eppk_group* g);
eppk_ctx*   eppk_group_ctx(eppk_group* g, uint32_t i);        /* member i (diagnostics; per-device calls such as eppk_index_size) */
/* Ranks the collective layer sees: the communicator's size under EPPK_GATHER_RCCL (ncclCommCount), else the member count. */
int         eppk_group_ranks_seen(const eppk_group* g);
/* A batch is spread over ceil(n_reqs / min_shard) members at most (default 2048): a small batch stays on one GPU. */
int         eppk_group_set_min_shard(eppk_group* g, uint32_t min_shard);
/* Replicated state: the same call on every member (eppk_snapshot_publish, eppk_index_*). */
int eppk_group_snapshot_publish(eppk_group* g, const eppk_pod_row* rows, uint32_t n_pods, uint64_t epoch);
int eppk_group_index_clear(eppk_group* g);
int eppk_group_index_insert(eppk_group* g, const uint64_t* hashes, const uint32_t* pods, uint32_t n);
int eppk_group_index_remove_pod(eppk_group* g, uint32_t pod);
int eppk_group_index_advance_epoch(eppk_group* g, uint32_t* new_epoch);
int eppk_group_index_evict_older(eppk_group* g, uint32_t min_epoch, uint32_t* n_evicted);
/* Ageing and per-pod capacity of the replicated index as a context has them: eppk_index_evict_older_device on every member's own
 * stream (valid while staging sets of eppk_group_pick_stage_* are in flight: behind the picks and LEARN updates begun before the call,
 * ahead of those begun after it -- the shim's ageing step does not drain the group's pipeline either), eppk_index_trim_pods on every
 * member (replicas are identical: n_removed is any member's count). */
int eppk_group_index_evict_older_device(eppk_group* g, uint32_t min_epoch);
int eppk_group_index_trim_pods(eppk_group* g, uint32_t cap_per_pod, uint64_t* n_removed);
/* eppk_pick_batch over the group: same arguments and results (out_pick / out_score hold all n_reqs entries, in request order). */
int eppk_group_pick_batch(eppk_group* g, const void* reqs, uint32_t n_reqs, const uint64_t* cand_mask, int32_t* out_pick,
                          double* out_score, uint32_t flags);
/* Ordered fallbacks (PickResult.Fallbacks, pkg/lwepp/handlers/server.go:72-77; docs/proposals/004-endpoint-picker-protocol/README.md:73)
 * and the picker "random-top-k" over the group: eppk_pick_topk / eppk_pick_random_topk with the batch sharded by request as above,
 * every member busy at once; same arguments, same results as the context calls on the unsharded batch (random-top-k hashes a request's
 * index in the BATCH, not in its shard). */
int eppk_group_pick_topk(eppk_group* g, const void* reqs, uint32_t n_reqs, const uint64_t* cand_mask, uint32_t k, int32_t* out_pick,
                         double* out_score);
int eppk_group_pick_random_topk(eppk_group* g, const void* reqs, uint32_t n_reqs, const uint64_t* cand_mask, uint32_t k, uint64_t seed,
                                int32_t* out_pick, double* out_score);
/* The PIPELINED host path over the group: eppk_pick_stage_buffers / _begin / _end (above) with the same calling rules, flags
 * (EPPK_PICK_LEARN) and error behaviour.  A set's rows / masks / results live in ONE pinned buffer of the group that every member's DMA
 * engine reads; member i uploads and scores its shard on its own staging set's stream while the caller fills the other set.  With
 * EPPK_PICK_LEARN every member uploads the WHOLE batch, the picks of all shards are gathered onto every member (the group's gather
 * mode: peer pushes, one ncclAllGather, or -- EPPK_GATHER_HOST -- uploaded again by _end once they have reached the host) and each member
 * chains the SAME post-route update to its replica behind them; a later begin of either set scores against that index on every
 * member.  Ageing between two begins: eppk_group_index_advance_epoch + eppk_group_index_evict_older_device. */
int eppk_group_pick_stage_buffers(eppk_group* g, uint32_t set, void** reqs, uint64_t** cand_mask);
int eppk_group_pick_stage_begin(eppk_group* g, uint32_t set, uint32_t n_reqs, int use_mask, uint32_t flags);
int eppk_group_pick_stage_end(eppk_group* g, uint32_t set, int32_t* out_pick, double* out_score);
/* The DEVICE-RESIDENT form: member i scores the n_rows[i] request rows at d_reqs[i] -- already in ITS device memory (what a
 * producer that hashes prompts on the device leaves there, eppk_hash_prompts_device) -- into d_out_pick[i] / d_out_score[i] (its
 * memory too; d_out_score nullable), enqueued on the member's own stream (eppk_group_stream) without touching the host: no staging,
 * no synchronisation -- eppk_group_sync waits for every member.  A shard may hold the member's rows of SEVERAL batches back to back
 * (one launch per member for a whole gather bucket: what makes strong scaling of 64k-request batches over 8 devices launch-bound
 * otherwise).  flags: EPPK_GROUP_GATHER -> every member also receives every other member's picks, member-major, in
 * d_gathered[i] (sum of n_rows entries; member j's picks at offset n_rows[0] + ... + n_rows[j-1]): peer copies over xGMI
 * (EPPK_GATHER_PEER) or one ncclAllGather (EPPK_GATHER_RCCL: equal shards); each member's stream continues when its array is
 * complete.  EPPK_GROUP_LEARN is refused here (it needs the whole batch on every member: eppk_group_pick_batch). */
int   eppk_group_pick_device(eppk_group* g, const void* const* d_reqs, const uint32_t* n_rows, int32_t* const* d_out_pick,
                             double* const* d_out_score, int32_t* const* d_gathered, uint32_t flags);
int   eppk_group_sync(eppk_group* g);
void* eppk_group_stream(eppk_group* g, uint32_t i);      /* member i's hipStream_t */
/* Device pointer to member i's copy of the gathered picks of the last LEARN / GATHER batch (n_reqs entries, request order). */
const int32_t* eppk_group_device_picks(eppk_group* g, uint32_t i);

/* ---- adjacent host-side steps of the same path ---------------------------------------------- */

/* Chain-hash a prompt into block hashes (0602-…/README.md:99): XXH64, seed 0,
 * h[-1] = XXH64(model), h[i] = XXH64(prompt[i*block_chars .. (i+1)*block_chars) || LE64(h[i-1])).
 * Only full blocks are hashed.  Returns the number of hashes written (<= max_out) or <0. */
int eppk_hash_prompt(const uint8_t* model, size_t model_len, const uint8_t* prompt, size_t prompt_len,
                     uint32_t block_chars, uint64_t* out, uint32_t max_out);
uint64_t eppk_xxh64(const void* data, size_t len, uint64_t seed);

/* The same chain for a whole batch ON THE DEVICE, writing complete request rows (header + hashes) that
 * eppk_pick_batch_device consumes — no host hashing on the critical path.
 *   d_prompts     prompt r at d_prompts + r*prompt_stride (8-byte aligned, prompt_stride % 8 == 0)
 *   d_prompt_len  u32[n]: bytes of prompt r (<= prompt_stride); only full blocks are hashed
 *   d_seed        u64[n]: h[-1] of request r = eppk_xxh64(model name) (one host hash per model)
 *   d_adapter     i32[n]: adapter id or EPPK_ADAPTER_BASE
 *   block_chars   multiple of 8
 *   d_reqs_out    n rows of stride 8 + 8*max_blocks
 * All pointers are device pointers; asynchronous on `stream` (NULL = the context's stream). */
int eppk_hash_prompts_device(eppk_ctx* ctx, const void* d_prompts, uint64_t prompt_stride, const uint32_t* d_prompt_len,
                             const uint64_t* d_seed, const int32_t* d_adapter, uint32_t n_reqs, uint32_t block_chars,
                             void* d_reqs_out, void* stream);

/* Candidate subset filter of handleRequestHeaders (request.go:104-133) as a bitmask builder.
 *   addrs/ports  n_pods C strings: Endpoint.Address / Endpoint.Port (datastore.go:43-44)
 *   filter       comma-separated "ip" (all ports) or "ip:port" entries, whitespace-trimmed
 *   out_mask     ceil(n_pods/64) u64 words
 * Returns the number of candidates (0 = fail closed), or <0. */
int eppk_subset_mask(const char* const* addrs, const char* const* ports, uint32_t n_pods,
                     const char* filter, uint64_t* out_mask);

/* ---- the subset filter for a whole batch, resolved on the device (SEMANTICS.md §8) -------------------
 * request.go:104-133 walks every pod of the datastore per request (two map look-ups each): O(pods) host work per request,
 * and a host-built mask costs pods/8 bytes of H2D per request.  Here the host only TOKENISES a request's filter into entries
 * (trim, SplitHostPort -- string work stays on the host) and fingerprints each entry (128 bits: XXH64 under two seeds of
 * "host" for an all-ports entry, of "host\0port" for an exact one); the device holds the fingerprints of the published
 * endpoints in a hash table and turns a batch of entry lists into the candidate-mask rows the pick kernels read.
 *
 *   eppk_subset_entries          filter (NULL = the request has no filter: one entry (0,0) that admits every pod; "" = a filter
 *                                that is present but empty: zero entries, zero candidates, fail closed) -> out_keys[cap][2].
 *                                Returns the number of entries the filter has (write at most cap; retry when it exceeds cap).
 *   eppk_snapshot_set_addresses  Endpoint.Address / Endpoint.Port (datastore.go:43-44) of the slots of the CURRENT snapshot
 *                                (n_pods must match it; NULL address = a hole).  Call after EVERY eppk_snapshot_publish: a publish
 *                                invalidates the table (it may have re-mapped slots; the device filter then answers
 *                                EPPK_ERR_NO_SNAPSHOT instead of resolving against stale addresses).
 *   eppk_subset_masks[_device]   entries of n_reqs requests in CSR form (keys[off[r] .. off[r+1])) -> mask rows
 *                                [n_reqs][ceil(n_pods/64)] u64, the layout of eppk_pick_batch*'s cand_mask.
 *   eppk_pick_batch_subset       eppk_pick_batch with the masks built on the device from entry lists (host buffers in, picks out).
 *   eppk_pick_batch_candidates_device   the results of eppk_pick_batch_device (k = 1) / eppk_pick_topk_device (k > 1) for a MASKED
 *                                batch, from a kernel whose cost grows with the number of CANDIDATES of a request instead of the
 *                                number of pods: the one to call when the masks leave a few dozen candidates (a subset hint);
 *                                the general entry points stay better for dense masks.  eppk_pick_batch_subset chooses by itself,
 *                                and batches of 8192 requests or more (k = 1; ordered fallbacks: 4096 or more) are handed to
 *                                eppk_pick_batch_device / eppk_pick_topk_device, whose kernels score such rows faster at that size
 *                                (same picks, same scores).
 * Two different addresses with the same 128-bit fingerprint would be confused (probability ~ 2^-100 per pair); the
 * string-exact eppk_subset_mask stays available. */
int eppk_pick_batch_candidates_device(eppk_ctx* ctx, const void* d_reqs, uint32_t n_reqs, const uint64_t* d_cand_mask, uint32_t k,
                                      int32_t* d_out_pick, double* d_out_score, void* stream);
void eppk_addr_fingerprint(const char* host, size_t host_len, const char* port /* NULL = all ports */, size_t port_len, uint64_t out[2]);
int eppk_subset_entries(const char* filter, uint64_t* out_keys, uint32_t cap);
int eppk_snapshot_set_addresses(eppk_ctx* ctx, const char* const* addrs, const char* const* ports, uint32_t n_pods);
int eppk_subset_masks_device(eppk_ctx* ctx, const uint64_t* d_keys, const uint32_t* d_off, uint32_t n_reqs, uint64_t* d_mask_out, void* stream);
int eppk_subset_masks(eppk_ctx* ctx, const uint64_t* keys, const uint32_t* off, uint32_t n_reqs, uint64_t* out_mask);
int eppk_pick_batch_subset(eppk_ctx* ctx, const void* reqs, uint32_t n_reqs, const uint64_t* keys, const uint32_t* off,
                           int32_t* out_pick, double* out_score);

/* RoundRobinPicker.Pick (handlers/server.go:90-101): idx = atomic(++*counter) % n_candidates.
 * The fail-open fallback of the shim.  Returns the index, or EPPK_NO_PICK when n_candidates == 0. */
int32_t eppk_round_robin(uint64_t* counter, uint32_t n_candidates);

/* ---- measurement hooks (bench.py) ----------------------------------------------------------- */

/* Which kernel serves this context's chain: 1 = fused sparse kernel (chain = pod-only scorers, then LORA / PREFIX in either
 * order), 2 = the same kernel with an interpreted tail (at most two pod-only scorers behind the first LORA / PREFIX, e.g. the
 * scorer list of the reference example's decode profile, `score: [prefix-cache: 3, kv-cache-util: 5]`, 0845-…/examples/example.yaml:21-23
 * -- that profile's picker is `random-top-3`, :25: eppk_pick_random_topk), 0 = generic
 * per-pair kernel (duplicated LORA / PREFIX scorers, more than two trailing pod-only scorers). */
int eppk_chain_is_fused(const eppk_ctx* ctx);

/* Diagnostic.  Unmasked single-pick batches of a fused chain with a PREFIX scorer (max_blocks <= 63) go through the
 * four-requests-per-wavefront kernel first (csrc/eppk_kernels.hip.h: pick_quad_kernel); a request outside its common shape
 * (differing or overflowed pod lists, more than 32 cached blocks, reserved hashes, an out-of-range row ...) is DEFERRED to the general
 * kernel -- by the LAST workgroup of the same launch while the recent batches deferred next to nothing (the one-launch form), by a
 * second launch right behind it on the same stream otherwise -- same picks and scores either way.  Synchronises the device and
 * returns how many pick launches took that route and how many requests they deferred.  (Environment: EPPK_QUAD=0 switches the route
 * off; EPPK_QUAD_MIN = smallest batch that takes it, default 4096 requests; EPPK_QUAD_TAIL=0 keeps the two-launch form.  A workload
 * that keeps deferring a large part of its batches pauses the route by itself.) */
int eppk_quad_stats(eppk_ctx* ctx, uint64_t* launches, uint64_t* deferred);

/* Enabling resets the event ring, the launch counter and the probe statistics.  While enabled,
 * every pick launch (on == 1) -- or every on-th one (on > 1: sampled, what a throughput measurement uses so that the
 * instrumentation is not part of what it times) -- is bracketed by HIP events recorded on the launch stream and accumulates its
 * index-probe counts on device (one atomic per wavefront); eppk_profile_drain / eppk_profile_bytes report the sampled launches. */
int eppk_profile_enable(eppk_ctx* ctx, int on);
/* Synchronise and copy out up to cap kernel durations (ms) recorded since the last drain. */
int eppk_profile_drain(eppk_ctx* ctx, float* ms, uint32_t cap, uint32_t* n_out);
/* Algorithmic (compulsory) bytes of all pick launches since profiling was enabled — SURVEY §8(d)
 * byte model, with the probe counts the sequential walk of SEMANTICS.md §3 needs (measured on
 * device): launches * (n_pods*64 + n_reqs*(stride+4)) + hits*(8 + pod-set row) + misses*8.
 * lookups = index look-ups of that walk; launches = pick launches counted.  Synchronises. */
int eppk_profile_bytes(eppk_ctx* ctx, uint64_t* bytes, uint64_t* lookups, uint32_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* EPPK_H */
