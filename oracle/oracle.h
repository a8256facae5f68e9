/*
 * oracle.h — CPU restatement of the endpoint-pick path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
 * product (libeppk) never links, loads or calls it.
 *
 * PARITY UNPINNED: /root/reference no longer contains the scorer/scheduler source nor any golden
 * vector for it.  This file restates SEMANTICS.md, whose rules cite the
 * reference's specs; what the reference does pin (round robin, subset filter) is restated from
 * pkg/lwepp/handlers/server.go:90-101 and request.go:104-133 and checked against the cases of
 * pkg/lwepp/handlers/request_test.go:50-551 in tests/test_pinned_behaviour.py.
 */
#ifndef EPPK_ORACLE_H
#define EPPK_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#include "../include/eppk.h" /* row formats only (eppk_pod_row, eppk_req_hdr, eppk_weighted_scorer) */

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_index orc_index;

orc_index* orc_index_new(void);
void       orc_index_free(orc_index* ix);
void       orc_index_clear(orc_index* ix);
/* "hash(chunk i): append s" — docs/proposals/0602-…/README.md:101-108 (set semantics). */
void       orc_index_insert(orc_index* ix, uint64_t hash, uint32_t pod);
void       orc_index_remove_pod(orc_index* ix, uint32_t pod);
/* the index side of publishing `pods` (SEMANTICS.md 6b): every slot that is a hole (flags & EPPK_POD_INACTIVE) is forgotten */
void       orc_index_scrub_inactive(orc_index* ix, const eppk_pod_row* pods, uint32_t n_pods);
/* ageing: ++epoch (inserts stamp their hash with it); drop every hash last stamped before min_epoch -> number dropped */
uint32_t   orc_index_advance_epoch(orc_index* ix);
uint32_t   orc_index_evict_older(orc_index* ix, uint32_t min_epoch);
/* SEMANTICS.md 6c: per-pod capacity, oldest epochs first; returns the (hash, pod) pairs removed */
uint64_t   orc_index_trim_pods(orc_index* ix, uint32_t n_pods_max, uint32_t cap);
/* number of hashes with a non-empty pod set */
uint64_t   orc_index_size(const orc_index* ix);
/* copy out the pod set of one hash (sorted ascending); returns its size */
uint32_t   orc_index_lookup(const orc_index* ix, uint64_t hash, uint32_t* pods, uint32_t cap);

/* One Schedule() per request, sequentially: Filter -> Score* -> Picker
 * (docs/proposals/0845-…/README.md:68-85; interface.go:113-142).  SEMANTICS.md §2-3.
 * out_probes (nullable): index lookups performed per request (for the byte model). */
int orc_pick_batch(const eppk_weighted_scorer* chain, uint32_t n_scorers,
                   const eppk_pod_row* pods, uint32_t n_pods, const orc_index* ix,
                   const void* reqs, uint32_t max_blocks, uint32_t n_reqs, const uint64_t* cand_mask,
                   int32_t* out_pick, double* out_score, uint32_t* out_probes);

/* Same work, requests block-partitioned over `threads` pthreads (CPU baseline, all cores). */
int orc_pick_batch_mt(const eppk_weighted_scorer* chain, uint32_t n_scorers,
                      const eppk_pod_row* pods, uint32_t n_pods, const orc_index* ix,
                      const void* reqs, uint32_t max_blocks, uint32_t n_reqs, const uint64_t* cand_mask,
                      int32_t* out_pick, double* out_score, int threads);

/* The same decisions by a second algorithm (snapshot tables + the pods each request's prefix walk names): what bench.py times as
 * the CPU baseline.  Unmasked batches.  `pods` is borrowed by the tables and must outlive them; the tables cache per-adapter classes
 * across batches of one snapshot.  Bit-exact against orc_pick_batch (tests/test_oracle_golden.py). */
typedef struct orc_tables orc_tables;
orc_tables* orc_tables_new(const eppk_weighted_scorer* chain, uint32_t n_scorers, const eppk_pod_row* pods, uint32_t n_pods);
void        orc_tables_free(orc_tables* tb);
int orc_pick_batch_sparse(orc_tables* tb, const orc_index* ix, const void* reqs, uint32_t max_blocks, uint32_t n_reqs,
                          int32_t* out_pick, double* out_score, int threads);

/* SEMANTICS.md 2b: the batch in `epochs` sub-batches; after each, queue[pick] += 1 in `pods` (MUTATED: the caller keeps them
 * for the next batch of the same snapshot).  epochs == 0: orc_pick_batch (no bump). */
int orc_pick_batch_assumed(const eppk_weighted_scorer* chain, uint32_t n_scorers,
                           eppk_pod_row* pods, uint32_t n_pods, const orc_index* ix,
                           const void* reqs, uint32_t max_blocks, uint32_t n_reqs, const uint64_t* cand_mask,
                           uint32_t epochs, int32_t* out_pick, double* out_score);

/* SEMANTICS.md 3b: picker "random-top-k" -- entry (splitmix64(seed + (r+1)*golden) mod n) of the request's ordered fallback
 * list of at most k (<= 8) candidates. */
int orc_pick_random_topk(const eppk_weighted_scorer* chain, uint32_t n_scorers,
                         const eppk_pod_row* pods, uint32_t n_pods, const orc_index* ix,
                         const void* reqs, uint32_t max_blocks, uint32_t n_reqs, const uint64_t* cand_mask,
                         uint32_t k, uint64_t seed, int32_t* out_pick, double* out_score);

/* SEMANTICS.md 2a: ordered fallbacks -- the k (<= 8) best candidates of every request under (total descending, index ascending),
 * [n_reqs][k] entries padded with EPPK_NO_PICK / 0.0; `threads` host threads over request ranges. */
int orc_pick_topk(const eppk_weighted_scorer* chain, uint32_t n_scorers,
                  const eppk_pod_row* pods, uint32_t n_pods, const orc_index* ix,
                  const void* reqs, uint32_t max_blocks, uint32_t n_reqs, const uint64_t* cand_mask,
                  uint32_t k, int threads, int32_t* out_pick, double* out_score);

/* Full weighted totals of ONE request over all pods (non-candidates get NaN); debugging aid. */
int orc_score_row(const eppk_weighted_scorer* chain, uint32_t n_scorers,
                  const eppk_pod_row* pods, uint32_t n_pods, const orc_index* ix,
                  const void* req, const uint64_t* cand_mask_row, double* out_total);

/* post-pass of SEMANTICS.md §6 */
void orc_index_insert_picks(orc_index* ix, const void* reqs, uint32_t max_blocks, uint32_t n_reqs,
                            const int32_t* picks);

/* XXH64 and the block-hash chain (SEMANTICS.md §4) */
uint64_t orc_xxh64(const void* data, size_t len, uint64_t seed);
int      orc_hash_prompt(const uint8_t* model, size_t model_len, const uint8_t* prompt, size_t prompt_len,
                         uint32_t block_chars, uint64_t* out, uint32_t max_out);

/* pinned behaviour: handlers/server.go:90-101 and request.go:104-133 */
int32_t orc_round_robin(uint64_t* counter, uint32_t n_candidates);
int     orc_subset_mask(const char* const* addrs, const char* const* ports, uint32_t n_pods,
                        const char* filter, uint64_t* out_mask);

#ifdef __cplusplus
}
#endif
#endif
