/*
 * oracle.c — CPU restatement of the endpoint-pick path.  TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * PARITY UNPINNED for the scorer chain: restates SEMANTICS.md.
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (oracle/Makefile) — binary64 multiply and add
 * are never fused, like Go on amd64.
 *
 * Shape: one sequential Schedule() per request, exactly the per-request loop the batched kernel
 * replaces (docs/proposals/0845-scheduler-architecture-proposal/README.md:68-85):
 *     Filter* -> for each WeightedScorer: Score() then weighted accumulate -> Picker.
 * Data structures are deliberately the naive ones (per-pod rows, hash -> sorted pod list), NOT the
 * device layouts, so that agreement with the kernel is evidence and not a tautology.
 */
#define _GNU_SOURCE /* qsort_r */
#include "oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------- */
/* XXH64 — the published xxHash64 algorithm (SEMANTICS.md §4).  The reference names the Go module
 * github.com/cespare/xxhash/v2 v2.3.0 only as an indirect dependency (go.mod:6), no call sites.  */

static const uint64_t XP1 = 11400714785074694791ULL;
static const uint64_t XP2 = 14029467366897019727ULL;
static const uint64_t XP3 = 1609587929392839161ULL;
static const uint64_t XP4 = 9650029242287828579ULL;
static const uint64_t XP5 = 2870177450012600261ULL;

static uint64_t rd64(const uint8_t* p) {
  uint64_t v = 0;
  for (int i = 7; i >= 0; --i) v = (v << 8) | p[i];
  return v;
}
static uint32_t rd32(const uint8_t* p) {
  return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
static uint64_t rol(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
static uint64_t xround(uint64_t acc, uint64_t lane) { return rol(acc + lane * XP2, 31) * XP1; }
static uint64_t xmerge(uint64_t h, uint64_t acc) { return (h ^ xround(0, acc)) * XP1 + XP4; }

uint64_t orc_xxh64(const void* data, size_t len, uint64_t seed) {
  const uint8_t* p = (const uint8_t*)data;
  const uint8_t* end = p + len;
  uint64_t h;
  if (len >= 32) {
    uint64_t a = seed + XP1 + XP2, b = seed + XP2, c = seed, d = seed - XP1;
    while ((size_t)(end - p) >= 32) {
      a = xround(a, rd64(p));
      b = xround(b, rd64(p + 8));
      c = xround(c, rd64(p + 16));
      d = xround(d, rd64(p + 24));
      p += 32;
    }
    h = rol(a, 1) + rol(b, 7) + rol(c, 12) + rol(d, 18);
    h = xmerge(h, a);
    h = xmerge(h, b);
    h = xmerge(h, c);
    h = xmerge(h, d);
  } else {
    h = seed + XP5;
  }
  h += (uint64_t)len;
  while ((size_t)(end - p) >= 8) {
    h ^= xround(0, rd64(p));
    h = rol(h, 27) * XP1 + XP4;
    p += 8;
  }
  if ((size_t)(end - p) >= 4) {
    h ^= (uint64_t)rd32(p) * XP1;
    h = rol(h, 23) * XP2 + XP3;
    p += 4;
  }
  while (p < end) {
    h ^= (uint64_t)(*p) * XP5;
    h = rol(h, 11) * XP1;
    ++p;
  }
  h ^= h >> 33;
  h *= XP2;
  h ^= h >> 29;
  h *= XP3;
  h ^= h >> 32;
  return h;
}

/* hash(chunk i) = hash(chunk i content + hash(chunk i-1)) — 0602-…/README.md:99 */
int orc_hash_prompt(const uint8_t* model, size_t model_len, const uint8_t* prompt, size_t prompt_len,
                    uint32_t block_chars, uint64_t* out, uint32_t max_out) {
  if (block_chars == 0 || (!prompt && prompt_len) || (!model && model_len) || (!out && max_out)) return -1;
  uint8_t* buf = (uint8_t*)malloc((size_t)block_chars + 8);
  if (!buf) return -6;
  uint64_t prev = orc_xxh64(model, model_len, 0);
  uint32_t n = 0;
  for (size_t off = 0; off + block_chars <= prompt_len && n < max_out; off += block_chars) {
    memcpy(buf, prompt + off, block_chars);
    for (int i = 0; i < 8; ++i) buf[block_chars + i] = (uint8_t)(prev >> (8 * i));
    prev = orc_xxh64(buf, (size_t)block_chars + 8, 0);
    out[n++] = prev;
  }
  free(buf);
  return (int)n;
}

/* ------------------------------------------------------------------------------------------- */
/* Approximate prefix index: hash -> set(pod) (0602-…/README.md:101-112).  Naive structure:     */
/* open-addressed table of entries, each owning a sorted, growable pod list.                    */

typedef struct {
  uint64_t  hash;
  uint32_t* pods;
  uint32_t  n, cap;
  int       used;
  uint32_t  stamp; /* index epoch of the last insert of this hash (SEMANTICS.md 6a) */
} orc_entry;

struct orc_index {
  orc_entry* tab;
  uint64_t   cap;  /* power of two */
  uint64_t   used; /* entries allocated (including ones whose set became empty) */
  uint32_t   epoch; /* current index epoch, starts at 1 */
};

static uint64_t ix_home(uint64_t h, uint64_t cap) {
  h ^= h >> 31;
  h *= 0x7fb5d329728ea185ULL;
  h ^= h >> 27;
  return h & (cap - 1);
}

orc_index* orc_index_new(void) {
  orc_index* ix = (orc_index*)calloc(1, sizeof(*ix));
  if (!ix) return NULL;
  ix->cap = 1024;
  ix->epoch = 1;
  ix->tab = (orc_entry*)calloc(ix->cap, sizeof(orc_entry));
  if (!ix->tab) { free(ix); return NULL; }
  return ix;
}

void orc_index_clear(orc_index* ix) {
  if (!ix) return;
  for (uint64_t i = 0; i < ix->cap; ++i) free(ix->tab[i].pods);
  memset(ix->tab, 0, ix->cap * sizeof(orc_entry));
  ix->used = 0;
}

void orc_index_free(orc_index* ix) {
  if (!ix) return;
  orc_index_clear(ix);
  free(ix->tab);
  free(ix);
}

static orc_entry* ix_find(const orc_index* ix, uint64_t hash, int for_insert) {
  uint64_t i = ix_home(hash, ix->cap);
  for (;;) {
    orc_entry* e = &ix->tab[i];
    if (!e->used) return for_insert ? e : NULL;
    if (e->hash == hash) return e;
    i = (i + 1) & (ix->cap - 1);
  }
}

static void ix_grow(orc_index* ix) {
  orc_entry* old = ix->tab;
  uint64_t ocap = ix->cap;
  ix->cap = ocap * 2;
  ix->tab = (orc_entry*)calloc(ix->cap, sizeof(orc_entry));
  for (uint64_t i = 0; i < ocap; ++i)
    if (old[i].used) *ix_find(ix, old[i].hash, 1) = old[i];
  free(old);
}

void orc_index_insert(orc_index* ix, uint64_t hash, uint32_t pod) {
  if ((ix->used + 1) * 2 > ix->cap) ix_grow(ix);
  orc_entry* e = ix_find(ix, hash, 1);
  if (!e->used) {
    e->used = 1;
    e->hash = hash;
    e->pods = NULL;
    e->n = e->cap = 0;
    e->stamp = 0;
    ix->used++;
  }
  if (e->stamp < ix->epoch) e->stamp = ix->epoch; /* every insert stamps the hash, present pod or not */
  uint32_t lo = 0;
  while (lo < e->n && e->pods[lo] < pod) ++lo;
  if (lo < e->n && e->pods[lo] == pod) return; /* set semantics */
  if (e->n == e->cap) {
    e->cap = e->cap ? e->cap * 2 : 4;
    e->pods = (uint32_t*)realloc(e->pods, e->cap * sizeof(uint32_t));
  }
  memmove(e->pods + lo + 1, e->pods + lo, (e->n - lo) * sizeof(uint32_t));
  e->pods[lo] = pod;
  e->n++;
}

void orc_index_remove_pod(orc_index* ix, uint32_t pod) {
  for (uint64_t i = 0; i < ix->cap; ++i) {
    orc_entry* e = &ix->tab[i];
    if (!e->used) continue;
    for (uint32_t j = 0; j < e->n; ++j)
      if (e->pods[j] == pod) {
        memmove(e->pods + j, e->pods + j + 1, (e->n - j - 1) * sizeof(uint32_t));
        e->n--;
        break;
      }
  }
}

/* SEMANTICS.md 6b: publishing a snapshot forgets every slot that is a hole in it; an insert that names a hole is ignored. */
void orc_index_scrub_inactive(orc_index* ix, const eppk_pod_row* pods, uint32_t n_pods) {
  for (uint32_t p = 0; p < n_pods; ++p)
    if (pods[p].flags & EPPK_POD_INACTIVE) orc_index_remove_pod(ix, p);
}

/* Ageing (SEMANTICS.md 6a; docs/proposals/0602-…/README.md:82 "mimicking a similar cache eviction strategy"). */
uint32_t orc_index_evict_older(orc_index* ix, uint32_t min_epoch) {
  uint32_t gone = 0;
  for (uint64_t i = 0; i < ix->cap; ++i) {
    orc_entry* e = &ix->tab[i];
    if (e->used && e->n > 0 && e->stamp < min_epoch) { e->n = 0; ++gone; }
  }
  return gone;
}

/* SEMANTICS.md 6a, "window": a hash whose stamp would be 255 epochs old or more after the tick is evicted by the tick (the limit of
 * this build: the device keeps a stamp as an 8-bit tag in the hash's bucket header). */
#define ORC_EPOCH_WINDOW 254u
uint32_t orc_index_advance_epoch(orc_index* ix) {
  ++ix->epoch;
  if (ix->epoch > ORC_EPOCH_WINDOW) orc_index_evict_older(ix, ix->epoch - ORC_EPOCH_WINDOW);
  return ix->epoch;
}

/* SEMANTICS.md 6c.  age(h) = min(63, epoch - stamp(h)); cutage(p) = the smallest b >= 1 with #{h containing p : age(h) <= b} > cap;
 * p leaves every hash of age >= cutage(p).  (Naive: one pass per pod.) */
uint64_t orc_index_trim_pods(orc_index* ix, uint32_t n_pods_max, uint32_t cap) {
  uint64_t removed = 0;
  for (uint32_t p = 0; p < n_pods_max; ++p) {
    uint64_t hist[64];
    memset(hist, 0, sizeof hist);
    for (uint64_t i = 0; i < ix->cap; ++i) {
      const orc_entry* e = &ix->tab[i];
      if (!e->used || e->n == 0) continue;
      for (uint32_t j = 0; j < e->n; ++j)
        if (e->pods[j] == p) {
          uint32_t age = ix->epoch - e->stamp;
          hist[age < 63u ? age : 63u]++;
        }
    }
    uint64_t cum = 0;
    int cut = -1;
    for (int b = 0; b < 64; ++b) {
      cum += hist[b];
      if (cum > cap) { cut = b < 1 ? 1 : b; break; }
    }
    if (cut < 0) continue;
    for (uint64_t i = 0; i < ix->cap; ++i) {
      orc_entry* e = &ix->tab[i];
      if (!e->used || e->n == 0) continue;
      uint32_t age = ix->epoch - e->stamp;
      if (age > 63u) age = 63u;
      if ((int)age < cut) continue;
      for (uint32_t j = 0; j < e->n; ++j)
        if (e->pods[j] == p) {
          memmove(e->pods + j, e->pods + j + 1, (e->n - j - 1) * sizeof(uint32_t));
          e->n--;
          removed++;
          break;
        }
    }
  }
  return removed;
}

uint64_t orc_index_size(const orc_index* ix) {
  uint64_t n = 0;
  for (uint64_t i = 0; i < ix->cap; ++i) n += (ix->tab[i].used && ix->tab[i].n > 0);
  return n;
}

uint32_t orc_index_lookup(const orc_index* ix, uint64_t hash, uint32_t* pods, uint32_t cap) {
  const orc_entry* e = ix ? ix_find(ix, hash, 0) : NULL;
  if (!e) return 0;
  for (uint32_t j = 0; j < e->n && j < cap; ++j) pods[j] = e->pods[j];
  return e->n;
}

/* ------------------------------------------------------------------------------------------- */
/* Scorers (SEMANTICS.md §3).  Each fills score[c] for the request's candidates cand[0..nc).     */

static double clamp01(double s) {
  if (!(s >= 0.0)) return 0.0;
  if (s > 1.0) return 1.0;
  return s;
}

static void score_queue(const eppk_pod_row* pods, const uint32_t* cand, uint32_t nc, double* score) {
  uint32_t mn = pods[cand[0]].queue, mx = mn;
  for (uint32_t c = 1; c < nc; ++c) {
    uint32_t q = pods[cand[c]].queue;
    if (q < mn) mn = q;
    if (q > mx) mx = q;
  }
  for (uint32_t c = 0; c < nc; ++c) {
    if (mx == mn) score[c] = 1.0;
    else score[c] = (double)(mx - pods[cand[c]].queue) / (double)(mx - mn);
  }
}

static void score_kv(const eppk_pod_row* pods, const uint32_t* cand, uint32_t nc, double* score) {
  for (uint32_t c = 0; c < nc; ++c) score[c] = 1.0 - pods[cand[c]].kv_util;
}

static int bit128(const uint64_t w[2], int a) { return (int)((w[a >> 6] >> (a & 63)) & 1u); }
static uint32_t pop128(const uint64_t w[2]) {
  return (uint32_t)__builtin_popcountll(w[0]) + (uint32_t)__builtin_popcountll(w[1]);
}

static void score_lora(const eppk_pod_row* pods, const uint32_t* cand, uint32_t nc, int32_t adapter,
                       double* score) {
  for (uint32_t c = 0; c < nc; ++c) {
    const eppk_pod_row* r = &pods[cand[c]];
    int in_active = adapter >= 0 && bit128(r->active, adapter);
    int in_waiting = adapter >= 0 && bit128(r->waiting, adapter);
    uint32_t loaded = pop128(r->active) + pop128(r->waiting);
    if (in_active) score[c] = 1.0;
    else if (loaded < r->max_lora) score[c] = 0.8;
    else if (in_waiting) score[c] = 0.6;
    else score[c] = 0.0;
  }
}

/* matched[] is a scratch array over ALL pods (zeroed here). Returns index lookups performed. */
static uint32_t score_prefix(const orc_index* ix, uint32_t n_pods, const uint64_t* hashes, uint32_t n_blocks,
                             const uint32_t* cand, uint32_t nc, uint32_t* matched, double* score) {
  uint32_t probes = 0;
  memset(matched, 0, (size_t)n_pods * sizeof(uint32_t));
  for (uint32_t i = 0; i < n_blocks; ++i) {
    ++probes;
    const orc_entry* e = ix ? ix_find(ix, hashes[i], 0) : NULL;
    if (!e || e->n == 0) break; /* first hash with an empty pod set ends the walk */
    for (uint32_t j = 0; j < e->n; ++j)
      if (e->pods[j] < n_pods) matched[e->pods[j]]++;
  }
  for (uint32_t c = 0; c < nc; ++c)
    score[c] = n_blocks ? (double)matched[cand[c]] / (double)n_blocks : 0.0;
  return probes;
}

typedef struct {
  uint32_t* cand;
  uint32_t* matched;
  double*   score;
  double*   total;
} orc_scratch;

static int scratch_init(orc_scratch* s, uint32_t n_pods) {
  size_t n = n_pods ? n_pods : 1;
  s->cand = (uint32_t*)malloc(n * sizeof(uint32_t));
  s->matched = (uint32_t*)malloc(n * sizeof(uint32_t));
  s->score = (double*)malloc(n * sizeof(double));
  s->total = (double*)malloc(n * sizeof(double));
  return (s->cand && s->matched && s->score && s->total) ? 0 : -6;
}
static void scratch_free(orc_scratch* s) {
  free(s->cand); free(s->matched); free(s->score); free(s->total);
}

/* One scheduling cycle.  Returns 0, or -1 on an unknown scorer kind / bad row. */
static int schedule_one(const eppk_weighted_scorer* chain, uint32_t n_scorers, const eppk_pod_row* pods,
                        uint32_t n_pods, const orc_index* ix, const uint8_t* req, uint32_t max_blocks,
                        const uint64_t* mask_row, orc_scratch* s, int32_t* pick, double* pick_score,
                        uint32_t* probes_out, uint32_t* nc_out) {
  eppk_req_hdr hdr;
  memcpy(&hdr, req, sizeof hdr);
  const uint64_t* hashes = (const uint64_t*)(req + sizeof hdr);
  if (hdr.n_blocks > max_blocks) return -1;
  if (hdr.adapter < -1 || hdr.adapter >= (int32_t)EPPK_MAX_ADAPTERS) return -1;

  /* Filter: the candidate subset (request.go:104-133 expressed as a bitmask) */
  uint32_t nc = 0;
  for (uint32_t p = 0; p < n_pods; ++p) {
    if (pods[p].flags & EPPK_POD_INACTIVE) continue; /* a hole of the snapshot names no endpoint: never a candidate (SEMANTICS.md 6b) */
    if (!mask_row || ((mask_row[p >> 6] >> (p & 63)) & 1u)) s->cand[nc++] = p;
  }
  if (nc_out) *nc_out = nc;
  if (probes_out) *probes_out = 0;
  if (nc == 0) { /* fail closed */
    *pick = EPPK_NO_PICK;
    *pick_score = 0.0;
    return 0;
  }

  /* Score: weighted accumulation in chain order */
  for (uint32_t c = 0; c < nc; ++c) s->total[c] = 0.0;
  for (uint32_t k = 0; k < n_scorers; ++k) {
    switch (chain[k].kind) {
      case EPPK_SCORER_QUEUE: score_queue(pods, s->cand, nc, s->score); break;
      case EPPK_SCORER_KV: score_kv(pods, s->cand, nc, s->score); break;
      case EPPK_SCORER_LORA: score_lora(pods, s->cand, nc, hdr.adapter, s->score); break;
      case EPPK_SCORER_PREFIX: {
        uint32_t pr = score_prefix(ix, n_pods, hashes, hdr.n_blocks, s->cand, nc, s->matched, s->score);
        if (probes_out) *probes_out += pr;
        break;
      }
      default: return -1;
    }
    const double w = (double)chain[k].weight;
    for (uint32_t c = 0; c < nc; ++c) s->total[c] = s->total[c] + clamp01(s->score[c]) * w;
  }

  /* Picker: best score, first maximum in snapshot order */
  uint32_t best = 0;
  for (uint32_t c = 1; c < nc; ++c)
    if (s->total[c] > s->total[best]) best = c;
  *pick = (int32_t)s->cand[best];
  *pick_score = s->total[best];
  return 0;
}

static int pick_range(const eppk_weighted_scorer* chain, uint32_t n_scorers, const eppk_pod_row* pods,
                      uint32_t n_pods, const orc_index* ix, const uint8_t* reqs, uint32_t max_blocks,
                      uint32_t r0, uint32_t r1, const uint64_t* cand_mask, int32_t* out_pick,
                      double* out_score, uint32_t* out_probes) {
  orc_scratch s;
  if (scratch_init(&s, n_pods)) { scratch_free(&s); return -6; }
  const size_t stride = sizeof(eppk_req_hdr) + 8u * (size_t)max_blocks;
  const size_t mw = (n_pods + 63u) / 64u;
  int rc = 0;
  for (uint32_t r = r0; r < r1 && rc == 0; ++r) {
    int32_t pick;
    double sc;
    uint32_t pr;
    rc = schedule_one(chain, n_scorers, pods, n_pods, ix, reqs + stride * r, max_blocks,
                      cand_mask ? cand_mask + mw * r : NULL, &s, &pick, &sc, &pr, NULL);
    if (rc) break;
    out_pick[r] = pick;
    if (out_score) out_score[r] = sc;
    if (out_probes) out_probes[r] = pr;
  }
  scratch_free(&s);
  return rc;
}

int orc_pick_batch(const eppk_weighted_scorer* chain, uint32_t n_scorers, const eppk_pod_row* pods,
                   uint32_t n_pods, const orc_index* ix, const void* reqs, uint32_t max_blocks,
                   uint32_t n_reqs, const uint64_t* cand_mask, int32_t* out_pick, double* out_score,
                   uint32_t* out_probes) {
  if ((!chain && n_scorers) || (!pods && n_pods) || (!reqs && n_reqs) || (!out_pick && n_reqs)) return -1;
  if (n_scorers > EPPK_MAX_SCORERS) return -1;
  return pick_range(chain, n_scorers, pods, n_pods, ix, (const uint8_t*)reqs, max_blocks, 0, n_reqs,
                    cand_mask, out_pick, out_score, out_probes);
}

typedef struct {
  const eppk_weighted_scorer* chain; uint32_t n_scorers;
  const eppk_pod_row* pods; uint32_t n_pods; const orc_index* ix;
  const uint8_t* reqs; uint32_t max_blocks, r0, r1; const uint64_t* mask;
  int32_t* pick; double* score; int rc;
} orc_job;

static void* job_main(void* arg) {
  orc_job* j = (orc_job*)arg;
  j->rc = pick_range(j->chain, j->n_scorers, j->pods, j->n_pods, j->ix, j->reqs, j->max_blocks, j->r0,
                     j->r1, j->mask, j->pick, j->score, NULL);
  return NULL;
}

int orc_pick_batch_mt(const eppk_weighted_scorer* chain, uint32_t n_scorers, const eppk_pod_row* pods,
                      uint32_t n_pods, const orc_index* ix, const void* reqs, uint32_t max_blocks,
                      uint32_t n_reqs, const uint64_t* cand_mask, int32_t* out_pick, double* out_score,
                      int threads) {
  if (threads < 1) threads = 1;
  if (threads > 256) threads = 256;
  if ((uint32_t)threads > n_reqs) threads = n_reqs ? (int)n_reqs : 1;
  pthread_t tid[256];
  orc_job job[256];
  int rc = 0;
  for (int t = 0; t < threads; ++t) {
    orc_job* j = &job[t];
    j->chain = chain; j->n_scorers = n_scorers; j->pods = pods; j->n_pods = n_pods; j->ix = ix;
    j->reqs = (const uint8_t*)reqs; j->max_blocks = max_blocks; j->mask = cand_mask;
    j->pick = out_pick; j->score = out_score; j->rc = 0;
    j->r0 = (uint32_t)((uint64_t)n_reqs * (uint64_t)t / (uint64_t)threads);
    j->r1 = (uint32_t)((uint64_t)n_reqs * (uint64_t)(t + 1) / (uint64_t)threads);
    if (pthread_create(&tid[t], NULL, job_main, j)) { job_main(j); tid[t] = 0; }
  }
  for (int t = 0; t < threads; ++t) {
    if (tid[t]) pthread_join(tid[t], NULL);
    if (job[t].rc) rc = job[t].rc;
  }
  return rc;
}

/* ------------------------------------------------------------------------------------------- */
/* The same decisions by a second algorithm: snapshot tables + the pods a request's prefix walk names.
 * Nothing of the reference is restated here beyond schedule_one() above; this is what a careful CPU implementation of a whole batch
 * would do, and it is what bench.py times as `cpu_baseline` (the O(R x P) loop above is the shape of the reference's per-request
 * Schedule(), not a fair CPU competitor).  Per (chain, snapshot): the clamped per-pod scores of the request-independent scorers,
 * and per adapter class a the totals T_a[p] of a pod WITHOUT any prefix match (x + 0.0 * w == x, so the prefix terms drop out)
 * with the candidates ordered by (T_a descending, index ascending).  Per request: walk the hashes, count matches per named pod,
 * score the named candidates through the whole chain in chain order (same expression as schedule_one, -ffp-contract=off), take the
 * best of them, take the first pod of the class order that the walk did not name, and keep the higher total, the lower index on a
 * tie: the first maximum in snapshot order.  Unmasked batches only (a mask changes the QUEUE extremes per request).
 * tests/test_oracle_golden.py holds it bit-exact against orc_pick_batch. */

struct orc_tables {
  eppk_weighted_scorer chain[EPPK_MAX_SCORERS];
  uint32_t n_scorers, n_pods, nc, has_prefix, has_lora;
  uint32_t* cand;            /* active slots, ascending */
  uint8_t*  is_cand;         /* [n_pods] */
  double*   qs;              /* [n_pods] clamp01(queue score) among all candidates */
  double*   ks;              /* [n_pods] clamp01(1 - kv_util) */
  const eppk_pod_row* pods;  /* borrowed: must outlive the tables */
  double*   T[EPPK_MAX_ADAPTERS + 1];      /* class a + 1 -> [n_pods] totals without prefix matches */
  uint32_t* order[EPPK_MAX_ADAPTERS + 1];  /* class a + 1 -> [nc] candidates by (T desc, index asc) */
};

static double lora_score_one(const eppk_pod_row* r, int32_t adapter) {
  int in_active = adapter >= 0 && bit128(r->active, adapter);
  int in_waiting = adapter >= 0 && bit128(r->waiting, adapter);
  uint32_t loaded = pop128(r->active) + pop128(r->waiting);
  if (in_active) return 1.0;
  if (loaded < r->max_lora) return 0.8;
  if (in_waiting) return 0.6;
  return 0.0;
}

void orc_tables_free(orc_tables* tb) {
  if (!tb) return;
  for (uint32_t a = 0; a <= EPPK_MAX_ADAPTERS; ++a) { free(tb->T[a]); free(tb->order[a]); }
  free(tb->cand); free(tb->is_cand); free(tb->qs); free(tb->ks);
  free(tb);
}

orc_tables* orc_tables_new(const eppk_weighted_scorer* chain, uint32_t n_scorers, const eppk_pod_row* pods, uint32_t n_pods) {
  if (n_scorers > EPPK_MAX_SCORERS || (!chain && n_scorers) || (!pods && n_pods)) return NULL;
  orc_tables* tb = (orc_tables*)calloc(1, sizeof(*tb));
  if (!tb) return NULL;
  size_t n = n_pods ? n_pods : 1;
  tb->n_scorers = n_scorers; tb->n_pods = n_pods; tb->pods = pods;
  tb->cand = (uint32_t*)malloc(n * sizeof(uint32_t));
  tb->is_cand = (uint8_t*)calloc(n, 1);
  tb->qs = (double*)calloc(n, sizeof(double));
  tb->ks = (double*)calloc(n, sizeof(double));
  if (!tb->cand || !tb->is_cand || !tb->qs || !tb->ks) { orc_tables_free(tb); return NULL; }
  for (uint32_t k = 0; k < n_scorers; ++k) {
    tb->chain[k] = chain[k];
    if (chain[k].kind == EPPK_SCORER_PREFIX) tb->has_prefix = 1;
    else if (chain[k].kind == EPPK_SCORER_LORA) tb->has_lora = 1;
    else if (chain[k].kind != EPPK_SCORER_QUEUE && chain[k].kind != EPPK_SCORER_KV) { orc_tables_free(tb); return NULL; }
  }
  for (uint32_t p = 0; p < n_pods; ++p)
    if (!(pods[p].flags & EPPK_POD_INACTIVE)) { tb->cand[tb->nc++] = p; tb->is_cand[p] = 1; }
  if (tb->nc) {
    double* sc = (double*)malloc(tb->nc * sizeof(double));
    if (!sc) { orc_tables_free(tb); return NULL; }
    score_queue(pods, tb->cand, tb->nc, sc);
    for (uint32_t c = 0; c < tb->nc; ++c) tb->qs[tb->cand[c]] = clamp01(sc[c]);
    score_kv(pods, tb->cand, tb->nc, sc);
    for (uint32_t c = 0; c < tb->nc; ++c) tb->ks[tb->cand[c]] = clamp01(sc[c]);
    free(sc);
  }
  return tb;
}

static int cmp_order(const void* a, const void* b, void* ctx) {
  const double* T = (const double*)ctx;
  uint32_t x = *(const uint32_t*)a, y = *(const uint32_t*)b;
  if (T[x] > T[y]) return -1;
  if (T[x] < T[y]) return 1;
  return x < y ? -1 : (x > y ? 1 : 0);
}

static int tables_build_class(orc_tables* tb, int32_t adapter) {
  const uint32_t a = (uint32_t)(adapter + 1);
  if (tb->T[a]) return 0;
  size_t n = tb->n_pods ? tb->n_pods : 1;
  double* T = (double*)calloc(n, sizeof(double));
  uint32_t* ord = (uint32_t*)malloc((tb->nc ? tb->nc : 1) * sizeof(uint32_t));
  if (!T || !ord) { free(T); free(ord); return -6; }
  for (uint32_t c = 0; c < tb->nc; ++c) {
    const uint32_t p = tb->cand[c];
    double total = 0.0;
    for (uint32_t k = 0; k < tb->n_scorers; ++k) {
      double sc;
      switch (tb->chain[k].kind) {
        case EPPK_SCORER_QUEUE: sc = tb->qs[p]; break;
        case EPPK_SCORER_KV: sc = tb->ks[p]; break;
        case EPPK_SCORER_LORA: sc = clamp01(lora_score_one(&tb->pods[p], adapter)); break;
        default: sc = 0.0; break; /* PREFIX without a match */
      }
      total = total + sc * (double)tb->chain[k].weight;
    }
    T[p] = total;
    ord[c] = p;
  }
  qsort_r(ord, tb->nc, sizeof(uint32_t), cmp_order, T);
  tb->T[a] = T; tb->order[a] = ord;
  return 0;
}

typedef struct { uint32_t* stamp; uint32_t* cnt; uint32_t* named; uint32_t tick; } sparse_scratch;

static int sparse_range(const orc_tables* tb, const orc_index* ix, const uint8_t* reqs, uint32_t max_blocks, uint32_t r0,
                        uint32_t r1, int32_t* out_pick, double* out_score) {
  const size_t stride = sizeof(eppk_req_hdr) + 8u * (size_t)max_blocks;
  size_t n = tb->n_pods ? tb->n_pods : 1;
  sparse_scratch s;
  s.stamp = (uint32_t*)calloc(n, sizeof(uint32_t));
  s.cnt = (uint32_t*)malloc(n * sizeof(uint32_t));
  s.named = (uint32_t*)malloc(n * sizeof(uint32_t));
  s.tick = 0;
  int rc = (s.stamp && s.cnt && s.named) ? 0 : -6;
  for (uint32_t r = r0; r < r1 && rc == 0; ++r) {
    const uint8_t* req = reqs + stride * r;
    eppk_req_hdr hdr;
    memcpy(&hdr, req, sizeof hdr);
    const uint64_t* hashes = (const uint64_t*)(req + sizeof hdr);
    if (hdr.n_blocks > max_blocks || hdr.adapter < -1 || hdr.adapter >= (int32_t)EPPK_MAX_ADAPTERS) { rc = -1; break; }
    if (tb->nc == 0) { out_pick[r] = EPPK_NO_PICK; if (out_score) out_score[r] = 0.0; continue; }
    const int32_t cls = tb->has_lora ? hdr.adapter : -1;
    const double* T = tb->T[cls + 1];
    const uint32_t* ord = tb->order[cls + 1];
    uint32_t nn = 0;
    ++s.tick;
    if (tb->has_prefix && ix) {
      for (uint32_t i = 0; i < hdr.n_blocks; ++i) {
        const orc_entry* e = ix_find(ix, hashes[i], 0);
        if (!e || e->n == 0) break;
        for (uint32_t j = 0; j < e->n; ++j) {
          const uint32_t p = e->pods[j];
          if (p >= tb->n_pods) continue;
          if (s.stamp[p] != s.tick) { s.stamp[p] = s.tick; s.cnt[p] = 1; s.named[nn++] = p; }
          else s.cnt[p]++;
        }
      }
    }
    /* the named candidates, through the whole chain */
    int32_t best = -1;
    double best_t = 0.0;
    for (uint32_t j = 0; j < nn; ++j) {
      const uint32_t p = s.named[j];
      if (!tb->is_cand[p]) continue;
      double total = 0.0;
      for (uint32_t k = 0; k < tb->n_scorers; ++k) {
        double sc;
        switch (tb->chain[k].kind) {
          case EPPK_SCORER_QUEUE: sc = tb->qs[p]; break;
          case EPPK_SCORER_KV: sc = tb->ks[p]; break;
          case EPPK_SCORER_LORA: sc = clamp01(lora_score_one(&tb->pods[p], hdr.adapter)); break;
          default: sc = clamp01((double)s.cnt[p] / (double)hdr.n_blocks); break;
        }
        total = total + sc * (double)tb->chain[k].weight;
      }
      if (best < 0 || total > best_t || (total == best_t && (int32_t)p < best)) { best = (int32_t)p; best_t = total; }
    }
    /* the best candidate the walk did not name */
    for (uint32_t c = 0; c < tb->nc; ++c) {
      const uint32_t p = ord[c];
      if (s.stamp[p] == s.tick) continue;
      if (best < 0 || T[p] > best_t || (T[p] == best_t && (int32_t)p < best)) { best = (int32_t)p; best_t = T[p]; }
      break;
    }
    out_pick[r] = best;
    if (out_score) out_score[r] = best_t;
  }
  free(s.stamp); free(s.cnt); free(s.named);
  return rc;
}

typedef struct { const orc_tables* tb; const orc_index* ix; const uint8_t* reqs; uint32_t max_blocks, r0, r1; int32_t* pick; double* score; int rc; } sparse_job;
static void* sparse_main(void* arg) {
  sparse_job* j = (sparse_job*)arg;
  j->rc = sparse_range(j->tb, j->ix, j->reqs, j->max_blocks, j->r0, j->r1, j->pick, j->score);
  return NULL;
}

int orc_pick_batch_sparse(orc_tables* tb, const orc_index* ix, const void* reqs, uint32_t max_blocks, uint32_t n_reqs,
                          int32_t* out_pick, double* out_score, int threads) {
  if (!tb || (!reqs && n_reqs) || (!out_pick && n_reqs)) return -1;
  const size_t stride = sizeof(eppk_req_hdr) + 8u * (size_t)max_blocks;
  /* classes this batch needs (kept in the tables for the next batch of the snapshot) */
  if (tb->nc) {
    if (!tb->has_lora) { if (tables_build_class(tb, -1)) return -6; }
    else for (uint32_t r = 0; r < n_reqs; ++r) {
      eppk_req_hdr hdr;
      memcpy(&hdr, (const uint8_t*)reqs + stride * r, sizeof hdr);
      if (hdr.adapter < -1 || hdr.adapter >= (int32_t)EPPK_MAX_ADAPTERS) return -1;
      if (!tb->T[hdr.adapter + 1] && tables_build_class(tb, hdr.adapter)) return -6;
    }
  }
  if (threads < 1) threads = 1;
  if (threads > 256) threads = 256;
  if ((uint32_t)threads > n_reqs) threads = n_reqs ? (int)n_reqs : 1;
  pthread_t tid[256];
  sparse_job job[256];
  int rc = 0;
  for (int t = 0; t < threads; ++t) {
    sparse_job* j = &job[t];
    j->tb = tb; j->ix = ix; j->reqs = (const uint8_t*)reqs; j->max_blocks = max_blocks; j->pick = out_pick; j->score = out_score; j->rc = 0;
    j->r0 = (uint32_t)((uint64_t)n_reqs * (uint64_t)t / (uint64_t)threads);
    j->r1 = (uint32_t)((uint64_t)n_reqs * (uint64_t)(t + 1) / (uint64_t)threads);
    if (threads == 1 || pthread_create(&tid[t], NULL, sparse_main, j)) { sparse_main(j); tid[t] = 0; }
  }
  for (int t = 0; t < threads; ++t) {
    if (tid[t]) pthread_join(tid[t], NULL);
    if (job[t].rc) rc = job[t].rc;
  }
  return rc;
}

int orc_pick_batch_assumed(const eppk_weighted_scorer* chain, uint32_t n_scorers, eppk_pod_row* pods, uint32_t n_pods,
                           const orc_index* ix, const void* reqs, uint32_t max_blocks, uint32_t n_reqs,
                           const uint64_t* cand_mask, uint32_t epochs, int32_t* out_pick, double* out_score) {
  if (epochs == 0)
    return orc_pick_batch(chain, n_scorers, pods, n_pods, ix, reqs, max_blocks, n_reqs, cand_mask, out_pick, out_score, NULL);
  if ((!chain && n_scorers) || (!pods && n_pods) || (!reqs && n_reqs) || (!out_pick && n_reqs)) return -1;
  const uint32_t per = (n_reqs + epochs - 1u) / epochs;
  for (uint32_t lo = 0; lo < n_reqs; lo += per) {
    const uint32_t hi = lo + per < n_reqs ? lo + per : n_reqs;
    /* every request of the epoch against the same gauges ... */
    int rc = pick_range(chain, n_scorers, pods, n_pods, ix, (const uint8_t*)reqs, max_blocks, lo, hi, cand_mask, out_pick,
                        out_score, NULL);
    if (rc) return rc;
    /* ... then the assumed load of what was just routed (006-scheduler/README.md:154-156) */
    for (uint32_t r = lo; r < hi; ++r)
      if (out_pick[r] >= 0) pods[out_pick[r]].queue += 1u;
  }
  return 0;
}

static uint64_t splitmix64_fin(uint64_t z) {
  z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ULL;
  z ^= z >> 27; z *= 0x94D049BB133111EBULL;
  z ^= z >> 31;
  return z;
}

int orc_pick_random_topk(const eppk_weighted_scorer* chain, uint32_t n_scorers, const eppk_pod_row* pods, uint32_t n_pods,
                         const orc_index* ix, const void* reqs, uint32_t max_blocks, uint32_t n_reqs,
                         const uint64_t* cand_mask, uint32_t k, uint64_t seed, int32_t* out_pick, double* out_score) {
  if (k < 1 || k > EPPK_MAX_TOPK) return -1;
  orc_scratch s;
  if (scratch_init(&s, n_pods)) { scratch_free(&s); return -6; }
  const size_t stride = sizeof(eppk_req_hdr) + 8u * (size_t)max_blocks;
  const size_t mw = (n_pods + 63u) / 64u;
  int rc = 0;
  for (uint32_t r = 0; r < n_reqs && rc == 0; ++r) {
    int32_t pick; double sc; uint32_t nc = 0;
    rc = schedule_one(chain, n_scorers, pods, n_pods, ix, (const uint8_t*)reqs + stride * r, max_blocks,
                      cand_mask ? cand_mask + mw * r : NULL, &s, &pick, &sc, NULL, &nc);
    if (rc) break;
    /* the k best under (total desc, index asc): k selection passes over the candidates (naive on purpose) */
    uint32_t top[EPPK_MAX_TOPK]; uint32_t n = 0;
    for (uint32_t i = 0; i < k && i < nc; ++i) {
      int best = -1;
      for (uint32_t c = 0; c < nc; ++c) {
        int taken = 0;
        for (uint32_t t = 0; t < n; ++t) taken |= top[t] == c;
        if (taken) continue;
        if (best < 0 || s.total[c] > s.total[best]) best = (int)c;   /* candidates ascend by index: strict > keeps the lowest */
      }
      top[n++] = (uint32_t)best;
    }
    if (n == 0) { out_pick[r] = EPPK_NO_PICK; if (out_score) out_score[r] = 0.0; continue; }
    const uint64_t u = splitmix64_fin(seed + ((uint64_t)r + 1u) * 0x9E3779B97F4A7C15ULL);
    const uint32_t j = (uint32_t)(u % (uint64_t)n);
    out_pick[r] = (int32_t)s.cand[top[j]];
    if (out_score) out_score[r] = s.total[top[j]];
  }
  scratch_free(&s);
  return rc;
}

/* Ordered fallbacks (SEMANTICS.md 2a; PickResult.Fallbacks, pkg/lwepp/handlers/server.go:72-77; "multiple comma-separated fallbacks",
 * docs/proposals/004-endpoint-picker-protocol/README.md:73): the k best candidates of every request under (total descending, index
 * ascending), padded with EPPK_NO_PICK / 0.0 -- the same k selection passes as orc_pick_random_topk, for whole batches and on several
 * threads, so that the GPU's fallback lists can be checked at full batch sizes (binding.py keeps the independent formulation -- a
 * lexsort over orc_score_row's totals -- and tests/test_oracle_golden.py holds the two equal). */
static int topk_range(const eppk_weighted_scorer* chain, uint32_t n_scorers, const eppk_pod_row* pods, uint32_t n_pods,
                      const orc_index* ix, const uint8_t* reqs, uint32_t max_blocks, uint32_t r0, uint32_t r1,
                      const uint64_t* cand_mask, uint32_t k, int32_t* out_pick, double* out_score) {
  orc_scratch s;
  if (scratch_init(&s, n_pods)) { scratch_free(&s); return -6; }
  const size_t stride = sizeof(eppk_req_hdr) + 8u * (size_t)max_blocks;
  const size_t mw = (n_pods + 63u) / 64u;
  int rc = 0;
  for (uint32_t r = r0; r < r1 && rc == 0; ++r) {
    int32_t pick; double sc; uint32_t nc = 0;
    rc = schedule_one(chain, n_scorers, pods, n_pods, ix, reqs + stride * r, max_blocks, cand_mask ? cand_mask + mw * r : NULL, &s,
                      &pick, &sc, NULL, &nc);
    if (rc) break;
    uint32_t top[EPPK_MAX_TOPK]; uint32_t n = 0;
    for (uint32_t i = 0; i < k && i < nc; ++i) {
      int best = -1;
      for (uint32_t c = 0; c < nc; ++c) {
        int taken = 0;
        for (uint32_t t = 0; t < n; ++t) taken |= top[t] == c;
        if (taken) continue;
        if (best < 0 || s.total[c] > s.total[best]) best = (int)c;   /* candidates ascend by index: strict > keeps the lowest */
      }
      top[n++] = (uint32_t)best;
    }
    for (uint32_t i = 0; i < k; ++i) {
      out_pick[(size_t)r * k + i] = i < n ? (int32_t)s.cand[top[i]] : EPPK_NO_PICK;
      if (out_score) out_score[(size_t)r * k + i] = i < n ? s.total[top[i]] : 0.0;
    }
  }
  scratch_free(&s);
  return rc;
}

typedef struct {
  const eppk_weighted_scorer* chain; uint32_t n_scorers; const eppk_pod_row* pods; uint32_t n_pods; const orc_index* ix;
  const uint8_t* reqs; uint32_t max_blocks, r0, r1, k; const uint64_t* mask; int32_t* pick; double* score; int rc;
} topk_job;

static void* topk_main(void* arg) {
  topk_job* j = (topk_job*)arg;
  j->rc = topk_range(j->chain, j->n_scorers, j->pods, j->n_pods, j->ix, j->reqs, j->max_blocks, j->r0, j->r1, j->mask, j->k, j->pick, j->score);
  return NULL;
}

int orc_pick_topk(const eppk_weighted_scorer* chain, uint32_t n_scorers, const eppk_pod_row* pods, uint32_t n_pods,
                  const orc_index* ix, const void* reqs, uint32_t max_blocks, uint32_t n_reqs, const uint64_t* cand_mask,
                  uint32_t k, int threads, int32_t* out_pick, double* out_score) {
  if (k < 1 || k > EPPK_MAX_TOPK || n_scorers > EPPK_MAX_SCORERS) return -1;
  if ((!chain && n_scorers) || (!pods && n_pods) || (!reqs && n_reqs) || (!out_pick && n_reqs)) return -1;
  if (threads < 1) threads = 1;
  if (threads > 256) threads = 256;
  if ((uint32_t)threads > n_reqs) threads = n_reqs ? (int)n_reqs : 1;
  pthread_t tid[256];
  topk_job job[256];
  int rc = 0;
  for (int t = 0; t < threads; ++t) {
    topk_job* j = &job[t];
    j->chain = chain; j->n_scorers = n_scorers; j->pods = pods; j->n_pods = n_pods; j->ix = ix; j->reqs = (const uint8_t*)reqs;
    j->max_blocks = max_blocks; j->k = k; j->mask = cand_mask; j->pick = out_pick; j->score = out_score; j->rc = 0;
    j->r0 = (uint32_t)((uint64_t)n_reqs * (uint64_t)t / (uint64_t)threads);
    j->r1 = (uint32_t)((uint64_t)n_reqs * (uint64_t)(t + 1) / (uint64_t)threads);
    if (threads == 1 || pthread_create(&tid[t], NULL, topk_main, j)) { topk_main(j); tid[t] = 0; }
  }
  for (int t = 0; t < threads; ++t) {
    if (tid[t]) pthread_join(tid[t], NULL);
    if (job[t].rc) rc = job[t].rc;
  }
  return rc;
}

int orc_score_row(const eppk_weighted_scorer* chain, uint32_t n_scorers, const eppk_pod_row* pods,
                  uint32_t n_pods, const orc_index* ix, const void* req, const uint64_t* mask_row,
                  double* out_total) {
  orc_scratch s;
  if (scratch_init(&s, n_pods)) { scratch_free(&s); return -6; }
  int32_t pick; double sc; uint32_t nc = 0;
  /* max_blocks is only a bound check here: accept the row's own n_blocks */
  eppk_req_hdr hdr; memcpy(&hdr, req, sizeof hdr);
  int rc = schedule_one(chain, n_scorers, pods, n_pods, ix, (const uint8_t*)req, hdr.n_blocks, mask_row,
                        &s, &pick, &sc, NULL, &nc);
  if (rc == 0) {
    for (uint32_t p = 0; p < n_pods; ++p) out_total[p] = NAN;
    for (uint32_t c = 0; c < nc; ++c) out_total[s.cand[c]] = s.total[c];
  }
  scratch_free(&s);
  return rc;
}

void orc_index_insert_picks(orc_index* ix, const void* reqs, uint32_t max_blocks, uint32_t n_reqs,
                            const int32_t* picks) {
  const size_t stride = sizeof(eppk_req_hdr) + 8u * (size_t)max_blocks;
  for (uint32_t r = 0; r < n_reqs; ++r) {
    if (picks[r] < 0) continue;
    const uint8_t* row = (const uint8_t*)reqs + stride * r;
    eppk_req_hdr hdr; memcpy(&hdr, row, sizeof hdr);
    const uint64_t* h = (const uint64_t*)(row + sizeof hdr);
    for (uint32_t i = 0; i < hdr.n_blocks && i < max_blocks; ++i) orc_index_insert(ix, h[i], (uint32_t)picks[r]);
  }
}

/* ------------------------------------------------------------------------------------------- */
/* Pinned behaviour adjacent to the pick.                                                        */

/* RoundRobinPicker.Pick — pkg/lwepp/handlers/server.go:90-101 */
int32_t orc_round_robin(uint64_t* counter, uint32_t n_candidates) {
  if (n_candidates == 0) return EPPK_NO_PICK;              /* :91-93  codes.Unavailable */
  uint64_t index = __atomic_add_fetch(counter, 1, __ATOMIC_SEQ_CST); /* :95 atomic.AddUint64 returns the new value */
  return (int32_t)(index % (uint64_t)n_candidates);        /* :96 */
}

/* strings.TrimSpace trims Unicode White_Space; on UTF-8: bytes of the white-space rune at a[0..n), or 0 */
static size_t space_len(const char* a, size_t n) {
  const unsigned char* u = (const unsigned char*)a;
  if (n >= 1 && (u[0] == ' ' || (u[0] >= 9 && u[0] <= 13))) return 1;
  if (n >= 2 && u[0] == 0xC2 && (u[1] == 0x85 || u[1] == 0xA0)) return 2;                       /* NEL, NBSP */
  if (n >= 3 && u[0] == 0xE1 && u[1] == 0x9A && u[2] == 0x80) return 3;                        /* U+1680 */
  if (n >= 3 && u[0] == 0xE2 && u[1] == 0x80 && ((u[2] >= 0x80 && u[2] <= 0x8A) || u[2] == 0xA8 || u[2] == 0xA9 || u[2] == 0xAF)) return 3;
  if (n >= 3 && u[0] == 0xE2 && u[1] == 0x81 && u[2] == 0x9F) return 3;                        /* U+205F */
  if (n >= 3 && u[0] == 0xE3 && u[1] == 0x80 && u[2] == 0x80) return 3;                        /* U+3000 */
  return 0;
}

/* Behaviour of Go's net.SplitHostPort as used at request.go:110: returns 1 and [host,port) spans on
 * success, 0 on any error (missing port, too many colons, bad brackets). */
static int split_host_port(const char* s, size_t n, size_t* h0, size_t* h1, size_t* p0) {
  size_t j = 0, k = 0;
  long i = -1;
  for (size_t t = 0; t < n; ++t) if (s[t] == ':') i = (long)t;
  if (i < 0) return 0; /* missing port */
  if (s[0] == '[') {
    long end = -1;
    for (size_t t = 0; t < n; ++t) if (s[t] == ']') { end = (long)t; break; }
    if (end < 0) return 0;
    if ((size_t)(end + 1) == n) return 0;
    if (end + 1 != i) return 0;
    *h0 = 1; *h1 = (size_t)end;
    j = 1; k = (size_t)end + 1;
  } else {
    *h0 = 0; *h1 = (size_t)i;
    for (size_t t = 0; t < (size_t)i; ++t) if (s[t] == ':') return 0; /* too many colons */
  }
  for (size_t t = j; t < n; ++t) if (s[t] == '[') return 0;
  for (size_t t = k; t < n; ++t) if (s[t] == ']') return 0;
  *p0 = (size_t)i + 1;
  return 1;
}

static int span_eq(const char* a, size_t n, const char* z) { return strlen(z) == n && memcmp(a, z, n) == 0; }

/* Subset filter of handleRequestHeaders — pkg/lwepp/handlers/request.go:104-133.
 * filter == NULL: no subset filter, every pod is a candidate (:136-137).
 * Empty-after-trim entries are dropped (the metadata path, :58-61). */
int orc_subset_mask(const char* const* addrs, const char* const* ports, uint32_t n_pods, const char* filter,
                    uint64_t* out_mask) {
  if ((!addrs || !ports || !out_mask) && n_pods) return -1;
  const uint32_t mw = (n_pods + 63u) / 64u;
  memset(out_mask, 0, (size_t)mw * 8u);
  int count = 0;
  if (!filter) {
    for (uint32_t p = 0; p < n_pods; ++p) out_mask[p >> 6] |= 1ULL << (p & 63);
    return (int)n_pods;
  }
  for (uint32_t p = 0; p < n_pods; ++p) {
    int allowed = 0;
    const char* s = filter;
    while (!allowed) {
      const char* e = strchr(s, ',');
      size_t n = e ? (size_t)(e - s) : strlen(s);
      const char* a = s;
      for (size_t k; n && (k = space_len(a, n)) != 0;) { a += k; n -= k; }
      for (;;) {
        size_t k = 0;
        for (size_t back = 1; back <= 3 && back <= n; ++back)
          if (space_len(a + n - back, back) == back) { k = back; break; }
        if (!k) break;
        n -= k;
      }
      if (n) {
        size_t h0, h1, p0;
        if (split_host_port(a, n, &h0, &h1, &p0)) {
          /* ip:port entry allows exactly that port (:110-114, :124-127) */
          if (span_eq(a + h0, h1 - h0, addrs[p]) && span_eq(a + p0, n - p0, ports[p])) allowed = 1;
        } else if (span_eq(a, n, addrs[p])) {
          allowed = 1; /* ip-only entry allows all ports (:115-117, :122-123) */
        }
      }
      if (!e) break;
      s = e + 1;
    }
    if (allowed) { out_mask[p >> 6] |= 1ULL << (p & 63); ++count; }
  }
  return count;
}
