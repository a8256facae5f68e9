// eppk_kernels.hip.h — gfx950 kernels of the batched endpoint pick.  (Device code only.)
//
// What is computed: SEMANTICS.md.  How it is laid out for CDNA4:
//
//   * one WAVEFRONT (64 lanes) owns one request at a time; lane l owns pods {j*64 + l}.
//   * every per-pod bit set that a request needs (prefix-index pod sets, LoRA active/waiting sets,
//     free-slot set, candidate mask) is stored / re-laid "lane-transposed": a row of 64 lane words
//     (type LW = u16/u32/u64 for P <= 1024/2048/4096) where bit j of lane word l is pod j*64+l.
//     One coalesced 64*sizeof(LW)-byte load therefore hands every lane exactly the membership bits
//     of the pods it scores, with no cross-lane traffic.
//   * the prefix walk adds up to n_blocks such rows into bit-sliced (vertical) counters: NPL planes
//     of LW per lane, i.e. matched[pod] for 64 pods per lane in NPL registers.
//   * pod-only scorers that lead the chain are fused on the host into base[p] (same binary64 ops in
//     the same order, so bit-exact) and staged once per workgroup into LDS; the pair loop is then
//     two binary64 adds, two LDS table look-ups, and a strict-greater running argmax.
//   * beside its dense row every index slot keeps the same pod set as a SHORT LIST of 16-bit pod ids (<= 24 members, 64 B):
//     a request whose hits all have lists is scored from the lists -- one 16-byte load per lane for 16 hits -- and when the
//     lists are identical (the blocks of a shared prefix are cached together) without any counting at all.
//   * argmax across lanes: DPP max reduction on the score, ties to the lowest pod index.
//   * no MFMA: this is integer/bit/f64-add work (north_star); the bound is instruction issue once the index bytes are
//     lists (an index beyond the caches is bound by HBM gathers of 64-byte lines: DESIGN.md §3.1, §7).
//   * candidate masks (MASKED) and ordered fallbacks (TOPK) take the list routes too: the mask row is used as it is
//     (natural layout, one word per lane), fallbacks are a k-way merge of the listed pods with the adapter's top table.
//   * holes of a snapshot (eppk_pod_row.flags & EPPK_POD_INACTIVE) are ANDed out of `valid` once per kernel.
//   * "EPPK_DBG_NO_UNIFORM" (correct results, slower) sends every list request through the general histogram route: the
//     GPU suite is run once with such a build whenever that route changes.
//
// Replaces (reference, all spec-only or Go): Scorer.Score + weighted sum + Picker.Pick
//   docs/proposals/0845-scheduler-architecture-proposal/interfaces/interface.go:113-142,
//   called per request through EndpointPicker.Pick, pkg/lwepp/handlers/server.go:79-82.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "../../include/eppk.h"   // eppk_pod_row (the snapshot producer reads raw rows)

// Build-time knobs
#ifndef EPPK_FAST_MAX_THREADS
#define EPPK_FAST_MAX_THREADS 1024 // largest workgroup the fast kernel may be launched with (caps its VGPR budget at 512 * 256 / threads)
#endif
#ifndef EPPK_SW_PIPE
#define EPPK_SW_PIPE 1    // 1: key gather of request r+1 issued during request r (16 landing VGPRs live across the evaluation);
#endif                    // 0: keys gathered and consumed at the top of the request (fewer VGPRs -> more waves per SIMD)
#ifndef EPPK_LATE_HOOK
#define EPPK_LATE_HOOK 0
#endif
#ifndef EPPK_ROW_PREFETCH
#define EPPK_ROW_PREFETCH 10 // the request row of iteration r + this many is pulled into L2 by a throw-away load (0 = off): the rows
#endif                      // stream from HBM (bench.py rotates > 256 MB of batches) and the pipeline's own row load, two iterations
                            // ahead, has less slack than an HBM round trip
#ifndef EPPK_MIN_WAVES
#define EPPK_MIN_WAVES 1 // __launch_bounds__ minimum waves per SIMD for the fast kernel
#endif
// EPPK_DBG_* macros below are timing experiments (ablation builds made by scripts/ab.sh); they produce WRONG results and are
// never defined in the product build.

namespace eppk {

constexpr uint32_t kNotFound = 0xFFFFFFFFu;
constexpr uint32_t kNoPod = 0xFFFFFFFFu;
constexpr uint32_t kHomeMul = 0x9E3779B1u;    // 2^32 / golden ratio (odd): multiplicative hashing of the folded key
constexpr uint64_t kTomb = ~0ull;           // key of a slot whose pod set became empty (never matches, never reused)
constexpr uint32_t kBucket = 8u;            // u64 words per bucket = half a 128-byte line (protocol v5, round 6):
                                            //   dword 0      flags (bit 0: "a key that hashed here lives in a later bucket")
                                            //   dwords 1..5  META of the keys in words 3..7: {tag : 8 (top byte), set id : 24}
                                            //   words 3..7   keys, filled front to back (0 = empty, ~0 = tombstone)
constexpr uint32_t kKeySub0 = 3u;           // first key word of a bucket; slot = bucket * 8 + word (3..7), as before the index of the key word
constexpr uint32_t kKeysPerBucket = kBucket - kKeySub0;
// The META dword of a key: its stamp TAG (0 = no valid pod set behind this word: empty, tombstone, or a claim in progress; else
// 1 + (stamp - 1) mod 255) and the SET ID of its pod set:
//   sid <  kSidSets   the set is exactly ONE pod, sid = its id: no list line is needed to know the set
//   sid >= kSidSets   line sid - kSidSets of the SET TABLE holds an immutable, interned copy of the slot's sorted pod list (2..24 ids): two
//                     keys with the same sid have the same pod set, exactly (the reverse need not hold: an interning race or a full
//                     table may give equal sets different ids or none)
//   kSidNone          no id: the set lives in its dense row (> 24 pods), or it has changed since it was last canonicalised (only between
//                     an update launch and the index_canon_kernel behind it), or the set table was full
// The slot's own list line stays the source of truth for every reader but pick_quad_kernel, which decides "all hits of this request list
// the same pods" from the bucket lines it has gathered anyway and fetches ONE set line per request (one line per hit before: 16 of the 36
// L2 requests per decision of the headline, half of the HBM requests of a cold index, and what made a returning request the slow path).
constexpr uint32_t kSidMask = 0x00FFFFFFu, kSidNone = 0x00FFFFFFu, kSidSets = 0x00010000u;
__host__ __device__ __forceinline__ constexpr uint32_t words_cap(uint32_t slots) { return slots / 32u * 15u; }   // 3/4 of the key words (5 of every 8 words) may be non-empty
__host__ __device__ __forceinline__ constexpr bool is_key_word(uint32_t row) { return (row & (kBucket - 1u)) >= kKeySub0; }
__host__ __device__ __forceinline__ constexpr uint32_t meta_dword(uint32_t slot) { return (slot & ~(kBucket - 1u)) * 2u + (slot & (kBucket - 1u)) - 2u; }   // dword index into keys[]
constexpr uint32_t kStatSlots = 32768u;  // per-wavefront probe-statistics slots: stats[4 + 2*wave + {0,1}]

// ---- kernel argument blocks (plain structs, passed by value) --------------------------------

struct KSnap {
  const double*   base;    // [J*64] fused leading pod-only terms (fast path)
  const double*   post[2]; // [J*64] products of the pod-only scorers that FOLLOW a LORA / PREFIX scorer (GEN fast path)
  const uint32_t* queue;   // [J*64]
  const double*   kv;      // [J*64]
  const void*     thi_t;   // [129][64] LW LoRA tier planes per adapter row (row 128 = base model), bit j of [a][l] = pod j*64+l:
  const void*     tlo_t;   //   tier = 2*hi + lo -> {0: 0.0, 1: 0.6 waiting, 2: 0.8 free slot, 3: 1.0 active};
                           //   hi = active | free, lo = active | (~free & waiting), precomputed at publish
  const double*   topv;    // [129][64] per adapter row (128 = base model): the 64 best pods by
  const uint32_t* topi;    //           T_a[p] = base[p] (+ lw[tier(a,p)]), sorted (T desc, p asc); kNoPod-padded
  const void*     qmin_t;  // [64] LW  pods whose queue == qmin / qmax (masked fast path: are the request's
  const void*     qmax_t;  //          QUEUE normalisers the global ones?)
  const uint64_t* nat;     // [3][64] u64 NATURAL layout (word w = pods 64w .. 64w+63): active-and-existing pods, pods at the minimum
                           //          queue depth, pods at the maximum -- what the masked list routes AND with a request's mask row
  const void*     act_t;   // [64] LW  ACTIVE slots: bit j of lane word l clear = pod j*64+l is a hole of the snapshot
                           //          (eppk_pod_row.flags & EPPK_POD_INACTIVE): never a candidate, outside the QUEUE normalisers
                           //          and the top tables; the index never lists it (scrubbed at publish, inserts dropped)
  uint32_t        lead_queue;  // the fused leading run contains a QUEUE scorer
  const double*   pterm;   // [(B+1)][pterm_ld] exact clamp01(c/n) * w_prefix for 1 <= n <= B, c <= n (null when B > 64)
  uint32_t pterm_ld;
  const uint8_t*  blob;    // all of the above live in ONE allocation: the fast kernel reads them through a single buffer
  uint32_t blob_bytes;     // descriptor (SGPR base + 32-bit offsets: no 64-bit per-lane pointers in VGPRs).  topv, topi, thi_t
                           // and tlo_t sit at the compile-time offsets of SnapOff<LW> (immediates, not SGPRs)
  uint32_t n_pods;
  uint32_t J;              // ceil(n_pods/64)
  const uint32_t* qrange;  // [2] minimum / maximum queue depth over the ACTIVE pods (unmasked QUEUE scorer); device memory: the
                           // snapshot producer computes it (snap_qrange_kernel), also when assumed load moved the gauges
  uint32_t* status;        // sticky launch-status flags of the context (eppk_launch_status): bit 0 = a request row handed to a
                           // *_device entry point was out of range (n_blocks > max_blocks or adapter outside [-1, 128)); such a
                           // request is NOT scored: its pick is EPPK_NO_PICK (SEMANTICS.md §7: never silently truncated).
                           // The fast kernel does not keep this pointer in SGPRs: it stores the flag into the LAST dword of the
                           // snapshot blob (kBlobStatusTail bytes from its end), reachable through the blob's descriptor.
};
constexpr uint32_t kBlobStatusTail = 4u;
constexpr uint32_t kStatusBadRow = 1u;      // eppk.h: EPPK_LAUNCH_BAD_REQUEST_ROW
constexpr uint32_t kStatusBadPick = 2u;     // eppk.h: EPPK_LAUNCH_BAD_PICK
constexpr uint32_t kStatusIndexStall = 4u;  // eppk.h: EPPK_LAUNCH_INDEX_STALL

// Prefix index: a BUCKETED open-addressing table.  A key lives in the first free key word of its home bucket (8 u64 words =
// 64 bytes: flags + five meta dwords, then five keys in words 3..7, filled front to back, 0 = empty, ~0 = tombstone; kBucket above).
// A look-up therefore reads ONE 64-byte bucket -- the key AND the identity of its pod set -- and is finished unless the key is
// absent from an overflowed bucket (the library allocates one bucket per four API slots: 0.06 % of the buckets at the recommended
// load of a quarter of the API slots, 1.7 % at the hard limit of a half) -- no data-dependent probe chain on the hot path.  (With per-slot linear probing the 32 parallel look-ups of a request
// needed max-over-lanes dependent round trips: 3-6 at load 0.5; measured 31 of 95 us per batch.)
// Slot = bucket * 8 + word; the pod-set row of a key has its slot's index (rows of the three meta words are unused).
struct KIndex {
  const uint64_t* keys;    // [slots+2]; keys[slots], keys[slots+1] = presence of the reserved hashes 0 / ~0.
                           // Invariant: a key that is present has a NON-EMPTY row.
  const void*     bitmaps; // [slots+3][64] LW: rows slots / slots+1 hold hashes 0 / ~0, row slots+2 is all-zero
  uint32_t slots;          // PHYSICAL slots (words of the table): power of two >= 128 (0 = no index); slots / 8 buckets.  Twice the API's
                           // index_slots: five of a bucket's eight words hold keys
  uint32_t shift;          // 32 - log2(slots / 8)
  uint32_t small;          // rows + keys (ONE allocation, rows first) are < 4 GiB: both are read through one buffer descriptor
  uint32_t table_bytes;    // bytes of that allocation when small
  uint32_t keys_off;       // byte offset of keys inside it
  uint32_t sets_mask;      // the SET TABLE (kSidSets above): sets_mask + 1 lines of 64 bytes (a power of two) right behind the slots' lists
                           // in the same allocation -- line slots + 4 + i of `lists_all` -- so that pick_quad_kernel reads both through one descriptor
  const uint32_t* lists;   // [slots+4 (+ set table)][16] short pod lists (kListCap ids of 16 bits + count) for the pick kernels' LIST ROUTES: nullptr when
                           // those are off (EPPK_LISTS=0) or the table is beyond a 32-bit buffer descriptor (4 GiB: slots >= 2^26)
  const uint32_t* lists_all;   // the same table, always: a set with at most kListCap members lives ONLY in its list (its dense row is
                               // all-zero; "lists first", index maintenance section), so every reader of pod sets starts here
};

// Short pod lists.  A slot's pod set is a list of 16-bit pod ids (ascending) while it has at most kListCap members -- 64 bytes
// instead of the 512 of a dense row at P = 4096, and what a prefix block's pod set looks like in practice (a block is cached on a
// handful of replicas); only a larger set lives in the slot's dense row (one bit per pod).  The pick kernel reads the lists of a
// request's hits with ONE 16-byte load per lane; a hit whose list has overflowed sends the request to the dense route, where
// every hit's set becomes a lane word -- from its row when overflowed, expanded from its list otherwise (set_from_list).
// Layout (u16 view, 32 entries): four 16-byte chunks; id number j lives in chunk j & 3 at position j >> 2, so that the ids
// of a short list are spread over the four lanes that read the slot; entries 6..7 of chunk 0 (dword 3) are the 32-bit count;
// unused entries are 0xFFFF.  count > kListCap: overflowed (ids unspecified), the dense row alone is authoritative.
constexpr uint32_t kListCap = 24u;
constexpr uint32_t kListDwords = 16u;
constexpr uint32_t kListNone = 0xFFFFu;
__host__ __device__ __forceinline__ constexpr uint32_t list_pos(uint32_t j) { return 8u * (j & 3u) + (j >> 2); }   // u16 index of id j

// Byte offsets of the per-adapter tables at the head of a snapshot blob (eppk.hip lays the blob out with the same constants).
template <typename LW> struct SnapOff {
  static constexpr uint32_t topv = 0u;                                   // f64 [129][64]
  static constexpr uint32_t topi = 129u * 64u * 8u;                      // u32 [129][64]
  static constexpr uint32_t thi = topi + 129u * 64u * 4u;                // LW  [129][64]
  static constexpr uint32_t tlo = thi + 129u * 64u * (uint32_t)sizeof(LW);
  static constexpr uint32_t thl = tlo + 129u * 64u * (uint32_t)sizeof(LW);   // LW [129][64][2]: the same two planes, {hi, lo} of a lane word side by
                                                                             // side (pick_quad_kernel: one load, one cache line per listed pod)
  static constexpr uint32_t top16 = thl + 2u * 129u * 64u * (uint32_t)sizeof(LW);   // {f64 T, u32 pod, u32 0} [129][16]: the first 16 entries of
                                                                                     // every top table again, value and pod side by side
                                                                                     // (pick_quad_kernel: one 16-byte load per lane)
  static constexpr uint32_t end = top16 + 129u * 16u * 16u;
};
struct TopEntry { double t; uint32_t p, pad; };

struct KChain {            // the whole weighted chain (generic kernel)
  uint32_t n;
  uint32_t kind[8];
  double   w[8];
};

struct KTail {             // the request-dependent tail after fusion (fast kernel)
  double lw[4];            // clamp01(tier score) * w_lora, tier 0..3 = {0.0, 0.6, 0.8, 1.0}
  double wp;               // (double) w_prefix
  // GEN instantiations: the chain behind the fused leading pod-only scorers, in chain order.  kind 0 = LORA, 1 = PREFIX,
  // 2 / 3 = a pod-only scorer whose exact per-pod product clamp01(s) * w is the array post[0] / post[1] of the snapshot.
  uint32_t n_tail;
  uint32_t kind[4];
};

// ---- small device helpers --------------------------------------------------------------------

// Home bucket of a key: top log2(buckets) bits of a 32-bit multiplicative hash of the folded key (block hashes are XXH64
// outputs already; the multiply only guards against structured keys).  shift = 32 - log2(buckets).
// One v_xor + one v_mul_lo_u32 + one shift (a 64-bit multiply costs 6 VALU ops, three of them quarter rate).
__device__ __forceinline__ uint32_t home_bucket(uint64_t h, uint32_t shift) {
  return (((uint32_t)h ^ (uint32_t)(h >> 32)) * kHomeMul) >> shift;
}

__device__ __forceinline__ double clamp01(double s) {
  if (!(s >= 0.0)) return 0.0;
  if (s > 1.0) return 1.0;
  return s;
}

// Look one hash up, one lane per key (generic kernel, long-chunk fallback); returns its slot or kNotFound.
__device__ __forceinline__ uint32_t probe(const KIndex& ix, uint64_t h, bool active) {
  if (!active || ix.slots == 0) return kNotFound;
  if (h == 0) return ix.keys[ix.slots] ? ix.slots : kNotFound;             // reserved hashes: presence words
  if (h == kTomb) return ix.keys[ix.slots + 1u] ? ix.slots + 1u : kNotFound;
  const uint32_t bmask = (ix.slots / kBucket) - 1u;
  uint32_t b = home_bucket(h, ix.shift);
#pragma unroll 1
  for (uint32_t n = 0; n <= bmask; ++n) {
    const uint64_t* kb = ix.keys + (size_t)b * kBucket;
#pragma unroll 1
    for (uint32_t i = kKeySub0; i < kBucket; ++i) {
      const uint64_t k = kb[i];
      if (k == h) return b * kBucket + i;
      if (k == 0) return kNotFound;                 // buckets fill front to back and never shrink: an empty word ends the search
    }
    if (!(kb[0] & 1ull)) return kNotFound;          // full but never overflowed
    b = (b + 1) & bmask;
  }
  return kNotFound;
}

// Bit-sliced counter: c[k] holds bit k of 8*sizeof(LW) independent counters. add a 0/1 vector.
template <typename LW, int NPL>
__device__ __forceinline__ void planes_add(LW (&c)[NPL], LW m) {
  LW carry = m;
#pragma unroll
  for (int k = 0; k < NPL; ++k) {
    const LW t = c[k] & carry;
    c[k] ^= carry;
    carry = t;
  }
}

template <typename LW> struct lane_word;
template <> struct lane_word<uint16_t> { static constexpr int halves = 1; static constexpr int bits = 16; };
template <> struct lane_word<uint32_t> { static constexpr int halves = 1; static constexpr int bits = 32; };
template <> struct lane_word<uint64_t> { static constexpr int halves = 2; static constexpr int bits = 64; };

template <typename LW>
__device__ __forceinline__ uint32_t half32(LW x, int h) {
  if constexpr (sizeof(LW) == 8) return (uint32_t)(x >> (32 * h));
  else return (uint32_t)x;
}

__device__ __forceinline__ double shfl_xor_f64(double v, int off) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __shfl_xor(lo, off);
  hi = __shfl_xor(hi, off);
  return __hiloint2double(hi, lo);
}

// (score desc, index asc) argmax across the 64 lanes; every lane ends with the winner.
__device__ __forceinline__ void wave_argmax(double& best, uint32_t& bidx) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const double ob = shfl_xor_f64(best, off);
    const uint32_t oi = (uint32_t)__shfl_xor((int)bidx, off);
    if (ob > best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
  }
}

__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// The dword of a list line that lane l reads on the dense routes: lanes 0..23 the one that holds id l, the others the count.
__device__ __forceinline__ uint32_t list_lane_dw(int lane) { return (uint32_t)lane < kListCap ? 4u * ((uint32_t)lane & 3u) + ((uint32_t)lane >> 3) : 3u; }
// A listed set as a lane word: lane l < cnt ORs the bit of "its" id into the owning lane's word in LDS (lv = the dword it read,
// list_lane_dw), then every lane picks its word up and leaves zero behind.  Six LDS operations per set whatever its size.
template <typename LW>
__device__ __forceinline__ LW set_from_list(uint32_t lv, uint32_t cnt, int lane, LW* scr) {
  if (cnt == 0u) return (LW)0;                                   // (wave-uniform: the all-zero row behind a request's last hit)
  if ((uint32_t)lane < cnt) {
    const uint32_t id = (((uint32_t)lane >> 2) & 1u) ? (lv >> 16) : (lv & 0xFFFFu);
    const uint32_t e = id & 63u, j = id >> 6;
    if (j < 8u * (uint32_t)sizeof(LW)) {
      if constexpr (sizeof(LW) == 8) atomicOr((unsigned long long*)scr + e, 1ull << j);
      else if constexpr (sizeof(LW) == 4) atomicOr((unsigned int*)scr + e, 1u << j);
      else atomicOr((unsigned int*)scr + (e >> 1), (1u << j) << (16u * (e & 1u)));
    }
  }
  wave_lds_fence();
  const LW w = scr[lane];
  scr[lane] = (LW)0;
  wave_lds_fence();
  return w;
}
// The prefix walk of one request (SEMANTICS.md §3 PREFIX) into bit-sliced counters.
// Returns the number of non-empty pod sets added (= leading non-empty look-ups).  scr: this wavefront's 64 lane words of LDS
// (all-zero between uses): a set with at most kListCap members is expanded from its list, a larger one read from its dense row.
template <typename LW, int NPL>
__device__ __forceinline__ uint32_t prefix_walk(const KIndex& ix, const uint64_t* hs, uint32_t nb, int lane,
                                                LW (&c)[NPL], LW* scr) {
  const LW* bm = (const LW*)ix.bitmaps;
  const uint32_t ldw = list_lane_dw(lane);
  uint32_t hits = 0;
  bool stop = false;
  for (uint32_t b0 = 0; b0 < nb && !stop; b0 += 64) {
    const uint32_t i = b0 + (uint32_t)lane;
    const bool act = i < nb;
    const uint64_t h = act ? hs[i] : 0;
    const uint32_t slot = probe(ix, h, act);               // all of the chunk's keys in parallel
    const unsigned long long found = __ballot(slot != kNotFound);
    const uint32_t chunk = (nb - b0) < 64u ? (nb - b0) : 64u;
    const uint32_t m = (~found == 0ull) ? 64u : (uint32_t)__builtin_ctzll(~found);  // leading found
    for (uint32_t k0 = 0; k0 < m && !stop; k0 += 8) {      // 8 independent list-line loads in flight
      uint32_t lv[8];
      LW w[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const uint32_t k = k0 + (uint32_t)u;
        const uint32_t s = __builtin_amdgcn_readlane(slot, (k < m) ? k : 0);
        lv[u] = (k < m) ? ix.lists_all[(size_t)s * kListDwords + ldw] : 0u;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const uint32_t k = k0 + (uint32_t)u;
        const uint32_t cnt = (uint32_t)__builtin_amdgcn_readlane((int)lv[u], (int)kListCap);
        if (cnt > kListCap) {
          const uint32_t s = __builtin_amdgcn_readlane(slot, (k < m) ? k : 0);
          w[u] = bm[(size_t)s * 64u + (uint32_t)lane];
        } else {
          w[u] = set_from_list<LW>(lv[u], cnt, lane, scr);
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const uint32_t k = k0 + (uint32_t)u;
        if (k < m && !stop) {
          if (__ballot(w[u] != 0) == 0ull) stop = true;    // key present but pod set empty
          else { planes_add<LW, NPL>(c, w[u]); ++hits; }
        }
      }
    }
    if (m < chunk) stop = true;                            // first absent key ends the walk
  }
  return hits;
}

// ---- carry-save counting ---------------------------------------------------------------------
template <typename LW>
__device__ __forceinline__ void half_add(LW a, LW b, LW& sum, LW& carry) {
  sum = a ^ b;
  carry = a & b;
}

// Transpose one natural-layout candidate mask row ([J] u64, bit p%64 of word p/64) into a lane word.
// `mine` = natural word `lane` of a pod set (lanes >= J: anything) -> the set's lane word (bit j = pod j*64 + lane)
template <typename LW>
__device__ __forceinline__ LW transpose_words(const uint64_t mine, uint32_t J, int lane) {
  LW out = 0;
  for (uint32_t j = 0; j < J; ++j) {
    const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)mine, j);
    const uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(mine >> 32), j);
    const uint64_t mj = ((uint64_t)hi << 32) | lo;
    out |= (LW)((LW)((mj >> lane) & 1ull) << j);
  }
  return out;
}
template <typename LW>
__device__ __forceinline__ LW transpose_mask(const uint64_t* row, uint32_t J, int lane) {
  const uint64_t mine = ((uint32_t)lane < J) ? row[lane] : 0ull;
  LW out = 0;
  for (uint32_t j = 0; j < J; ++j) {
    const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)mine, j);
    const uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(mine >> 32), j);
    const uint64_t mj = ((uint64_t)hi << 32) | lo;
    out |= (LW)((LW)((mj >> lane) & 1ull) << j);
  }
  return out;
}

// Lane word with bit j set iff pod j*64+lane exists.
template <typename LW>
__device__ __forceinline__ LW valid_word(uint32_t n_pods, int lane) {
  // pods owned by this lane: j < ceil((n_pods - lane)/64)
  const uint32_t cnt = ((uint32_t)lane < n_pods) ? (n_pods - (uint32_t)lane + 63u) / 64u : 0u;
  if (cnt >= (uint32_t)lane_word<LW>::bits) return (LW)~(LW)0;
  return (LW)(((LW)1 << cnt) - 1);
}

// ---- FAST pick kernel --------------------------------------------------------------------------
// Chain = [pod-only scorers fused into base] ++ tail, tail in {∅, L, P, LP, PL}; no candidate mask.
//
// Sparse evaluation.  Let M = {p : matched[p] > 0} for this request.  For p ∉ M the prefix term is
// pw[0] = ±0.0, and adding ±0.0 to a value that is never −0.0 is the identity (base and base+lw are sums
// that start from +0.0), so total[p] == T_a[p] = base[p] (+ lw[tier(a,p)]) EXACTLY — a quantity that
// depends only on (adapter, pod).  The host therefore publishes, per adapter, the 64 best pods by
// (T desc, p asc).  Per request the kernel
//   1. walks the prefix index (the only HBM-heavy part) into bit-sliced counters (carry-save tree),
//   2. takes the first table entry that is not in M          -> best pod outside M,
//   3. evaluates the full expression only for the pods in M  -> best pod inside M,
//   4. merges both under (score desc, index asc).
// If all 64 table entries are in M (a prefix cached almost everywhere) step 2 becomes a scan of T_a
// over the pods outside M (same arithmetic, base[] read from global memory).
//
// base[] (and the exact prefix-term table) are staged once per workgroup in LDS: the evaluation of a pod of M is then
// two LDS reads instead of two dependent L2 round trips.  All per-request addressing is scalar (the wave id is made
// uniform with readfirstlane), the counting tree uses v_bitop3_b32 full adders (2 VALU ops per 32 pods), and the
// argmax is a DPP max reduction; the row of the next request is prefetched while the current one is evaluated.

template <bool HAS_L, bool HAS_P, bool P_FIRST>
__device__ __forceinline__ double eval_total(double base, double lterm, double pterm) {
  double t = base;
  if (HAS_L && HAS_P) {
    if (P_FIRST) { t = t + pterm; t = t + lterm; }
    else { t = t + lterm; t = t + pterm; }
  } else if (HAS_L) {
    t = t + lterm;
  } else if (HAS_P) {
    t = t + pterm;
  }
  return t;
}

__device__ __forceinline__ double tier_term(const KTail& tl, uint32_t tier) {
  return tier == 3u ? tl.lw[3] : tier == 2u ? tl.lw[2] : tier == 1u ? tl.lw[1] : tl.lw[0];
}

// Exact evaluation of one request over its CANDIDATES ONLY, every scorer in chain order (the generic kernel's
// arithmetic), each lane walking the set bits of its candidate word.  Used by the masked fast kernel when the
// request's QUEUE normalisers differ from the snapshot-wide ones (so base[] / the top tables do not apply);
// cost is proportional to the candidates per lane — small subsets (the common reason for that case) are cheap.
template <typename LW, int NPL>
// `cand` = the request's candidates (the QUEUE normalisers range over all of them); `eval` = the ones to evaluate (a fallback
// round excludes the pods already reported).
__device__ __forceinline__ void masked_exact(const KSnap& sn, const KChain& ch, LW cand, LW eval, const LW (&c)[NPL], LW thi, LW tlo,
                                             uint32_t nb, int lane, double& best, uint32_t& bidx) {
  bool has_q = false;
  for (uint32_t k = 0; k < ch.n; ++k) has_q |= ch.kind[k] == 1u;
  uint32_t qmin = 0, qmax = 0;
  if (has_q) {
    uint32_t mn = 0xFFFFFFFFu, mx = 0u;
    LW rem = cand;
    while (__any(rem != 0)) {
      if (rem != 0) {
        const uint32_t j = (sizeof(LW) == 8) ? (uint32_t)__builtin_ctzll((unsigned long long)rem) : (uint32_t)__builtin_ctz((uint32_t)rem);
        rem = (LW)(rem & (LW)(rem - 1));
        const uint32_t q = sn.queue[j * 64u + (uint32_t)lane];
        mn = q < mn ? q : mn;
        mx = q > mx ? q : mx;
      }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const uint32_t omn = (uint32_t)__shfl_xor((int)mn, off), omx = (uint32_t)__shfl_xor((int)mx, off);
      mn = omn < mn ? omn : mn;
      mx = omx > mx ? omx : mx;
    }
    qmin = mn; qmax = mx;
  }
  const double qden = (double)(qmax - qmin);
  LW rem = eval;
  while (__any(rem != 0)) {
    if (rem != 0) {
      const uint32_t j = (sizeof(LW) == 8) ? (uint32_t)__builtin_ctzll((unsigned long long)rem) : (uint32_t)__builtin_ctz((uint32_t)rem);
      rem = (LW)(rem & (LW)(rem - 1));
      const uint32_t p = j * 64u + (uint32_t)lane;
      const uint32_t tier = (uint32_t)(((thi >> j) & 1) << 1) | (uint32_t)((tlo >> j) & 1);
      uint32_t cnt = 0;
#pragma unroll
      for (int k = 0; k < NPL; ++k) cnt |= (uint32_t)((c[k] >> j) & 1) << k;
      double t = 0.0;
      for (uint32_t k = 0; k < ch.n; ++k) {
        double sc;
        switch (ch.kind[k]) {
          case 1u: sc = (qmax == qmin) ? 1.0 : (double)(qmax - sn.queue[p]) / qden; break;
          case 2u: sc = 1.0 - sn.kv[p]; break;
          case 3u: sc = tier == 3u ? 1.0 : tier == 2u ? 0.8 : tier == 1u ? 0.6 : 0.0; break;
          default: sc = nb ? (double)cnt / (double)nb : 0.0; break;
        }
        t = t + clamp01(sc) * ch.w[k];
      }
      if (t > best) { best = t; bidx = p; }
    }
  }
}

// ---- gfx950 3-input bit ops + DPP reductions ---------------------------------------------------
// v_bitop3_b32 (new in gfx950): any 3-input boolean function in ONE VALU op.  A full adder on bit vectors is
// then 2 ops per 32 bits (sum = a^b^c: truth table 0x96; carry = maj(a,b,c): 0xE8) instead of 3 with v_bfi.
template <int TT>
__device__ __forceinline__ uint32_t bitop3_32(uint32_t a, uint32_t b, uint32_t c) {
  return __builtin_amdgcn_bitop3_b32(a, b, c, TT);
}
template <int TT, typename LW>
__device__ __forceinline__ LW bitop3(LW a, LW b, LW c) {
  if constexpr (sizeof(LW) == 8)
    return ((uint64_t)bitop3_32<TT>((uint32_t)(a >> 32), (uint32_t)(b >> 32), (uint32_t)(c >> 32)) << 32) |
           bitop3_32<TT>((uint32_t)a, (uint32_t)b, (uint32_t)c);
  else
    return (LW)bitop3_32<TT>((uint32_t)a, (uint32_t)b, (uint32_t)c);
}
template <typename LW>
__device__ __forceinline__ void fa3(LW a, LW b, LW c, LW& sum, LW& carry) {
  sum = bitop3<0x96, LW>(a, b, c);
  carry = bitop3<0xE8, LW>(a, b, c);
}

// 16 -> 5 carry-save tree: b[k] = bit k of the per-pod number of rows (0..16) containing the pod.  11 full + 4 half adders.
template <typename LW>
__device__ __forceinline__ void csa16(const LW (&w)[16], LW (&b)[5]) {
  LW s1, s2, s3, s4, s5, c1, c2, c3, c4, c5;
  fa3<LW>(w[0], w[1], w[2], s1, c1);
  fa3<LW>(w[3], w[4], w[5], s2, c2);
  fa3<LW>(w[6], w[7], w[8], s3, c3);
  fa3<LW>(w[9], w[10], w[11], s4, c4);
  fa3<LW>(w[12], w[13], w[14], s5, c5);
  LW t1, t2, d1, d2, d3;
  fa3<LW>(s1, s2, s3, t1, d1);
  fa3<LW>(s4, s5, w[15], t2, d2);
  half_add<LW>(t1, t2, b[0], d3);                 // ones done; twos: c1..c5, d1, d2, d3
  LW u1, u2, e1, e2, e3, e4, v1;
  fa3<LW>(c1, c2, c3, u1, e1);
  fa3<LW>(c4, c5, d1, u2, e2);
  fa3<LW>(u1, u2, d2, v1, e3);
  half_add<LW>(v1, d3, b[1], e4);                 // twos done; fours: e1..e4
  LW x1, f1, f2;
  fa3<LW>(e1, e2, e3, x1, f1);
  half_add<LW>(x1, e4, b[2], f2);                 // fours done; eights: f1, f2
  half_add<LW>(f1, f2, b[3], b[4]);
}
// 8 -> 4 tree (4 full + 3 half adders)
template <typename LW>
__device__ __forceinline__ void csa8(const LW (&w)[8], LW (&b)[4]) {
  LW s1, c1, s2, c2, s3, c3, c4, s5, c5, c6;
  fa3<LW>(w[0], w[1], w[2], s1, c1);
  fa3<LW>(w[3], w[4], w[5], s2, c2);
  fa3<LW>(w[6], w[7], s1, s3, c3);
  half_add<LW>(s2, s3, b[0], c4);
  fa3<LW>(c1, c2, c3, s5, c5);
  half_add<LW>(s5, c4, b[1], c6);
  half_add<LW>(c5, c6, b[2], b[3]);
}
// c += b (b is an NB-bit bit-sliced number); ripple carry.
template <typename LW, int NPL, int NB>
__device__ __forceinline__ void planes_addn(LW (&c)[NPL], const LW (&b)[NB]) {
  static_assert(NPL >= NB, "planes");
  LW carry, t;
  half_add<LW>(c[0], b[0], t, carry); c[0] = t;
#pragma unroll
  for (int k = 1; k < NB; ++k) { fa3<LW>(c[k], b[k], carry, t, carry); c[k] = t; }
#pragma unroll
  for (int k = NB; k < NPL; ++k) { half_add<LW>(c[k], carry, t, carry); c[k] = t; }
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double v) {   // lanes outside ROW_MASK keep v
  if constexpr (ROW_MASK == 0xf) {     // every lane has a source: no tied "old" operand, so no register copies
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
  } else {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
  }
}
__device__ __forceinline__ double vmax_f64(double a, double b) {   // inputs are never NaN (SEMANTICS.md: clamp01 kills NaN)
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// max over the wavefront, returned wave-uniform.  quad xor 1, quad xor 2, half mirror, mirror (all 16 lanes of a
// row agree), row_bcast:15 into rows 1/3, row_bcast:31 into rows 2/3; lane 63 holds the result.
__device__ __forceinline__ double wave_max_f64(double v) {
  v = vmax_f64(v, dpp_f64<0xB1, 0xf>(v));
  v = vmax_f64(v, dpp_f64<0x4E, 0xf>(v));
  v = vmax_f64(v, dpp_f64<0x141, 0xf>(v));
  v = vmax_f64(v, dpp_f64<0x140, 0xf>(v));
  v = vmax_f64(v, dpp_f64<0x142, 0xa>(v));
  v = vmax_f64(v, dpp_f64<0x143, 0xc>(v));
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_min_u32(uint32_t v) {
  uint32_t o;
  if constexpr (ROW_MASK == 0xf) o = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, false);
  else o = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROW_MASK, 0xf, false);
  return o < v ? o : v;
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
  v = dpp_min_u32<0xB1, 0xf>(v);
  v = dpp_min_u32<0x4E, 0xf>(v);
  v = dpp_min_u32<0x141, 0xf>(v);
  v = dpp_min_u32<0x140, 0xf>(v);
  v = dpp_min_u32<0x142, 0xa>(v);
  v = dpp_min_u32<0x143, 0xc>(v);
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// (score desc, index asc) argmax across the wavefront, wave-uniform result: f64 max by DPP, then the lowest pod index
// among the lanes that hold it (one ballot + readlane when the maximum is unique, a u32 min reduction otherwise).
__device__ __forceinline__ void wave_argmax_dpp(double& best, uint32_t& bidx) {
  const double wmax = wave_max_f64(best);
  const bool tie = best == wmax;
  const unsigned long long tm = __ballot(tie);
  uint32_t widx;
  if (__builtin_popcountll(tm) == 1) widx = (uint32_t)__builtin_amdgcn_readlane((int)bidx, __builtin_ctzll(tm));
  else widx = wave_min_u32(tie ? bidx : kNoPod);
  best = wmax;
  bidx = widx;
}

// ---- pair probe (fast kernel): TWO lanes per key, 32 keys per wavefront instruction group -------------------------
// Lane l serves key l>>1; the even lane reads words 0..3 of the key's home bucket (header + 3 keys), the odd lane words
// 4..7: two 16-byte loads per lane, and the whole look-up of a request's first 32 hashes is one independent gather --
// no probe chain.
constexpr uint32_t kKeysPerProbe = 32u;

// Work list of the fast kernel's WL instantiations.  pick_quad_kernel scores what it can and DEFERS the rest -- one private
// segment per wavefront: list[seg * cap + j], j < cnt[seg] -- and pick_fast_kernel<..., WL = true> then runs over exactly those
// requests (same stream, right behind it).
struct KWork {
  const uint32_t* cnt;      // [n_segs] requests in each segment
  const uint32_t* list;     // [n_segs][cap] request indices
  const uint32_t* total;    // sum of cnt[] (the quad kernel adds to it only from wavefronts that deferred something)
  uint32_t* report;         // pinned HOST word: the kernel stores *total there (the library's "is the quad pass paying off" feedback)
  uint32_t cap, n_segs;
};

struct ReqRegs {            // pipeline registers of one request
  uint64_t hdr, h;          // row header; the hash this lane pair probes (landing registers of the row prefetch)
  uint4 kw[2];              // this lane's half of the home bucket (landing registers of the key gather)
  uint32_t bkt;             // home bucket
};

__device__ __forceinline__ uint32_t dpp_xor1(uint32_t v) {   // value of the other lane of the pair
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true);
}

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint64_t buffer_load_u64(__amdgpu_buffer_rsrc_t rs, uint32_t voff, uint32_t soff) {
  const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)voff, (int)soff, 0);
  return ((uint64_t)v.y << 32) | v.x;
}
// Home bucket of q.h (q.h must already be 0 in lanes without a key).
__device__ __forceinline__ void pair_probe_prepare(const KIndex& ix, ReqRegs& q) { q.bkt = home_bucket(q.h, ix.shift); }
// Issue the two 16-byte loads of this lane's half of the home bucket (rk = buffer descriptor of the index allocation).
__device__ __forceinline__ void pair_probe_issue(__amdgpu_buffer_rsrc_t rk, uint32_t keys_off, ReqRegs& q, int lane) {
#if defined(EPPK_DBG_KEYS_NONE)
  q.kw[0] = q.kw[1] = make_uint4(0, 0, 0, 0);
#else
  const uint32_t voff = q.bkt * (kBucket * 8u) + ((uint32_t)lane & 1u) * 32u;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rk, (int)(voff + 16u * (uint32_t)i), (int)keys_off, 0);
    q.kw[i] = make_uint4(v.x, v.y, v.z, v.w);
  }
#endif
}

__device__ __forceinline__ uint64_t u64_of(uint32_t lo, uint32_t hi) { return ((uint64_t)hi << 32) | lo; }

// Position (0..3) of h among this lane's 4 bucket words, 4 if absent.  Words 0..2 of a bucket -- the even lane's first three -- are
// flags and meta dwords, not keys (kBucket).
__device__ __forceinline__ uint32_t match4(const uint4 (&kw)[2], uint64_t h, bool even) {
  uint32_t pos = 4u;
  if (u64_of(kw[1].z, kw[1].w) == h) pos = 3u;
  if (!even && u64_of(kw[1].x, kw[1].y) == h) pos = 2u;
  if (!even && u64_of(kw[0].z, kw[0].w) == h) pos = 1u;
  if (!even && u64_of(kw[0].x, kw[0].y) == h) pos = 0u;
  return pos;
}

// Turn the loaded buckets into the request's leading-hit count m (<= nchunk <= 32) and row map: slot_eff of lane pair k
// = slot of hit k for k < m, the all-zero row otherwise (rows are addressed with v_readlane(slot_eff, 2k)).
// Rare paths: reserved hashes 0 / ~0 (presence words) and keys absent from an overflowed bucket (next buckets).
__device__ __forceinline__ uint32_t pair_probe_finish(const KIndex& ix, const ReqRegs& q, uint32_t nchunk, int lane, uint32_t& slot_eff) {
  const uint32_t sub = (uint32_t)lane & 1u, ki = (uint32_t)lane >> 1;
  const bool act = ki < nchunk;
  const uint64_t h = q.h;
  uint32_t slot;
#ifdef EPPK_DBG_SKIP_KEYS   // timing experiment only (wrong results): 16 pseudo-hits
  slot = (ki < 16u && act) ? q.bkt * kBucket + kKeySub0 : kNotFound;
#else
  if (__builtin_expect(__any(act && (h + 1ull) <= 1ull), 0)) {             // h == 0 or h == ~0 somewhere in the request (rare)
    slot = probe(ix, h, act);
  } else {
    const uint32_t pos = match4(q.kw, h, sub == 0u);
    uint32_t s = pos < 4u ? q.bkt * kBucket + sub * 4u + pos : kNotFound;
    const uint32_t so = dpp_xor1(s);
    s = so < s ? so : s;
    uint32_t ovf = sub == 0u ? (q.kw[0].x & 1u) : 0u;  // header bit 0, seen by the even lane
    ovf |= dpp_xor1(ovf);
    bool pend = act && s == kNotFound && ovf != 0u;
    if (__builtin_expect(__any(pend), 0)) {            // absent from an overflowed bucket: walk the following buckets
      const uint32_t bmask = (ix.slots / kBucket) - 1u;
      uint32_t b = q.bkt;
#pragma unroll 1
      for (uint32_t n = 0; n < bmask && __any(pend); ++n) {
        if (pend) {
          b = (b + 1) & bmask;
          const uint64_t* kb = ix.keys + (size_t)b * kBucket + (size_t)sub * 4u;
          uint32_t s2 = kNotFound;
#pragma unroll 1
          for (uint32_t i = 0; i < 4u; ++i)
            if (kb[i] == h && sub * 4u + i >= kKeySub0) s2 = b * kBucket + sub * 4u + i;
          uint32_t o2 = sub == 0u ? (uint32_t)(kb[0] & 1ull) : 0u;
          const uint32_t s2o = dpp_xor1(s2);
          s2 = s2o < s2 ? s2o : s2;
          o2 |= dpp_xor1(o2);
          if (s2 != kNotFound) { s = s2; pend = false; }
          else if (o2 == 0u) pend = false;
        }
      }
    }
    slot = act ? s : kNotFound;
  }
#endif
  const unsigned long long fm = __ballot(slot != kNotFound);      // both lanes of a pair agree
  const uint32_t m = (~fm == 0ull) ? kKeysPerProbe : (uint32_t)__builtin_ctzll(~fm) >> 1;
  slot_eff = (ki < m) ? slot : ix.slots + 2u;
  return m;
}

// Row loads.  A row address is wave-uniform (slot from v_readlane) plus lane * sizeof(LW).  Tables below 4 GiB are read
// through ONE raw buffer descriptor with the row's byte offset in the instruction's SGPR offset operand: zero VALU ops and
// zero 64-bit address arithmetic per row (buffer_load ... v_lane_off, s[rsrc], s_row_off offen).  Hit k of the request is in
// lane pair k of slot_eff.
template <typename LW>
__device__ __forceinline__ LW buffer_load_lw(__amdgpu_buffer_rsrc_t rs, uint32_t voff, uint32_t soff) {
  if constexpr (sizeof(LW) == 8) {
    const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)voff, (int)soff, 0);
    return ((uint64_t)v.y << 32) | v.x;
  } else if constexpr (sizeof(LW) == 4) {
    return (LW)__builtin_amdgcn_raw_buffer_load_b32(rs, (int)voff, (int)soff, 0);
  } else {
    return (LW)__builtin_amdgcn_raw_buffer_load_b16(rs, (int)voff, (int)soff, 0);
  }
}
// BIG (index of 4 GiB and more, sized for the 288 GB of HBM): a row's address is wave-uniform -- base + slot * row bytes, 64-bit
// scalar arithmetic on the slot that v_readlane delivers -- so the row is read with a global load whose base sits in an SGPR pair
// (saddr form) and whose lane offset is the 32-bit lane * sizeof(LW): still no per-lane 64-bit address arithmetic.
// (Round 1 used a STRUCTURED buffer descriptor here, stride = one row, slot = buffer index.  Its index * stride product wraps at
// 4 GiB on gfx950: rows beyond that offset read as garbage -- found in round 2 by the dense-rows-only run of the 8.6 GB
// closed-loop test; the round-1 test stopped at exactly 4 GiB.)
struct RowSrc {                  // how the fast kernel reaches the pod sets of a request's hits on its DENSE route
  __amdgpu_buffer_rsrc_t raw;    // small index: raw descriptor over rows + keys, rows addressed by SGPR byte offsets
  const uint8_t* base;           // BIG: the rows' base address
  const uint32_t* lists;         // KIndex::lists_all: where a set with at most kListCap members lives
  void* scr;                     // this wavefront's 64 lane words of LDS, all-zero between uses (set_from_list)
};
// The pod sets of hits k0 .. k0 + N - 1 (hit k of the request is in lane pair k of slot_eff) as lane words: every hit's list line
// first (one dword per lane, N loads in flight), then per hit either the expansion of its list or -- overflowed -- its dense row.
template <typename LW, int N, bool BIG>
__device__ __forceinline__ void load_rows(const RowSrc& src, uint32_t slot_eff, uint32_t k0, int lane, LW (&w)[N]) {
  const uint32_t voff = (uint32_t)lane * (uint32_t)sizeof(LW);
  const uint32_t ldw = list_lane_dw(lane);
  uint32_t lv[N];
#pragma unroll
  for (int u = 0; u < N; ++u) {
    const uint32_t s = (uint32_t)__builtin_amdgcn_readlane((int)slot_eff, (int)(2u * (k0 + (uint32_t)u)));
    lv[u] = src.lists[(size_t)s * kListDwords + ldw];
  }
  const uint32_t roff = slot_eff * (uint32_t)(64u * sizeof(LW));
#pragma unroll
  for (int u = 0; u < N; ++u) {
    const uint32_t cnt = (uint32_t)__builtin_amdgcn_readlane((int)lv[u], (int)kListCap);
    if (__builtin_expect(cnt > kListCap, 0)) {
      if constexpr (BIG) {
        const uint32_t s = (uint32_t)__builtin_amdgcn_readlane((int)slot_eff, (int)(2u * (k0 + (uint32_t)u)));
        const LW* rowp = (const LW*)(src.base + (size_t)s * (size_t)(64u * sizeof(LW)));     // wave-uniform 64-bit base
        w[u] = rowp[lane];
      } else {
        const uint32_t soff = (uint32_t)__builtin_amdgcn_readlane((int)roff, (int)(2u * (k0 + (uint32_t)u)));
        w[u] = buffer_load_lw<LW>(src.raw, voff, soff);
      }
    } else {
      w[u] = set_from_list<LW>(lv[u], cnt, lane, (LW*)src.scr);
    }
  }
}

// rows [k0, m) of slot_eff added into non-zero counters, 16 in flight (8 for a short tail)
template <typename LW, int NPL, bool BIG>
__device__ __forceinline__ void count_more(const RowSrc& rs, uint32_t slot_eff, uint32_t k0, uint32_t m, int lane, LW (&c)[NPL]) {
  for (; k0 + 8u < m; k0 += 16u) {
    LW w[16], b[5];
    load_rows<LW, 16, BIG>(rs, slot_eff, k0, lane, w);
    csa16<LW>(w, b);
    planes_addn<LW, NPL, 5>(c, b);
  }
  if (k0 < m) {
    LW w[8], b[4];
    load_rows<LW, 8, BIG>(rs, slot_eff, k0, lane, w);
    csa8<LW>(w, b);
    planes_addn<LW, NPL, 4>(c, b);
  }
}

template <typename LW>
__device__ __forceinline__ uint32_t ctz_lw(LW x) {
  if constexpr (sizeof(LW) == 8) return (uint32_t)__builtin_ctzll((unsigned long long)x);
  else return (uint32_t)__builtin_ctz((uint32_t)x);
}

// masked_exact for the LIST routes (fast kernel) and for pick_quad_kernel: the same expressions in the same order (every scorer of the
// chain, binary64, the request's own QUEUE normalisers), without dense rows, carry-save counters or one-candidate trips: the set bits of a
// lane word are taken FOUR at a time, so the gauges of four candidates are in flight together (masked_exact walks one candidate per
// trip behind two dependent L2 round trips).  matched[p] comes from the caller's LDS: a byte histogram (the lists of the request's hits
// counted into it: fast kernel, exact_sweep) or a "listed" bitmap -- a listed candidate is then left out (the caller scores those
// itself: pick_quad_kernel, whose lanes hold the listed pods and their counts) and everybody else has matched 0 (exact_sweep_nat).
struct ExactChain {            // the chain in scalar registers (read once per request: no indexed access to the argument struct in the loops)
  uint32_t n, kinds;           // four bits per scorer
  double w[8];
  bool has_q;
};
__device__ __forceinline__ ExactChain exact_chain(const KChain& ch) {
  ExactChain ec;
  ec.n = ch.n; ec.kinds = 0u; ec.has_q = false;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    ec.w[k] = 0.0;
    if ((uint32_t)k < ec.n) { ec.kinds |= (ch.kind[k] & 15u) << (4 * k); ec.w[k] = ch.w[k]; ec.has_q |= ch.kind[k] == 1u; }
  }
  return ec;
}
// minimum / maximum queue depth over the request's candidates (wave-uniform result)
// (NAT: `cand` is natural word `lane` of the set -- pod lane * 64 + j -- instead of the lane word -- pod j * 64 + lane)
template <typename LW, bool NAT = false>
__device__ __forceinline__ void exact_qrange(const uint32_t* __restrict__ queue, const LW cand, const int lane, uint32_t& qmin, uint32_t& qmax) {
  constexpr int U = 4;
  uint32_t mn = 0xFFFFFFFFu, mx = 0u;
  LW rem = cand;
  while (__any(rem != 0)) {
    uint32_t q[U];
    bool v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      v[u] = rem != 0;
      const uint32_t j = v[u] ? ctz_lw<LW>(rem) : 0u;
      rem = (LW)(rem & (LW)(rem - 1));
      q[u] = queue[NAT ? (v[u] ? (uint32_t)lane * 64u + j : 0u) : j * 64u + (uint32_t)lane];   // (a lane that has run out re-reads pod 0 / pod `lane`: in bounds, never used)
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      mn = (v[u] && q[u] < mn) ? q[u] : mn;
      mx = (v[u] && q[u] > mx) ? q[u] : mx;
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const uint32_t omn = (uint32_t)__shfl_xor((int)mn, off), omx = (uint32_t)__shfl_xor((int)mx, off);
    mn = omn < mn ? omn : mn;
    mx = omx > mx ? omx : mx;
  }
  qmin = mn; qmax = mx;
}
// one candidate's total: masked_exact's loop body (`sp` = its clamped PREFIX score, computed by the caller)
__device__ __forceinline__ double exact_total(const ExactChain& ec, const uint32_t qmin, const uint32_t qmax, const double qden, const uint32_t q, const double kv,
                                              const uint32_t tier, const double sp) {
  const double sq = clamp01((qmax == qmin) ? 1.0 : (double)(qmax - q) / qden);
  const double skv = clamp01(1.0 - kv);
  const double sl = tier == 3u ? 1.0 : tier == 2u ? 0.8 : tier == 1u ? 0.6 : 0.0;
  double t = 0.0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    if ((uint32_t)k < ec.n) {                           // (wave-uniform: scalar branches)
      const uint32_t kd = (ec.kinds >> (4 * k)) & 15u;
      const double sc = kd == 1u ? sq : kd == 2u ? skv : kd == 3u ? sl : sp;
      t = t + sc * ec.w[k];
    }
  }
  return t;
}
template <typename LW>
__device__ __forceinline__ void exact_sweep(const KSnap& sn, const ExactChain& ec, const uint32_t qmin, const uint32_t qmax, const LW cand, const LW thi, const LW tlo,
                                            const uint32_t nb, const int lane, const uint32_t* s_cnt, double& best, uint32_t& bidx) {
  constexpr int U = 4;
  const double qden = (double)(qmax - qmin), nbd = (double)nb;
  LW rem = cand;
  while (__any(rem != 0)) {
    uint32_t q[U], hw[U], jj[U];
    double kv[U];
    bool v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      v[u] = rem != 0;
      jj[u] = v[u] ? ctz_lw<LW>(rem) : 0u;
      rem = (LW)(rem & (LW)(rem - 1));
      const uint32_t p = jj[u] * 64u + (uint32_t)lane;
      q[u] = sn.queue[p];
      kv[u] = sn.kv[p];
      hw[u] = s_cnt[p >> 2];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t j = jj[u], p = j * 64u + (uint32_t)lane;
      const bool take = v[u];
      const uint32_t cnt = (hw[u] >> ((p & 3u) * 8u)) & 0xFFu;
      const uint32_t tier = (uint32_t)(((thi >> j) & 1) << 1) | (uint32_t)((tlo >> j) & 1);
      double sp = 0.0;                                  // (0 / nb == +0.0: the division is skipped where no lane's candidate is listed)
      if (nb != 0u && __any(take && cnt != 0u)) sp = clamp01((double)cnt / nbd);
      const double t = exact_total(ec, qmin, qmax, qden, q[u], kv[u], tier, sp);
      if (take && (t > best || (t == best && p < bidx))) { best = t; bidx = p; }
    }
  }
}
// The sweep of pick_quad_kernel's rows: the candidate words as they lie in LDS (natural layout: lane w holds pods 64 w .. 64 w + 63, no
// transposition), a listed candidate (bit in `bits`) left to the caller, matched = 0 for everybody else; the LoRA tier of a candidate
// from the interleaved {hi, lo} table (one more load beside the two gauges: twelve loads of four candidates in flight per trip).
template <typename LW, bool HAS_L>
__device__ __forceinline__ void exact_sweep_nat(const KSnap& sn, const ExactChain& ec, const uint32_t qmin, const uint32_t qmax, const uint64_t cand, const uint32_t arow,
                                                const int lane, const uint32_t* bits, double& best, uint32_t& bidx) {
  constexpr int U = 4;
  const double qden = (double)(qmax - qmin);
  const LW* thl = (const LW*)((const uint8_t*)sn.blob + SnapOff<LW>::thl) + (size_t)arow * 128u;
  uint64_t rem = cand;
  while (__any(rem != 0ull)) {
    uint32_t q[U], hw[U], jj[U];
    double kv[U];
    LW th[U], tl_[U];
    bool v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      v[u] = rem != 0ull;
      jj[u] = v[u] ? (uint32_t)__builtin_ctzll(rem) : 0u;
      rem &= rem - 1ull;
      const uint32_t p = v[u] ? (uint32_t)lane * 64u + jj[u] : 0u;                // (a lane that has run out, or beyond the last word: pod 0, never used)
      jj[u] = p;
      q[u] = sn.queue[p];
      kv[u] = sn.kv[p];
      hw[u] = bits[p >> 5];
      if (HAS_L) { th[u] = thl[(p & 63u) * 2u]; tl_[u] = thl[(p & 63u) * 2u + 1u]; }   // (pod p: lane word p & 63, bit p >> 6)
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t p = jj[u];
      const bool take = v[u] && !((hw[u] >> (p & 31u)) & 1u);
      uint32_t tier = 0u;
      if (HAS_L) tier = (uint32_t)((((uint64_t)th[u] >> (p >> 6)) & 1ull) << 1) | (uint32_t)(((uint64_t)tl_[u] >> (p >> 6)) & 1ull);
      const double t = exact_total(ec, qmin, qmax, qden, q[u], kv[u], tier, 0.0);
      if (take && (t > best || (t == best && p < bidx))) { best = t; bidx = p; }
    }
  }
}
template <typename LW>
__device__ __forceinline__ void masked_exact_hist(const KSnap& sn, const KChain& ch, const LW cand, const LW thi, const LW tlo, const uint32_t nb,
                                                  const int lane, const uint32_t* s_hist, double& best, uint32_t& bidx) {
  const ExactChain ec = exact_chain(ch);
  uint32_t qmin = 0u, qmax = 0u;
  if (ec.has_q) exact_qrange<LW>(sn.queue, cand, lane, qmin, qmax);
  exact_sweep<LW>(sn, ec, qmin, qmax, cand, thi, tlo, nb, lane, s_hist, best, bidx);
}

// LDS of the fast kernel: base[J*64] f64 | pterm[(B+1)*ld] f64 (when the host built the table).
//
// The request loop is a 3-stage software pipeline per wavefront (unrolled twice so that the stage registers rotate by
// renaming, not by copies -- a copy of a landing register would wait for its load):
//   stage 0  request row of r+2      issued in iteration r   (HBM stream; a full iteration of slack)
//   stage 1  key gather of r+1       issued in iteration r   (needs the row of r+1, prefetched in iteration r-1)
//   stage 2  rows + tables of r, count, evaluate, pick.
// Issue order inside an iteration is rows(r) -> keys(r+1) -> row(r+2): vmcnt retires loads in order, so everything the
// current request waits for is queued AHEAD of the loads that serve later requests.
// WL (work-list) instantiations: the same kernel over the requests pick_quad_kernel deferred (KWork) instead of 0 .. n_reqs - 1.
// The kernel's body as a function of a VIRTUAL grid position: pick_fast_kernel calls it with its own block index and grid size;
// pick_quad_kernel<..., TAIL> calls the work-list form from EVERY workgroup when its own loop is over (`tail`), with the workgroup's
// real position: wavefront w of workgroup b then owns exactly the segment it filled itself -- no second launch, no waiting for any
// other workgroup, and a burst of deferred requests is spread over the whole grid.  smem: the dynamic LDS of the calling kernel
// (the layout below); stat_base: first probe-statistics slot of this grid's wavefronts.
// RESIDENT (pick_resident_kernel): the body is called once per doorbell by a workgroup that stays; `tail` then means "the snapshot's
// tables are in this workgroup's LDS already" (same snapshot as at the previous doorbell): only the barrier of the staging remains.
template <typename LW, int NPL, bool HAS_L, bool HAS_P, bool P_FIRST, bool MASKED, bool BIG, bool GEN, bool TOPK, bool WL = false, bool RESIDENT = false>
__device__ __forceinline__ void pick_fast_body(const uint32_t vblock, const uint32_t vgrid, const uint32_t stat_base, const bool tail, unsigned char* smem,
                                               const KSnap& sn, const KIndex& ix, const KTail& tl, const uint8_t* __restrict__ reqs,
                                               uint32_t stride, uint32_t n_reqs, uint32_t pwn,
                                               const uint64_t* __restrict__ cand_mask, const KChain& ch,
                                               int32_t* __restrict__ out_pick, double* __restrict__ out_score,
                                               unsigned long long* __restrict__ stats, uint32_t topk, const KWork& wk) {
  // (the work list is read with agent-scope loads: in the tail form its writers are other workgroups of the SAME launch, possibly
  // on another XCD, whose stores went through to memory -- pick_quad_kernel -- but not into this XCD's L2)
  auto wl_word = [](const uint32_t* p) -> uint32_t { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  if constexpr (WL) {
    if (!tail) {                                             // (tail: the other workgroups of the launch may still be adding to the total)
      const uint32_t total = wl_word(wk.total);
      if (vblock == 0 && threadIdx.x == 0) *wk.report = total;
      if (total == 0u) return;                               // nothing was deferred: done before any staging
    }
    // a workgroup none of whose wavefronts owns a segment with work leaves as well (tail: the barrier also says that every
    // wavefront of the workgroup is done with the LDS of the quad layout)
    uint32_t mine = 0;
    const uint32_t nw_ = vgrid * (blockDim.x >> 6);
    for (uint32_t seg = vblock * (blockDim.x >> 6) + (threadIdx.x >> 6); seg < wk.n_segs; seg += nw_) mine |= wl_word(&wk.cnt[seg]);
    if (!__syncthreads_or((int)mine)) return;
  }
  double* s_base = (double*)smem;
  double* s_lw = s_base + (size_t)sn.J * 64u;      // [4] LoRA tier terms (an LDS look-up keeps the evaluation loop branch-free)
  double* s_pterm = s_lw + 4;
  // GEN: the per-pod products of the (at most two) pod-only scorers behind LORA / PREFIX, behind the prefix-term table
  double* s_post0 = s_pterm + pwn;
  double* s_post1 = s_post0 + (size_t)sn.J * 64u;
  // the exact prefix-term table exists whenever max_blocks <= 63, i.e. for every NPL == 6 instantiation (eppk.hip)
  constexpr bool pterm_tab = HAS_P && NPL == 6;
  // SPARSE: requests whose hits all have a short pod list are counted from the lists (one 16-byte load per lane instead of
  // 64 * sizeof(LW) bytes per hit) in a per-wave byte histogram in LDS; everything else takes the dense rows as before.
  constexpr bool SPARSE = HAS_P && NPL == 6;      // (TOPK: the uniform-lists route only; anything else falls back to the dense rows)
  LW* s_scr_all = (LW*)(GEN ? s_post1 + (size_t)sn.J * 64u : s_post0);                // [waves][64] lane words: set_from_list's scratch (dense route)
  uint32_t* s_hist_all = (uint32_t*)(s_scr_all + (size_t)(blockDim.x >> 6) * 64u);    // [waves][J * 16] dwords: one byte per pod
  const bool use_lists = SPARSE && ix.lists != nullptr;
  if (!(RESIDENT && tail)) {      // (the histogram and the scratch words are all-zero between two requests, so also between two doorbells)
    for (uint32_t i = threadIdx.x; i < sn.J * 64u; i += blockDim.x) s_base[i] = sn.base[i];
    if (threadIdx.x == 0u) { s_lw[0] = tl.lw[0]; s_lw[1] = tl.lw[1]; s_lw[2] = tl.lw[2]; s_lw[3] = tl.lw[3]; }   // (no dynamic index into the argument struct: that costs scratch)
    if (GEN)
      for (uint32_t i = threadIdx.x; i < sn.J * 64u; i += blockDim.x) { s_post0[i] = sn.post[0][i]; s_post1[i] = sn.post[1][i]; }
    if (pterm_tab)
      for (uint32_t i = threadIdx.x; i < pwn; i += blockDim.x) s_pterm[i] = sn.pterm[i];
    if (use_lists)
      for (uint32_t i = threadIdx.x; i < (blockDim.x >> 6) * sn.J * 16u; i += blockDim.x) s_hist_all[i] = 0u;
    for (uint32_t i = threadIdx.x; i < (blockDim.x >> 6) * 64u; i += blockDim.x) s_scr_all[i] = (LW)0;
  }
  __syncthreads();

  const int lane = (int)(threadIdx.x & 63u);
  const uint32_t wpb = blockDim.x >> 6;
  // wave-uniform ids in SGPRs: every per-request address below is scalar arithmetic
  const uint32_t gwave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(vblock * wpb + (threadIdx.x >> 6)));
  const uint32_t nwaves = (uint32_t)__builtin_amdgcn_readfirstlane((int)(vgrid * wpb));
  // Buffer descriptors (gfx9 word 3: 32-bit data format).  Small index (< 4 GiB): ONE raw descriptor over rows + keys (one
  // allocation, rows first), rows addressed by SGPR byte offsets.  BIG: the rows through wave-uniform 64-bit
  // bases (RowSrc) and a raw descriptor over the keys.  Plus the snapshot tables and the request rows: every hot-loop load is
  // buffer_load(descriptor SGPRs, 32-bit lane offset, SGPR row selector) or a saddr global load -- no 64-bit per-lane pointers.
  RowSrc rs;
  rs.raw = __builtin_amdgcn_make_buffer_rsrc((void*)ix.bitmaps, 0, BIG ? 0 : (int)ix.table_bytes, 0x00020000);
  rs.base = (const uint8_t*)ix.bitmaps;
  rs.lists = ix.lists_all;
  rs.scr = (void*)(s_scr_all + (size_t)(threadIdx.x >> 6) * 64u);
  const __amdgpu_buffer_rsrc_t rk = BIG ? __builtin_amdgcn_make_buffer_rsrc((void*)ix.keys, 0, (int)((ix.slots + 2u) * 8u), 0x00020000) : rs.raw;
  const uint32_t keys_off = BIG ? 0u : ix.keys_off;
  const __amdgpu_buffer_rsrc_t rsn = __builtin_amdgcn_make_buffer_rsrc((void*)sn.blob, 0, (int)sn.blob_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void*)reqs, 0, (int)(n_reqs * stride), 0x00020000);
  const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc((void*)ix.lists, 0, use_lists ? (int)((ix.slots + 4u) * 64u) : 0, 0x00020000);
  uint32_t* s_hist = s_hist_all + (threadIdx.x >> 6) * sn.J * 16u;
  const uint32_t lane8 = (uint32_t)lane * 8u, lane4 = (uint32_t)lane * 4u, laneLW = (uint32_t)lane * (uint32_t)sizeof(LW);

  const LW valid = (LW)(valid_word<LW>(sn.n_pods, lane) & ((const LW*)sn.act_t)[lane]);   // existing AND active pods of this lane
  const LW qminw = (MASKED && sn.lead_queue) ? ((const LW*)sn.qmin_t)[lane] : (LW)0;
  const LW qmaxw = (MASKED && sn.lead_queue) ? ((const LW*)sn.qmax_t)[lane] : (LW)0;
  // MASKED list routes work on the request's mask row as it is (natural layout, lane w = word w): no transposition
  const uint64_t nat_act = (MASKED && SPARSE) ? sn.nat[lane] : 0ull;     // (the qmin / qmax words are re-read per request: 1 KiB, L1-resident)
  uint32_t w_hits = 0, w_lookups = 0;                                               // per wavefront and launch: < 2^32
  const uint32_t hwords = (stride - 8u) / 8u;                                       // hash words per request row
  const uint32_t hw0 = hwords < kKeysPerProbe ? hwords : kKeysPerProbe;             // ... probed by the pipelined first gather
  // u64 word of a request row that this lane pair reads as "its" hash (always in bounds: word 0 when the row has no hashes)
  const uint32_t ki = (uint32_t)lane >> 1;
  const uint32_t hidx8 = ((HAS_P && hw0) ? 1u + (ki < hw0 ? ki : hw0 - 1u) : 0u) * 8u;
  const bool use_index = HAS_P && ix.slots != 0u && hw0 != 0u;

  if (!WL && gwave >= n_reqs) return;

  // Stage 0: request row.  The header is wave-uniform: a scalar load straight into SGPRs (constant address space); the
  // lane pair's hash is one buffer load.
  auto issue_row = [&](uint32_t rr, uint32_t r_fallback, ReqRegs& q) {
    const uint32_t soff = (rr < n_reqs ? rr : r_fallback) * stride;
    q.hdr = *(const uint64_t __attribute__((address_space(4)))*)(reqs + soff);
    q.h = buffer_load_u64(rq, hidx8, soff);
  };
  // Stage 1: home buckets of the request's first 32 hashes
  auto prepare_keys = [&](ReqRegs& q) {
    if (use_index) {
      q.h = (ki < hw0) ? q.h : 0ull;
      pair_probe_prepare(ix, q);
    }
  };
  auto issue_keys = [&](ReqRegs& q) {
    if (use_index) pair_probe_issue(rk, keys_off, q, lane);
  };

  // ---- the stages of one request ------------------------------------------------------------------------------------
  struct ReqS {              // wave-uniform facts of a request in flight
    uint32_t r;              // request index
    int32_t adapter;
    uint32_t nb, m0, hits, arow;
    uint32_t badm;           // ~0 when the header is out of range (device entry points do not pre-validate rows), else 0: the row
                             // is then scored as an empty base-model request (nothing is indexed out of bounds) and the stores
                             // OR this mask into the pick (-> EPPK_NO_PICK) and clear the score with it
  };
  struct Tabs {              // per-adapter tables of a request: first 16 top-table entries (lanes 0..15), LoRA tier planes
    double top_t;
    uint32_t top_p;
    LW thi, tlo;
    uint64_t cn;             // MASKED list routes: the request's candidates, natural layout (lane w = pods 64w .. 64w+63)
  };

  // finish the probe of the request in `q` (its keys were gathered earlier): m0 leading hits, slot map
  auto stage_finish = [&](uint32_t r, const ReqRegs& q, ReqS& s, uint32_t& slot_eff) {
    s.r = r;
    s.adapter = (int32_t)(uint32_t)q.hdr;            // wave-uniform (scalar load)
    s.nb = (uint32_t)(q.hdr >> 32);
    s.badm = 0u;
    if (__builtin_expect(s.nb > hwords || s.adapter < -1 || s.adapter >= (int32_t)EPPK_MAX_ADAPTERS, 0)) {   // not scored + sticky flag
      s.badm = 0xFFFFFFFFu;
      s.adapter = -1; s.nb = 0u;
      if (lane == 0) __builtin_amdgcn_raw_buffer_store_b32(kStatusBadRow, rsn, (int)(sn.blob_bytes - kBlobStatusTail), 0, 0);
    }
    s.arow = (HAS_L && s.adapter >= 0) ? (uint32_t)s.adapter : 128u;
    s.m0 = 0;
    slot_eff = ix.slots + 2u;
    if (use_index && s.nb != 0u) s.m0 = pair_probe_finish(ix, q, s.nb < kKeysPerProbe ? s.nb : kKeysPerProbe, lane, slot_eff);
    s.hits = s.m0;
  };
  // top table of the adapter row: only its first 16 entries are fetched up front (lanes 0..15; the first entry outside M
  // is almost always among them), the other 48 on demand.  LoRA tier planes: only pods of M (and the rare full scan) need them.
  auto load_tiers = [&](const ReqS& s, Tabs& t) {
    t.thi = buffer_load_lw<LW>(rsn, laneLW, SnapOff<LW>::thi + s.arow * (uint32_t)(64u * sizeof(LW)));
    if (s.adapter >= 0) t.tlo = buffer_load_lw<LW>(rsn, laneLW, SnapOff<LW>::tlo + s.arow * (uint32_t)(64u * sizeof(LW)));   // base model: lo plane is all-zero
  };
  auto stage_tables = [&](const ReqS& s, Tabs& t) {
    t.top_t = -__builtin_inf();
    t.top_p = kNoPod;
    if (lane < 16) {
      t.top_t = __longlong_as_double((long long)buffer_load_u64(rsn, lane8, SnapOff<LW>::topv + s.arow * 512u));
      t.top_p = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsn, (int)lane4, (int)(SnapOff<LW>::topi + s.arow * 256u), 0);
    }
    t.thi = 0; t.tlo = 0;
    if (HAS_L && (MASKED || s.m0 > 0u)) load_tiers(s, t);
    t.cn = 0ull;
    if (MASKED && SPARSE && use_lists) {
      const uint64_t mw = ((uint32_t)lane < sn.J) ? cand_mask[(size_t)s.r * sn.J + (uint32_t)lane] : 0ull;   // one coalesced row load
      t.cn = mw & nat_act;
    }
  };
  // bit of pod `p` in a natural-layout set held one word per lane
  auto nat_bit = [&](uint64_t words, uint32_t p) -> bool {
    const uint32_t w = p >> 6;
    const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)words, (int)w), hi = (uint32_t)__shfl((int)(uint32_t)(words >> 32), (int)w);
    return ((((uint64_t)hi << 32) | lo) >> (p & 63u)) & 1ull;
  };
  // rows of the request, up to 16 in flight
  auto stage_rows = [&](const ReqS& s, uint32_t slot_eff, LW (&w)[16]) {
#ifdef EPPK_DBG_ONE_ROW     // timing experiment only (wrong results): one row load per request, no carry-save counting
    {
      LW t[1];
      load_rows<LW, 1, BIG>(rs, slot_eff, 0, lane, t);
#pragma unroll
      for (int u = 0; u < 16; ++u) w[u] = t[0];
      return;
    }
#endif
    if (s.m0 > 8u) {
      load_rows<LW, 16, BIG>(rs, slot_eff, 0, lane, w);
    } else {
      LW t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = 0;
      if (s.m0 > 0u) load_rows<LW, 8, BIG>(rs, slot_eff, 0, lane, t);
#pragma unroll
      for (int u = 0; u < 8; ++u) { w[u] = t[u]; w[8 + u] = 0; }
    }
  };
  // count the loaded rows into the bit-sliced counters (+ the rare continuations)
  auto stage_count = [&](ReqS& s, uint32_t slot_eff, const LW (&w)[16], LW (&c)[NPL]) {
#pragma unroll
    for (int k = 0; k < NPL; ++k) c[k] = 0;
    if (HAS_P) {
      const uint32_t m0 = s.m0, nb = s.nb;
#ifdef EPPK_DBG_ONE_ROW
      c[4] = w[0];
      if (stats) { w_hits += s.hits; w_lookups += (s.hits + 1u < nb) ? s.hits + 1u : nb; }
      return;
#endif
      if (m0 > 8u) {
        LW b[5];
        csa16<LW>(w, b);
#pragma unroll
        for (int k = 0; k < 5; ++k) c[k] = b[k];
      } else if (m0 > 0u) {
        LW t[8], b[4];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = w[u];
        csa8<LW>(t, b);
#pragma unroll
        for (int k = 0; k < 4; ++k) c[k] = b[k];
      }
      if (m0 > 16u) count_more<LW, NPL, BIG>(rs, slot_eff, 16u, m0, lane, c);
      if (__builtin_expect(m0 == kKeysPerProbe && nb > kKeysPerProbe, 0)) {                 // hashes beyond the first 32 (every earlier key hit): not pipelined
        uint32_t mlast = m0;
        for (uint32_t b0 = kKeysPerProbe; b0 < nb && mlast == kKeysPerProbe; b0 += kKeysPerProbe) {
          const uint32_t nchunk = (nb - b0) < kKeysPerProbe ? (nb - b0) : kKeysPerProbe;
          ReqRegs t;
          t.hdr = 0;
          t.h = buffer_load_u64(rq, (1u + b0 + (ki < nchunk ? ki : 0u)) * 8u, s.r * stride);
          t.h = (ki < nchunk) ? t.h : 0ull;
          pair_probe_prepare(ix, t);
          pair_probe_issue(rk, keys_off, t, lane);
          uint32_t slotc;
          mlast = pair_probe_finish(ix, t, nchunk, lane, slotc);
          count_more<LW, NPL, BIG>(rs, slotc, 0u, mlast, lane, c);
          s.hits += mlast;
        }
      }
      if (stats) { w_hits += s.hits; w_lookups += (s.hits + 1u < nb) ? s.hits + 1u : nb; }
    }
  };
  // total of pod p with cnt matched blocks and LoRA tier `tier` (2 bits: thi, tlo), binary64 adds in chain order
  auto pod_total = [&](uint32_t p, uint32_t cnt, uint32_t tier, const double* pt_row, double nbd) __attribute__((always_inline)) -> double {
    // pterm = clamp01(cnt / nb) * w_prefix: from the host-built exact table (NPL == 6), else one binary64 division
    double pterm;
    if constexpr (pterm_tab) pterm = pt_row[cnt];
    else pterm = clamp01((double)cnt / nbd) * tl.wp;
    double lterm = 0.0;
    if (HAS_L) lterm = s_lw[tier];
    double t;
    if constexpr (GEN) {               // interpreted tail: one binary64 add per scorer, in chain order
      t = s_base[p];
      for (uint32_t i = 0; i < tl.n_tail; ++i) {
        const uint32_t kd = tl.kind[i];
        t = t + (kd == 0u ? lterm : kd == 1u ? pterm : kd == 2u ? s_post0[p] : s_post1[p]);
      }
    } else {
      t = eval_total<HAS_L, HAS_P, P_FIRST>(s_base[p], lterm, pterm);
    }
    return t;
  };

  // ---- SPARSE: the request's pod sets as short lists -------------------------------------------------------------------
  // Lane (k = lane & 15, c = lane >> 4) reads chunk c of the list of hit k (la) and of hit 16 + k (lb): 8 ids each.
  auto issue_lists = [&](const ReqS& s, uint32_t slot_eff, u32x4_t& la, u32x4_t& lb) {
    const uint32_t k = (uint32_t)lane & 15u, cch = (uint32_t)lane >> 4;
    const uint32_t sa = (uint32_t)__shfl((int)slot_eff, (int)(2u * (k < s.m0 ? k : 0u)));   // (lanes beyond the hits re-read hit 0: stage_uniform)
#ifdef EPPK_DBG_NO_LISTS    // timing experiment only (wrong results): no list loads
    la = (u32x4_t)(sa & 1u); lb = (u32x4_t)(0xFFFFFFFFu); return;
#endif
    la = __builtin_amdgcn_raw_buffer_load_b128(rl, (int)(sa * 64u + cch * 16u), 0, 0);
    lb = (u32x4_t)(0xFFFFFFFFu);
    if (s.m0 > 16u) {
      const uint32_t sb = (uint32_t)__shfl((int)slot_eff, (int)(2u * (16u + k)));
      lb = __builtin_amdgcn_raw_buffer_load_b128(rl, (int)(sb * 64u + cch * 16u), 0, 0);
    }
  };
  // true iff one of the request's hits has an overflowed list (the dense rows must be used)
  auto lists_overflowed = [&](const ReqS& s, const u32x4_t& la, const u32x4_t& lb) -> bool {
    const uint32_t k = (uint32_t)lane & 15u, cch = (uint32_t)lane >> 4;
    const bool o = cch == 0u && ((k < s.m0 && la.w > kListCap) || (16u + k < s.m0 && lb.w > kListCap));
    return __any(o);
  };
  // The common shape of a request: all its hits list the SAME pods (the blocks of one shared prefix are cached together), so
  // matched = m0 for every listed pod.  The 16 lanes of a DPP row hold the same chunk of the 16 hits: one row_shr:1 compare
  // per dword detects it; lane (k, c) then evaluates id k of chunk c -- all pods of the list in ONE step, no histogram.
  // Returns false (nothing stored) when the lists differ or the top table's first 16 entries are all listed.
  auto stage_uniform = [&](const ReqS& s, const u32x4_t& la, const u32x4_t& lb, Tabs& tb) -> bool {
    const uint32_t r = s.r, nb = s.nb, m0 = s.m0;
    const uint32_t k = (uint32_t)lane & 15u;
    auto shr1 = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x111, 0xf, 0xf, false); };   // lane k-1 of the row (k = 0: itself)
    uint32_t diff = 0;
    if (k < m0) diff = (la.x ^ shr1(la.x)) | (la.y ^ shr1(la.y)) | (la.z ^ shr1(la.z)) | (la.w ^ shr1(la.w));
    if (16u + k < m0) diff |= (lb.x ^ la.x) | (lb.y ^ la.y) | (lb.z ^ la.z) | (lb.w ^ la.w);
    if (__any(diff != 0u)) return false;
#ifdef EPPK_DBG_SKIP_UNIFORM_EVAL   // timing experiment only (wrong results): no evaluation, no argmax, no table walk
    if (lane == 0) { out_pick[r] = (int32_t)(la.x & 0xFFFu); if (out_score) out_score[r] = 0.0; }
    return true;
#endif
    // id k of this lane's chunk (positions 6..7 of chunk 0 are the count, of the other chunks unused)
    const uint32_t dw = (k & 4u) ? ((k & 2u) ? la.w : la.z) : ((k & 2u) ? la.y : la.x);
    uint32_t id = (k & 1u) ? (dw >> 16) : (dw & 0xFFFFu);
    if (k >= 6u) id = kListNone;
    const bool listed = id < sn.n_pods;
    const uint32_t p = listed ? id : 0u;
    bool v = listed;
    if (MASKED) v = nat_bit(tb.cn, p) && listed;       // a listed pod outside the request's candidates is not evaluated
    uint32_t tier = 0;
    if (HAS_L) {
      const uint32_t src = p & 63u, jb = p >> 6;
      LW th, tl_;
      if constexpr (sizeof(LW) == 8) {
        th = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(tb.thi >> 32), (int)src) << 32) | (uint32_t)__shfl((int)(uint32_t)tb.thi, (int)src);
        tl_ = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(tb.tlo >> 32), (int)src) << 32) | (uint32_t)__shfl((int)(uint32_t)tb.tlo, (int)src);
      } else {
        th = (LW)__shfl((int)(uint32_t)tb.thi, (int)src);
        tl_ = (LW)__shfl((int)(uint32_t)tb.tlo, (int)src);
      }
      tier = (uint32_t)(((th >> jb) & 1) << 1) | (uint32_t)((tl_ >> jb) & 1);
    }
    const double t = pod_total(p, v ? m0 : 0u, tier, s_pterm + (size_t)nb * sn.pterm_ld, (double)nb);
    if constexpr (TOPK) {
      // Ordered fallbacks from the lists: a k-way merge of the listed pods (totals in the lanes; the best remaining one by a DPP
      // argmax per round) with the adapter's top table (already ordered; the cursor skips listed pods).  A table that runs out
      // within its first 16 entries sends the request to the dense rows (return false: every round is rewritten there).
      const uint32_t tk = topk;
      unsigned long long tcm = ~0ull;
      if (MASKED) tcm = __ballot(nat_bit(tb.cn, tb.top_p != kNoPod ? tb.top_p : 0u) && tb.top_p != kNoPod);
      uint32_t e = 0;
      bool table_end = false;
      for (uint32_t round = 0; round < tk; ++round) {
        double best = v ? t : -__builtin_inf();
        uint32_t bidx = v ? p : kNoPod;
        wave_argmax_dpp(best, bidx);
        double cand_t = -__builtin_inf();
        uint32_t cand_p = kNoPod;
        if (!table_end) {
          for (; e < 16u; ++e) {
            const uint32_t tp = (uint32_t)__builtin_amdgcn_readlane((int)tb.top_p, (int)e);
            if (tp == kNoPod) { table_end = true; break; }
            if (MASKED && !((tcm >> e) & 1ull)) continue;
            if (!__any(listed && id == tp)) {
              cand_t = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(tb.top_t), (int)e),
                                        __builtin_amdgcn_readlane(__double2loint(tb.top_t), (int)e));
              cand_p = tp;
              break;
            }
          }
          if (e == 16u && !table_end) {
            if (sn.n_pods > 16u) return false;          // entries 16..63 would be needed
            table_end = true;
          }
        }
        const bool take_table = cand_t > best || (cand_t == best && cand_p < bidx);
        if (take_table) { best = cand_t; bidx = cand_p; ++e; }
        else if (v && p == bidx) v = false;             // the listed winner leaves the pool
        const bool none = bidx == kNoPod || s.badm != 0u;
        if (lane == 0) {
          out_pick[(size_t)r * tk + round] = (none ? -1 : (int32_t)bidx) | (int32_t)s.badm;
          if (out_score) out_score[(size_t)r * tk + round] = __longlong_as_double(__double_as_longlong(none ? 0.0 : best) & ~(long long)(int32_t)s.badm);
        }
        if (none) {
          if (lane == 0)
            for (uint32_t i = round + 1u; i < tk; ++i) { out_pick[(size_t)r * tk + i] = -1; if (out_score) out_score[(size_t)r * tk + i] = 0.0; }
          break;
        }
      }
      return true;
    }
    double best = v ? t : -__builtin_inf();
    uint32_t bidx = v ? p : kNoPod;
    wave_argmax_dpp(best, bidx);
    // best pod outside M: the first top-table entry that is not listed
    double cand_t = -__builtin_inf();
    uint32_t cand_p = kNoPod;
    const double top0 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(tb.top_t), 0),
                                         __builtin_amdgcn_readlane(__double2loint(tb.top_t), 0));
    if (!(best > top0)) {
      unsigned long long tcm = ~0ull;                                     // MASKED: table entries (lanes 0..15) that are candidates
      if (MASKED) tcm = __ballot(nat_bit(tb.cn, tb.top_p != kNoPod ? tb.top_p : 0u) && tb.top_p != kNoPod);
      uint32_t e = 0;
      for (; e < 16u; ++e) {
        const uint32_t tp = (uint32_t)__builtin_amdgcn_readlane((int)tb.top_p, (int)e);
        if (tp == kNoPod) break;                                          // fewer than 16 pods: the table ends here
        if (MASKED && !((tcm >> e) & 1ull)) continue;                     // not a candidate of this request
        if (!__any(listed && id == tp)) {
          cand_t = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(tb.top_t), (int)e),
                                    __builtin_amdgcn_readlane(__double2loint(tb.top_t), (int)e));
          cand_p = tp;
          break;
        }
      }
      if (__builtin_expect(e == 16u && sn.n_pods > 16u, 0)) return false;  // all listed: the general path fetches the rest of the table
    }
    if (cand_t > best || (cand_t == best && cand_p < bidx)) { best = cand_t; bidx = cand_p; }
    const bool none = bidx == kNoPod;
    if (lane == 0) {
      out_pick[r] = (none ? -1 : (int32_t)bidx) | (int32_t)s.badm;
      if (out_score) out_score[r] = __longlong_as_double(__double_as_longlong(none ? 0.0 : best) & ~(long long)(int32_t)s.badm);
    }
    return true;
  };
  auto stage_sparse = [&](const ReqS& s, u32x4_t la, u32x4_t lb, Tabs& tb) {
    const uint32_t r = s.r, nb = s.nb, m0 = s.m0, arow = s.arow;
    const uint32_t k = (uint32_t)lane & 15u, cch = (uint32_t)lane >> 4;
    if (cch == 0u) { la.w = 0xFFFFFFFFu; lb.w = 0xFFFFFFFFu; }            // (the count)
    if (k >= m0) la = (u32x4_t)(0xFFFFFFFFu);
    if (16u + k >= m0) lb = (u32x4_t)(0xFFFFFFFFu);
    const uint32_t d[8] = {la.x, la.y, la.z, la.w, lb.x, lb.y, lb.z, lb.w};
    // pass 1: matched[p] += 1 per listed pod (byte counters: <= 32 hits); the lane that finds the byte at zero owns the pod.
    // Ids fill every chunk front to back, so the first step without a valid id ends a half.
    uint32_t first = 0;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const uint32_t id = (d[half * 4 + (i >> 1)] >> (16 * (i & 1))) & 0xFFFFu;
        if (!__any(id != kListNone)) break;
        const bool v = id < sn.n_pods;                                    // (a pod beyond the published snapshot is no candidate)
        if (v) {
          const uint32_t sh = (id & 3u) * 8u;
          const uint32_t old = atomicAdd(&s_hist[id >> 2], 1u << sh);
          if (((old >> sh) & 0xFFu) == 0u) first |= 1u << (half * 8 + i);
        }
      }
    }
    // pass 2: the owners evaluate their pods in full (every lane takes part in the tier look-up: bpermute reads active lanes only)
    double best = -__builtin_inf();
    uint32_t bidx = kNoPod;
    const double* pt_row = s_pterm + (size_t)nb * sn.pterm_ld;
    const double nbd = (double)nb;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (!__any((first >> (half * 8 + i)) != 0u)) break;              // no owner at this or any later step
        const bool f = (first >> (half * 8 + i)) & 1u;
        const uint32_t id = (d[half * 4 + (i >> 1)] >> (16 * (i & 1))) & 0xFFFFu;
        const uint32_t p = f ? id : 0u;
        const uint32_t cnt = (s_hist[p >> 2] >> ((p & 3u) * 8u)) & 0xFFu;
        uint32_t tier = 0;
        if (HAS_L) {
          const uint32_t src = p & 63u, jb = p >> 6;
          LW th, tl_;
          if constexpr (sizeof(LW) == 8) {
            th = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(tb.thi >> 32), (int)src) << 32) | (uint32_t)__shfl((int)(uint32_t)tb.thi, (int)src);
            tl_ = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(tb.tlo >> 32), (int)src) << 32) | (uint32_t)__shfl((int)(uint32_t)tb.tlo, (int)src);
          } else {
            th = (LW)__shfl((int)(uint32_t)tb.thi, (int)src);
            tl_ = (LW)__shfl((int)(uint32_t)tb.tlo, (int)src);
          }
          tier = (uint32_t)(((th >> jb) & 1) << 1) | (uint32_t)((tl_ >> jb) & 1);
        }
        const double t = pod_total(p, f ? cnt : 0u, tier, pt_row, nbd);
        bool fc = f;
        if (MASKED) fc = nat_bit(tb.cn, p) && f;                          // (every lane takes part in the shuffle)
        if (fc && (t > best || (t == best && p < bidx))) { best = t; bidx = p; }
      }
    }
    wave_argmax_dpp(best, bidx);
    // best pod outside M: first entry of the adapter's top table whose counter byte is zero
    double cand_t = -__builtin_inf();
    uint32_t cand_p = kNoPod;
    const double top0 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(tb.top_t), 0),
                                         __builtin_amdgcn_readlane(__double2loint(tb.top_t), 0));
    if (!(best > top0)) {
      auto entry_ok = [&](uint32_t tp) -> unsigned long long {
        const bool has = tp != kNoPod;
        const uint32_t q = has ? tp : 0u;
        bool ok = has && ((s_hist[q >> 2] >> ((q & 3u) * 8u)) & 0xFFu) == 0u;
        if (MASKED) ok = nat_bit(tb.cn, q) && ok;
        return __ballot(ok);
      };
      unsigned long long okm = entry_ok(tb.top_p);
      if (__builtin_expect(okm == 0ull && sn.n_pods > 16u, 0)) {         // none of the first 16: fetch entries 16..63
        if (lane >= 16) {
          tb.top_t = __longlong_as_double((long long)buffer_load_u64(rsn, lane8, SnapOff<LW>::topv + arow * 512u));
          tb.top_p = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsn, (int)lane4, (int)(SnapOff<LW>::topi + arow * 256u), 0);
        }
        okm = entry_ok(tb.top_p);
      }
      if (okm) {
        const int f = __builtin_ctzll(okm);
        cand_t = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(tb.top_t), f),
                                  __builtin_amdgcn_readlane(__double2loint(tb.top_t), f));
        cand_p = (uint32_t)__builtin_amdgcn_readlane((int)tb.top_p, f);
      } else if (__builtin_expect(sn.n_pods > 64u, 0)) {
        // rare: all 64 table entries are in M -> T_a over every pod outside M (total == T_a there)
        double rbest = -__builtin_inf();
        uint32_t ridx = kNoPod;
        for (uint32_t j = 0; j < sn.J; ++j) {
          const uint32_t p = j * 64u + (uint32_t)lane;
          double t = s_base[p];
          if constexpr (GEN) {
            const double lterm = HAS_L ? s_lw[(uint32_t)(((tb.thi >> j) & 1) << 1) | (uint32_t)((tb.tlo >> j) & 1)] : 0.0;
            for (uint32_t i = 0; i < tl.n_tail; ++i) {
              const uint32_t kd = tl.kind[i];
              if (kd != 1u) t = t + (kd == 0u ? lterm : kd == 2u ? s_post0[p] : s_post1[p]);
            }
          } else {
            if (HAS_L) t = t + s_lw[(uint32_t)(((tb.thi >> j) & 1) << 1) | (uint32_t)((tb.tlo >> j) & 1)];
          }
          bool okp = p < sn.n_pods && ((s_hist[p >> 2] >> ((p & 3u) * 8u)) & 0xFFu) == 0u;
          if (MASKED) {                      // pod j*64 + lane is bit `lane` of natural word j
            const uint64_t cw = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(tb.cn >> 32), (int)j) << 32) |
                                (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)tb.cn, (int)j);
            okp = okp && ((cw >> lane) & 1ull);
          } else {
            okp = okp && ((valid >> j) & 1);   // (holes of the snapshot)
          }
          if (okp && t > rbest) { rbest = t; ridx = p; }
        }
        wave_argmax_dpp(rbest, ridx);
        cand_t = rbest;
        cand_p = ridx;
      }
    }
    if (cand_t > best || (cand_t == best && cand_p < bidx)) { best = cand_t; bidx = cand_p; }
    // pass 3: the touched counters back to zero
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const uint32_t id = (d[half * 4 + (i >> 1)] >> (16 * (i & 1))) & 0xFFFFu;
        if (!__any(id != kListNone)) break;
        if (id < sn.n_pods) s_hist[id >> 2] = 0u;
      }
    }
    const bool none = bidx == kNoPod;
    if (lane == 0) {
      out_pick[r] = (none ? -1 : (int32_t)bidx) | (int32_t)s.badm;
      if (out_score) out_score[r] = __longlong_as_double(__double_as_longlong(none ? 0.0 : best) & ~(long long)(int32_t)s.badm);
    }
  };

  // MASKED single picks whose candidates miss a snapshot-wide QUEUE extreme (base[] and the top tables do not apply: the request's own
  // normalisers), from the lists: matched[] into the byte histogram as stage_sparse does, the candidate row transposed in registers,
  // every candidate evaluated in full (masked_exact_hist), histogram cleared.  The dense route (16 rows built from the lists, the
  // carry-save tree, masked_exact's one-candidate trips) cost such a request 30-odd us at 1/8 density -- and it ends its workgroup.
  auto stage_exact_lists = [&](const ReqS& s, u32x4_t la, u32x4_t lb, Tabs& tb) {
    const uint32_t r = s.r, m0 = s.m0;
    const uint32_t k = (uint32_t)lane & 15u, cch = (uint32_t)lane >> 4;
    if (cch == 0u) { la.w = 0xFFFFFFFFu; lb.w = 0xFFFFFFFFu; }            // (the count)
    if (k >= m0) la = (u32x4_t)(0xFFFFFFFFu);
    if (16u + k >= m0) lb = (u32x4_t)(0xFFFFFFFFu);
    const uint32_t d[8] = {la.x, la.y, la.z, la.w, lb.x, lb.y, lb.z, lb.w};
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const uint32_t id = (d[half * 4 + (i >> 1)] >> (16 * (i & 1))) & 0xFFFFu;
        if (!__any(id != kListNone)) break;
        if (id < sn.n_pods) atomicAdd(&s_hist[id >> 2], 1u << ((id & 3u) * 8u));
      }
    }
    const LW cand = (LW)(valid & transpose_words<LW>(tb.cn, sn.J, lane));
    double best = -__builtin_inf();
    uint32_t bidx = kNoPod;
    masked_exact_hist<LW>(sn, ch, cand, tb.thi, tb.tlo, s.nb, lane, s_hist, best, bidx);
    wave_argmax_dpp(best, bidx);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const uint32_t id = (d[half * 4 + (i >> 1)] >> (16 * (i & 1))) & 0xFFFFu;
        if (!__any(id != kListNone)) break;
        if (id < sn.n_pods) s_hist[id >> 2] = 0u;
      }
    }
    const bool none = bidx == kNoPod;
    if (lane == 0) {
      out_pick[r] = (none ? -1 : (int32_t)bidx) | (int32_t)s.badm;
      if (out_score) out_score[r] = __longlong_as_double(__double_as_longlong(none ? 0.0 : best) & ~(long long)(int32_t)s.badm);
    }
  };

  // evaluate, select, store
  auto stage_eval = [&](const ReqS& s, const LW (&c)[NPL], Tabs& tb) {
    const uint32_t r = s.r, nb = s.nb, hits = s.hits, arow = s.arow, m0 = s.m0;
    const uint32_t tk = TOPK ? topk : 1u;      // entries per request in out_pick / out_score
    const int32_t adapter = s.adapter;
    double& top_t = tb.top_t;
    uint32_t& top_p = tb.top_p;
    LW& thi = tb.thi;
    LW& tlo = tb.tlo;
    (void)adapter; (void)arow; (void)m0;
#ifdef EPPK_DBG_SKIP_EVAL     // timing experiment only (wrong results): no evaluation phase
    if (lane == 0) { out_pick[r] = (int32_t)(uint32_t)c[0] + (int32_t)hits; }
    return;
#endif
    LW nz = 0;  // M: pods with matched > 0
    if (HAS_P && hits) {
#pragma unroll
      for (int k = 0; k < NPL; ++k) nz |= c[k];
      nz &= valid;
    }
    LW cand = valid;   // Filter: the request's candidate subset (request.go:104-133 as a bitmask), lane-transposed
    if (MASKED) cand &= transpose_mask<LW>(cand_mask + (size_t)r * sn.J, sn.J, lane);

    // Masked requests: base[] and the top tables embed the snapshot-wide QUEUE normalisers; they apply iff the
    // candidates contain a pod at the global minimum and one at the global maximum queue depth.
    bool exact = false;
    if (MASKED && sn.lead_queue) exact = !(__any((cand & qminw) != 0) && __any((cand & qmaxw) != 0));

    // One selection round per requested output: round 0 is the pick; rounds 1..topk-1 (ordered fallbacks, eppk_pick_topk)
    // repeat the selection over the candidates not yet reported (`excl`, lane-transposed).  topk == 1: one trip, excl == 0.
    auto select_round = [&](const uint32_t round, LW& excl) __attribute__((always_inline)) -> bool {
    double best = -__builtin_inf();
    uint32_t bidx = kNoPod;
    double cand_t = -__builtin_inf();
    uint32_t cand_p = kNoPod;
    if (MASKED && exact) {
      const LW cand_e = (LW)(cand & (LW)~excl);
      if (__any(cand_e != 0)) {
        masked_exact<LW, NPL>(sn, ch, cand, cand_e, c, thi, tlo, nb, lane, best, bidx);
        wave_argmax_dpp(best, bidx);
      }
    } else {
      const LW mset = (LW)((MASKED ? (LW)(nz & cand) : nz) & (LW)~excl);   // candidates with a prefix match: evaluated in full
      const LW okset = (LW)(cand & (LW)~nz & (LW)~excl);                    // candidates whose total is exactly T_a[p]
      const bool any_m = HAS_P && __any(mset != 0);

      if (any_m) {
        const double* pt_row = s_pterm + (size_t)nb * sn.pterm_ld;   // cnt > 0 implies nb > 0
        const double nbd = (double)nb;
        LW rem = mset;
        while (rem != 0) {                   // each lane walks its own pods of M in ascending order (branch-free body)
          const uint32_t j = ctz_lw<LW>(rem);
          rem = (LW)(rem & (LW)(rem - 1));
          const uint32_t p = j * 64u + (uint32_t)lane;
          uint32_t cnt = 0;
#pragma unroll
          for (int k = 0; k < NPL; ++k) cnt |= (uint32_t)((c[k] >> j) & 1) << k;
          const uint32_t tier = HAS_L ? ((uint32_t)(((thi >> j) & 1) << 1) | (uint32_t)((tlo >> j) & 1)) : 0u;
          const double t = pod_total(p, cnt, tier, pt_row, nbd);
          if (t > best) { best = t; bidx = p; }
        }
        wave_argmax_dpp(best, bidx);
      }

      // Best candidate outside M: the first entry of the adapter's top table that is in okset.  No pod outside M can beat
      // the first entry's T, so the look-up is skipped when the best pod of M already beats it.
      const double top0 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(top_t), 0),
                                           __builtin_amdgcn_readlane(__double2loint(top_t), 0));
      if (!(round == 0u && any_m && best > top0)) {    // (a fallback round cannot use the bound: the table's head may be taken)
        // lanes whose table entry is a candidate outside M
        auto entry_ok = [&](uint32_t tp) -> unsigned long long {
          const bool has = tp != kNoPod;
          bool ok = has;
          if (MASKED || (HAS_P && hits) || round != 0u) {
            const uint32_t ql = has ? (tp & 63u) : 0u, qj = has ? (tp >> 6) : 0u;
            LW okq;
            if constexpr (sizeof(LW) == 8) {
              const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)okset, (int)ql);
              const uint32_t hi = (uint32_t)__shfl((int)(uint32_t)(okset >> 32), (int)ql);
              okq = ((uint64_t)hi << 32) | lo;
            } else {
              okq = (LW)__shfl((int)(uint32_t)okset, (int)ql);
            }
            ok = has && ((okq >> qj) & 1);
          }
          return __ballot(ok);
        };
        unsigned long long okm = entry_ok(top_p);
        if (__builtin_expect(okm == 0ull && sn.n_pods > 16u, 0)) {     // none of the first 16: fetch entries 16..63
          if (lane >= 16) {
            top_t = __longlong_as_double((long long)buffer_load_u64(rsn, lane8, SnapOff<LW>::topv + arow * 512u));
            top_p = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsn, (int)lane4, (int)(SnapOff<LW>::topi + arow * 256u), 0);
          }
          okm = entry_ok(top_p);
        }
        if (okm) {
          const int f = __builtin_ctzll(okm);
          cand_t = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(top_t), f),
                                    __builtin_amdgcn_readlane(__double2loint(top_t), f));
          cand_p = (uint32_t)__builtin_amdgcn_readlane((int)top_p, f);
        } else if (__builtin_expect(sn.n_pods > 64u && __any(okset != 0), 0)) {
          // rare: table exhausted although eligible pods remain -> T_a over every eligible pod outside M (total == T_a there)
          if (HAS_L && !(MASKED || m0 > 0u)) load_tiers(s, tb);   // (not loaded up front: the request had no prefix hit)
          double rbest = -__builtin_inf();
          uint32_t ridx = kNoPod;
          for (uint32_t j = 0; j < sn.J; ++j) {
            const uint32_t p = j * 64u + (uint32_t)lane;
            double t = s_base[p];
            if constexpr (GEN) {             // T_a[p]: the chain without its PREFIX entry (adding +-0.0 is the identity)
              const double lterm = HAS_L ? s_lw[(uint32_t)(((thi >> j) & 1) << 1) | (uint32_t)((tlo >> j) & 1)] : 0.0;
              for (uint32_t i = 0; i < tl.n_tail; ++i) {
                const uint32_t kd = tl.kind[i];
                if (kd != 1u) t = t + (kd == 0u ? lterm : kd == 2u ? s_post0[p] : s_post1[p]);
              }
            } else {
              if (HAS_L) t = t + s_lw[(uint32_t)(((thi >> j) & 1) << 1) | (uint32_t)((tlo >> j) & 1)];
            }
            const bool okp = (okset >> j) & 1;
            if (okp && t > rbest) { rbest = t; ridx = p; }
          }
          wave_argmax_dpp(rbest, ridx);
          cand_t = rbest;
          cand_p = ridx;
        }
      }
    }
    if (cand_t > best || (cand_t == best && cand_p < bidx)) { best = cand_t; bidx = cand_p; }
    const bool none = bidx == kNoPod || (TOPK && s.badm != 0u);       // (fallback lists of an out-of-range row: all EPPK_NO_PICK)
    if (lane == 0) {
      out_pick[(size_t)r * tk + round] = (none ? -1 : (int32_t)bidx) | (int32_t)s.badm;
      if (out_score) out_score[(size_t)r * tk + round] = __longlong_as_double(__double_as_longlong(none ? 0.0 : best) & ~(long long)(int32_t)s.badm);
    }
    if (!none && (uint32_t)lane == (bidx & 63u)) excl |= (LW)((LW)1 << (bidx >> 6));
    return none;
    };
    if constexpr (!TOPK) {                   // the pick: one round, nothing excluded (everything about `round` / `excl` folds away)
      LW none_excluded = 0;
      select_round(0u, none_excluded);
    } else {
      LW excl = 0;
      for (uint32_t round = 0; round < tk; ++round) {
        if (select_round(round, excl)) {     // candidates exhausted: pad the rest of the list
          if (lane == 0)
            for (uint32_t i = round + 1u; i < tk; ++i) { out_pick[(size_t)r * tk + i] = -1; if (out_score) out_score[(size_t)r * tk + i] = 0.0; }
          break;
        }
      }
    }
  };

  uint32_t pf_sink = 0, pf_prev = 0;   // landing registers of the row prefetches (kept alive by the asm at the end, never read)
  // Stage 2 of request r (its row in `cur`, keys gathered); issues stage 1 of r + 1 (row in `nxt`) and stage 0 of
  // r + 2 (into `cur`, which is free once the probe of r is finished).
  auto process = [&](uint32_t r, uint32_t r_next2, ReqRegs& cur, ReqRegs& nxt) {
    ReqS s;
    uint32_t slot0;
    stage_finish(r, cur, s, slot0);
    prepare_keys(nxt);     // the hash of r + 1 is consumed here (its home bucket): the wait for its prefetch sits at the top
    Tabs tb;
    stage_tables(s, tb);
    // SPARSE: hits beyond the first 32 (chunks) and hits with an overflowed list take the dense rows
    // MASKED: requests without a hit go through the lists too (an empty list): the dense route would transpose the mask row
    bool sp = SPARSE && use_lists && (MASKED || s.m0 > 0u) && !(s.m0 == kKeysPerProbe && s.nb > kKeysPerProbe);
    u32x4_t la = (u32x4_t)(0xFFFFFFFFu), lb = (u32x4_t)(0xFFFFFFFFu);
    LW w[16];
    if (sp) issue_lists(s, slot0, la, lb);
    else stage_rows(s, slot0, w);
    issue_keys(nxt);
    issue_row(r_next2, r, cur);
#if EPPK_ROW_PREFETCH > 0
    if constexpr (!WL) {   // L2 prefetch of a later row: five lanes touch its (at most five) 64-byte sectors; the value is never used
      const uint32_t rp = r + (uint32_t)EPPK_ROW_PREFETCH * nwaves;
      if (rp < n_reqs) {
        const uint32_t off = (uint32_t)lane * 64u;
        pf_sink ^= pf_prev;                 // (consumes the PREVIOUS iteration's prefetch, long landed: keeps every load alive without a wait)
        pf_prev = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rq, (int)(off < stride - 4u ? off : stride - 4u), (int)(rp * stride), 0);
      }
    }
#endif
    __builtin_amdgcn_sched_barrier(0);
    bool exact_lists = false;
    if (MASKED && SPARSE && sp) {
      // no candidate at all: fail closed right here; candidates that miss the snapshot-wide QUEUE extremes need the request's own
      // normalisers: the exact evaluation of the dense route
      if (!__any(tb.cn != 0ull)) {
        const uint32_t tk = TOPK ? topk : 1u;
        if (lane == 0)
          for (uint32_t i = 0; i < tk; ++i) { out_pick[(size_t)r * tk + i] = -1; if (out_score) out_score[(size_t)r * tk + i] = 0.0; }
        return;
      }
      if (sn.lead_queue && !(__any((tb.cn & sn.nat[64 + lane]) != 0ull) && __any((tb.cn & sn.nat[128 + lane]) != 0ull))) {
        if constexpr (!TOPK) {
          exact_lists = true;                      // (single picks: every candidate in full, matched[] from the lists)
        } else {
          sp = false;
          stage_rows(s, slot0, w);
        }
      }
    }
    if (SPARSE && sp && __builtin_expect(lists_overflowed(s, la, lb), 0)) {
      sp = false;
      stage_rows(s, slot0, w);
    }
    bool done = false;
    if (SPARSE && sp) {
      if (MASKED && !TOPK && exact_lists) {
        stage_exact_lists(s, la, lb, tb);
        done = true;
      } else {
#ifndef EPPK_DBG_NO_UNIFORM
      done = stage_uniform(s, la, lb, tb);
#endif
      if constexpr (!TOPK) {
        if (!done) { stage_sparse(s, la, lb, tb); done = true; }
      } else {
        if (!done) stage_rows(s, slot0, w);        // fallback lists from differing pod lists: the dense rows after all
      }
      }
      if (done && stats) { w_hits += s.hits; w_lookups += (s.hits + 1u < s.nb) ? s.hits + 1u : s.nb; }
    }
    if (!done) {
      LW c[NPL];
      stage_count(s, slot0, w, c);
      stage_eval(s, c, tb);
    }
  };

  // ---- prologue: rows of the first two requests, keys of the first
  ReqRegs qa, qb;
  qa.kw[0] = qa.kw[1] = make_uint4(0, 0, 0, 0);
  qb.kw[0] = qb.kw[1] = make_uint4(0, 0, 0, 0);
  qa.bkt = qb.bkt = 0;
  if constexpr (WL) {
    // the segments the quad kernel's wavefronts left behind: seg = gwave, gwave + nwaves, ...; the same pipeline inside a segment
    for (uint32_t seg = gwave; seg < wk.n_segs; seg += nwaves) {
      const uint32_t len = wl_word(&wk.cnt[seg]);
      if (len == 0u) continue;
      const uint32_t* sl = wk.list + (size_t)seg * wk.cap;
      auto req_at = [&](uint32_t j) -> uint32_t { return (uint32_t)__builtin_amdgcn_readfirstlane((int)wl_word(&sl[j < len ? j : len - 1u])); };   // (past the end: the last row again, never used)
      uint32_t r0 = req_at(0u), r1 = req_at(1u);
      issue_row(r0, r0, qa);
      issue_row(r1, r1, qb);
      prepare_keys(qa); issue_keys(qa);
      for (uint32_t j = 0; j < len; j += 2u) {
        const uint32_t r2 = req_at(j + 2u);
        process(r0, r2, qa, qb);
        if (j + 1u >= len) break;
        const uint32_t r3 = req_at(j + 3u);
        process(r1, r3, qb, qa);
        r0 = r2; r1 = r3;
      }
    }
  } else {
    issue_row(gwave, gwave, qa);
    issue_row(gwave + nwaves, gwave, qb);
    prepare_keys(qa); issue_keys(qa);
  }
  // ---- steady state, unrolled twice: the stage registers swap roles instead of being copied
  if constexpr (!WL) {
    for (uint32_t r = gwave; r < n_reqs; r += 2u * nwaves) {
      process(r, r + 2u * nwaves, qa, qb);
      if (r + nwaves >= n_reqs) break;
      process(r + nwaves, r + 3u * nwaves, qb, qa);
    }
  }

#if EPPK_ROW_PREFETCH > 0
  asm volatile("" ::"v"(pf_sink ^ pf_prev));     // (the prefetch loads must not be dead-code eliminated)
#endif
  // probe statistics: one private slot per wavefront (plain read-modify-write; same-address atomics
  // from ~10^4 waves serialise at ~12 ns each and would add >100 us of tail to the launch)
  if (HAS_P && stats && lane == 0 && (w_hits | w_lookups) && stat_base + gwave < kStatSlots) {
    stats[4 + 2 * (stat_base + gwave)] += w_hits;
    stats[5 + 2 * (stat_base + gwave)] += w_lookups;
  }
}

template <typename LW, int NPL, bool HAS_L, bool HAS_P, bool P_FIRST, bool MASKED, bool BIG, bool GEN, bool TOPK, bool WL = false>
__global__ __launch_bounds__(EPPK_FAST_MAX_THREADS, EPPK_MIN_WAVES) void pick_fast_kernel(KSnap sn, KIndex ix, KTail tl, const uint8_t* __restrict__ reqs,
                                                        uint32_t stride, uint32_t n_reqs, uint32_t pwn,
                                                        const uint64_t* __restrict__ cand_mask, KChain ch,
                                                        int32_t* __restrict__ out_pick, double* __restrict__ out_score,
                                                        unsigned long long* __restrict__ stats, uint32_t topk, KWork wk) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  pick_fast_body<LW, NPL, HAS_L, HAS_P, P_FIRST, MASKED, BIG, GEN, TOPK, WL>(blockIdx.x, gridDim.x, 0u, false, smem, sn, ix, tl, reqs, stride, n_reqs, pwn, cand_mask, ch,
                                                                              out_pick, out_score, stats, topk, wk);
}



// ---- QUAD pick kernel: FOUR requests per wavefront, every gather laid out for the vector memory pipe ------------------------
// pick_fast_kernel spends a whole wavefront on one request (~180 vector + ~145 scalar instructions per decision).  The common
// shape of a request -- no candidate mask, one pick, at most 32 blocks probed, every hit's pod set still in its short list, all
// those lists IDENTICAL (the blocks of a shared prefix are cached together), the best pod without a prefix match among the first
// 16 entries of the adapter's top table -- needs only 16 lanes: this kernel gives each DPP ROW (16 lanes) its own request, so one
// instruction stream serves four requests.
//
// What bounds a pick on an L2-resident index is not instruction issue, though, but the vector memory pipe (measured, round 2:
// scripts/micro/l2gather.hip, profiles/r02_g_*): a CU looks up about one (lane, 16 bytes) access per clock and the L2 delivers
// about one 64-byte line per two clocks and CU -- whatever the instruction count.  A first version of this kernel (one lane per
// key, the whole 64-byte bucket by four 16-byte loads per lane) executed 2.5 x fewer vector instructions than pick_fast_kernel
// and took exactly as long: 19 M accesses per 64k batch, the texture-address unit 80 % busy.  Hence the layout below: the four
// lanes of a QUAD read the four 16-byte pieces of ONE line, so a line costs one access instead of four.
//   lane = (g, q, j): g = DPP row = request 4b + g of block b;  step i (0..7) of the probe serves key 4i + q of that request;
//   j = which 16 bytes of the key's 64-byte home bucket (words 2j, 2j+1) / of a hit's 64-byte pod list this lane holds.
//   * rows: lane k = 4q + j of a row loads hashes k and 16 + k (two 8-byte loads: 128 contiguous bytes per row) and computes
//     their home buckets; step i gets hash and bucket of key 4i + q by ds_bpermute (the LDS crossbar is otherwise idle);
//   * probe: 8 loads of 16 bytes; each lane compares its two words with the key; the slot is OR-reduced over the quad by DPP;
//     found bits -> a 32-bit word per row (two DPP rotations) -> leading hits m = count of trailing ones;
//   * lists: step i loads piece j of the list of hit 4i + q; "all lists identical" = every step equals step 0 (same lane) and
//     every quad equals its neighbour quad at step 0 (DPP row_shr:4); lane (q, j) then evaluates id 4q + j of the common list
//     (and id 16 + 4q + j when the list is that long) -- the ids sit in its own piece;
//   * LoRA tier bits: the owning lane word of the adapter's two planes (one load each); base[] and the prefix terms from LDS;
//     binary64 adds in chain order, exactly pick_fast_kernel's pod_total;
//   * argmax: a DPP max reduction over the ROW (4 steps) + a row minimum of the pod index among the ties;
//   * best pod outside the list: lane k holds entry k of the adapter's top table; "is entry k listed?" through a per-row bitmap
//     in LDS (ds_or / ds_read / clear).
// Anything else -- reserved hashes, an overflowed or differing list, a chain that continues past 32 hits, an exhausted table, an
// out-of-range row, a looked-up key that was displaced from its home bucket -- is DEFERRED: the row's lane 0 appends the request
// index to the wavefront's private segment of the work list and pick_fast_kernel<..., WL = true> scores exactly those requests
// afterwards (same stream).  The FIRST missing key of a request whose home bucket has overflowed is followed through the next
// buckets right here (0.03 % of the buckets at load 1/4: a 64k batch always has a few; confirmed absent = nothing to defer).
// Software pipeline per wavefront: request rows two blocks ahead, key buckets one block ahead, lists / tables / tier words of
// the current block; the loads of later blocks are queued BEHIND everything the current block waits for (loads retire in order).
#ifndef EPPK_QUAD_MAX_THREADS
#define EPPK_QUAD_MAX_THREADS 1024   // largest workgroup it may be launched with (the host launches 512: two workgroups per CU)
#endif
#ifndef EPPK_QUAD_PREFETCH
#define EPPK_QUAD_PREFETCH 0    // > 0: the rows of the block this many iterations ahead are pulled into L2 (measured: a wavefront
                                // walks only ~4 blocks of a 64k batch; 6 ahead cost 0.4 us, profiles/r02_g_quad_ablations.txt)
#endif
#ifndef EPPK_QUAD_PIPE_KEYS
#define EPPK_QUAD_PIPE_KEYS 1   // 1: the key gather of the next block is in flight while this one is evaluated
#endif
#ifndef EPPK_QUAD_PARK
#define EPPK_QUAD_PARK 1        // 0: masked single picks defer the rows they cannot score from base[] and the top table (A/B runs)
#endif
#ifndef EPPK_QUAD_WAVES
#define EPPK_QUAD_WAVES 4       // wavefronts per SIMD the register allocation aims at (<= 128 VGPRs)
#endif
// MASKED: a candidate row per request (natural layout, [J] u64: request.go:104-133 as a bitmask).  Lane k of a row loads words 4k .. 4k+3
// (32 contiguous bytes), ANDs them with the active pods and parks them in the row's LDS area, where "is pod p a candidate?" is one
// ds_read_b64 for a listed pod or a table entry.  No candidate at all: EPPK_NO_PICK right here.  Candidates that miss the snapshot-wide
// QUEUE extremes need the request's own normalisers: deferred (pick_fast_kernel's exact evaluation), like a request whose first 16 table
// entries hold no candidate outside its list.
// TOPK: ordered fallbacks (eppk_pick_topk: out_pick / out_score hold topk entries per request).  The candidates of a row sit in its
// lanes -- the listed pods' totals (ids 4q + j and 16 + 4q + j) and the 16 table entries that are not listed -- so round i of the
// merge is one more row argmax over each lane's best remaining candidate; the winner leaves the pool.  A round that finds the table's
// pool empty although the table goes on beyond its 16th entry cannot know the next value: deferred.
// TAIL: no second launch.  A wavefront keeps the requests it defers in a segment of its own, so when a workgroup's loop is over the
// work-list form of pick_fast_kernel's body runs right there, over the segments of that workgroup's own wavefronts (pick_fast_body:
// the workgroup's LDS is re-staged in that kernel's layout -- only when it deferred anything, which a workgroup of the common batch
// does not).  Nobody waits for another workgroup, and a burst of deferred requests is spread over the whole grid.  (Round 3 gave the
// whole launch's work list to the LAST workgroup to arrive: one workgroup was no match for a list of hundreds, so the library kept a
// two-launch form beside it and chose between them from the deferred counts of EARLIER launches -- a latency cliff for the first batch
// of a burst.)  A second launch behind every batch cost 4-10 us of a 14 us pick although it had nothing to do: its workgroups cannot
// start before this kernel's persistent wavefronts leave (all 512 VGPRs per SIMD are theirs), and the host pays for two launches per
// batch.  The arrival counters survive for the report only (how much the launch deferred: the library's back-off feedback).
// The work-list pass of the one-launch form as a function of its OWN (never inlined).  Inlined, pick_fast_body shared the kernel's
// register allocation and the hot loop paid for it (round 3: 128 VGPRs, 24 bytes of scratch, 5 VGPR and 97 SGPR spills against
// 112 / 0 / 0 / 8 without the tail) although one workgroup per launch runs it, and almost never.  The function takes NO arguments
// beyond the LDS base: handed the kernel's argument structs -- by reference or by value -- the compiler copies them to the stack in
// the kernel's prologue (every wavefront pays) and, by reference, re-reads them from there inside the hot loop.  It reads the
// kernel's arguments itself, from the kernarg segment, through a struct that mirrors pick_quad_kernel's parameter list.
// Byte offset of the crossbar scratch of pick_quad_body (wpb x 512 bytes) in a LAUNCHED workgroup's LDS: right behind the quad layout --
// base | lw | pterm | "listed" bits (| natural sets | candidate words) -- rounded up to 16 bytes.  (The resident kernels keep it at the
// end of their allocation: the fast body's histogram, which must stay all-zero between two doorbells, overlaps the quad layout's tail.)
__host__ __device__ __forceinline__ constexpr uint32_t quad_xbar_off(uint32_t J, uint32_t pwn, uint32_t wpb, bool masked) {
  return ((J * 64u * 8u + 32u + pwn * 8u + wpb * 4u * J * 8u + (masked ? 192u * 8u + wpb * 4u * J * 8u : 0u)) + 15u) & ~15u;
}
struct QuadKernArgs {
  KSnap sn; KIndex ix; KTail tl; const uint8_t* reqs; uint32_t stride, n_reqs, pwn; const uint64_t* cand_mask; int32_t* out_pick; double* out_score;
  unsigned long long* stats; uint32_t* defer_cnt; uint32_t* defer_list; uint32_t defer_cap; uint32_t* defer_total; uint32_t* defer_total_next; uint32_t topk;
  uint32_t* done_ctr; uint32_t* report;
  KChain chain;    // (MASKED tails: the exact evaluation of a request whose candidates miss the snapshot-wide QUEUE extremes needs the chain)
  uint32_t* learn_out;
};
// (a callable function is not handed the kernarg segment pointer itself -- llvm.amdgcn.kernarg.segment.ptr is null there -- only the
// pointer to the IMPLICIT arguments, which lie right behind the explicit ones, 8-byte aligned: walk back from there)
static_assert(((sizeof(QuadKernArgs) + 7u) & ~(size_t)7u) == 520u, "QuadKernArgs must mirror pick_quad_kernel's parameter list (its explicit kernarg bytes: .kernarg_segment_size - 256)");
__device__ __forceinline__ const QuadKernArgs* quad_kernargs() {
  return (const QuadKernArgs*)((const char*)__builtin_amdgcn_implicitarg_ptr() - ((sizeof(QuadKernArgs) + 7u) & ~(size_t)7u));
}
// (1) The requests THIS workgroup deferred: the work-list form of pick_fast_kernel's body over the segments of its own wavefronts (the
//     workgroup's LDS is re-staged in that kernel's layout).  Called by a workgroup that deferred something -- rarely: the function
//     saves and restores some fifty callee-saved registers through scratch, which every wavefront of every launch would pay otherwise.
template <typename LW, bool HAS_L, bool P_FIRST, bool MASKED, bool TOPK>
__device__ __attribute__((noinline)) void quad_tail_pass(unsigned char* smem) {
  const QuadKernArgs* a = quad_kernargs();
  KWork wk;
  wk.cnt = a->defer_cnt; wk.list = a->defer_list; wk.total = a->defer_total; wk.report = a->report; wk.cap = a->defer_cap; wk.n_segs = gridDim.x * (blockDim.x >> 6);
  pick_fast_body<LW, 6, HAS_L, true, P_FIRST, MASKED, /*BIG*/ true, /*GEN*/ false, TOPK, /*WL*/ true>(
      blockIdx.x, gridDim.x, 0u, true, smem, a->sn, a->ix, a->tl, a->reqs, a->stride, a->n_reqs, a->pwn, a->cand_mask, a->chain, a->out_pick, a->out_score, a->stats, a->topk, wk);
}
// (2) The report (the library's feedback: how much the launch deferred) by the LAST workgroup to arrive.  "Who is last?" in two levels
//     -- the persistent workgroups all finish within a microsecond of each other, and 512 atomics on ONE word queue up for 6 us: 16
//     group counters (done_ctr[1 + (block & 15)]), the workgroup that completes its group bumps the top counter (done_ctr[0]), the
//     one that completes that is the last of the launch.  Thread 0 of every workgroup, behind a barrier.
__device__ __attribute__((noinline)) void quad_tail_report() {
  const QuadKernArgs* a = quad_kernargs();
  uint32_t* done_ctr = a->done_ctr;
  const uint32_t grp = blockIdx.x & 15u, n_grp = gridDim.x < 16u ? gridDim.x : 16u;
  const uint32_t in_grp = (gridDim.x - grp + 15u) / 16u;                            // workgroups with this group number
  if (atomicAdd(&done_ctr[1u + grp], 1u) == in_grp - 1u) {
    __hip_atomic_store(&done_ctr[1u + grp], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (for this buffer set's next launch)
    if (atomicAdd(&done_ctr[0], 1u) == n_grp - 1u) {
      __hip_atomic_store(&done_ctr[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *a->report = __hip_atomic_load(a->defer_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (pinned host word)
    }
  }
}

// What the functions that score PARKED rows read ((3), (3b), (3c) below): pointers into the launched kernel's kernarg segment, or into the
// resident kernels' argument block.  Filled in by pick_quad_body where a wavefront has parked rows -- rarely -- and handed over by address.
struct ParkArgs {
  const KSnap* sn; const KChain* chain;
  int32_t* out_pick; double* out_score;
  const uint64_t* cand_mask; const uint32_t* defer_list;
  uint32_t defer_cap, n_reqs, pwn, vblock, vgrid, topk;
};
// (3) MASKED single picks whose candidates miss a snapshot-wide QUEUE extreme -- base[] and the top tables embed the snapshot-wide
//     normalisers, the request needs its own (request.go:104-133 + the queue scorer's min / max over the CANDIDATES) -- scored where
//     pick_quad_kernel finds them, by the whole wavefront, one row after the other.  Until round 6 such a request was deferred, and the
//     work-list pass spent ~25 us on it (tail function entered 2.9 us after the wavefront left the loop, LDS re-staged 6.4, keys landed
//     10.7, lists 12+, pick stored 26.6: one request behind a chain of cold round trips, profiles/r06_masked_tail_stamps.txt) -- at 1/8
//     density one request in 1200 misses an extreme, and each of them ended its workgroup, i.e. the launch.  Here the row's state is
//     at hand: the candidates in LDS, the listed pods and their matched counts in the row's lanes.  Every candidate that is not listed
//     is evaluated with matched = 0 (exact_sweep_nat over the candidate words as they lie in LDS, four candidates per trip), the listed candidates
//     by the lanes that hold them; same expressions, same order as masked_exact.  Never inlined (the hot loop is compiled as if it
//     were not there: the loop only PARKS such a row -- index, listed pods, counts -- and the wavefront comes here when its loop is over);
//     handed the kernel's arguments by address (ParkArgs: the kernarg segment of a launched kernel, the argument block of a resident one).
//     `rows`: bit g = row g of the wavefront wants it (wave-uniform).  ls: bit 0 = pA, bit 1 = pB is a listed CANDIDATE of this lane's row.
template <typename LW, bool HAS_L>
__device__ __forceinline__ void quad_exact_rows_i(const uint32_t rows, const uint32_t r, const uint32_t nb, const uint32_t arow, const uint32_t pA, const uint32_t pB,
                                                  const uint32_t cntA, const uint32_t cntB, const uint32_t ls, const uint64_t* s_cn_w, uint32_t* s_bits_w,
                                                  const ParkArgs* a) {
  const KSnap& sn = *a->sn;
  const int lane = (int)(threadIdx.x & 63u);
  const uint32_t J = sn.J, g_mine = (uint32_t)lane >> 4;
  const ExactChain ec = exact_chain(*a->chain);
  for (uint32_t g = 0; g < 4u; ++g) {
    if (!((rows >> g) & 1u)) continue;
    const uint32_t rg = (uint32_t)__builtin_amdgcn_readlane((int)r, (int)(16u * g)), nbg = (uint32_t)__builtin_amdgcn_readlane((int)nb, (int)(16u * g));
    const uint32_t ag = (uint32_t)__builtin_amdgcn_readlane((int)arow, (int)(16u * g));
    const uint64_t cand = (uint32_t)lane < J ? s_cn_w[g * J + (uint32_t)lane] : 0ull;     // (mask & existing, active pods: pick_quad_body)
    uint32_t* bits = s_bits_w + g * (2u * J);
    const bool mine = g_mine == g;
    const bool hasA = mine && (ls & 1u), hasB = mine && (ls & 2u);
    if (hasA) atomicOr(&bits[pA >> 5], 1u << (pA & 31u));
    if (hasB) atomicOr(&bits[pB >> 5], 1u << (pB & 31u));
    uint32_t qmin = 0u, qmax = 0u;
    if (ec.has_q) exact_qrange<uint64_t, true>(sn.queue, cand, lane, qmin, qmax);
    double best = -__builtin_inf();
    uint32_t bidx = kNoPod;
    exact_sweep_nat<LW, HAS_L>(sn, ec, qmin, qmax, cand, ag, lane, bits, best, bidx);
    // the listed candidates, by the lanes that hold them
    const double qden = (double)(qmax - qmin), nbd = (double)nbg;
    const LW* thl = (const LW*)((const uint8_t*)sn.blob + SnapOff<LW>::thl) + (size_t)ag * 128u;
    if (__any(hasA || hasB)) {
      const uint32_t p0 = hasA ? pA : 0u, p1 = hasB ? pB : 0u;
      const uint32_t q0 = sn.queue[p0], q1 = sn.queue[p1];
      const double kv0 = sn.kv[p0], kv1 = sn.kv[p1];
      uint32_t tier0 = 0u, tier1 = 0u;
      if (HAS_L) {
        const LW h0 = thl[(p0 & 63u) * 2u], l0 = thl[(p0 & 63u) * 2u + 1u], h1 = thl[(p1 & 63u) * 2u], l1 = thl[(p1 & 63u) * 2u + 1u];
        tier0 = (uint32_t)((((uint64_t)h0 >> (p0 >> 6)) & 1ull) << 1) | (uint32_t)(((uint64_t)l0 >> (p0 >> 6)) & 1ull);
        tier1 = (uint32_t)((((uint64_t)h1 >> (p1 >> 6)) & 1ull) << 1) | (uint32_t)(((uint64_t)l1 >> (p1 >> 6)) & 1ull);
      }
      const double sp0 = nbg != 0u ? clamp01((double)cntA / nbd) : 0.0, sp1 = nbg != 0u ? clamp01((double)cntB / nbd) : 0.0;
      const double t0 = exact_total(ec, qmin, qmax, qden, q0, kv0, tier0, sp0);
      if (hasA && (t0 > best || (t0 == best && p0 < bidx))) { best = t0; bidx = p0; }
      if (__any(hasB)) {
        const double t1 = exact_total(ec, qmin, qmax, qden, q1, kv1, tier1, sp1);
        if (hasB && (t1 > best || (t1 == best && p1 < bidx))) { best = t1; bidx = p1; }
      }
    }
    wave_argmax_dpp(best, bidx);
    if (hasA) bits[pA >> 5] = 0u;
    if (hasB) bits[pB >> 5] = 0u;
    const bool none = bidx == kNoPod;
    if (lane == 0) {
      a->out_pick[rg] = none ? -1 : (int32_t)bidx;
      if (a->out_score) a->out_score[rg] = none ? 0.0 : best;
    }
  }
}

// The never-inlined forms of a LAUNCHED kernel put their ParkArgs together themselves, from the kernarg segment (scalar loads; the struct
// dissolves into registers).  Handed over by address from the caller's frame it lived in scratch, and every function began with a chain of
// scratch -> kernarg -> data loads: +5 us per call (1/8-density batch 43.8 -> 49.2 us, 64k x 8 endpoints 74 -> 92).
__device__ __forceinline__ ParkArgs park_args_launched() {
  const QuadKernArgs* a = quad_kernargs();
  ParkArgs pa;
  pa.sn = &a->sn; pa.chain = &a->chain; pa.out_pick = a->out_pick; pa.out_score = a->out_score; pa.cand_mask = a->cand_mask; pa.defer_list = a->defer_list;
  pa.defer_cap = a->defer_cap; pa.n_reqs = a->n_reqs; pa.pwn = a->pwn; pa.vblock = blockIdx.x; pa.vgrid = gridDim.x; pa.topk = a->topk;
  return pa;
}
template <typename LW, bool HAS_L>
__device__ __attribute__((noinline)) void quad_exact_rows(const uint32_t rows, const uint32_t r, const uint32_t nb, const uint32_t arow, const uint32_t pA, const uint32_t pB,
                                                          const uint32_t cntA, const uint32_t cntB, const uint32_t ls, const uint64_t* s_cn_w, uint32_t* s_bits_w) {
  const ParkArgs pa = park_args_launched();
  quad_exact_rows_i<LW, HAS_L>(rows, r, nb, arow, pA, pB, cntA, cntB, ls, s_cn_w, s_bits_w, &pa);
}
__device__ __forceinline__ uint32_t row16_any(const bool b, const uint32_t lane) { return (uint32_t)(__ballot(b) >> (lane & 48u)) & 0xFFFFu; }
// (3b) The same for FOUR parked rows at once, each scored by its own 16 lanes (lane k: words 4k .. 4k+3 of the row's candidates, handed
//      over in registers): the form for batches in which every row is parked -- a subset filter that leaves a handful of endpoints
//      (request.go:104-133: the realistic mask) misses a QUEUE extreme in nearly every request.  One trip serves four candidates of
//      every lane of every row; row-wide reductions by DPP.  (One row at a time with 64 lanes is the better form where parked rows are
//      rare and their candidates many -- a 1/8-density mask: (3) above; the caller chooses by the number of rows it has.)
//      TOPK (ordered fallbacks, eppk_pick_topk): a->topk rounds of the same sweep over the candidates not yet reported -- the QUEUE normalisers
//      range over ALL candidates in every round (masked_exact: `cand` / `eval`); a reported pod joins the "listed" bits, which the sweep
//      leaves out, or -- a listed candidate -- leaves its lane's pool.  Always this form for fallback lists, also for a single row.
template <typename LW, bool HAS_L, bool TOPK = false>
__device__ __forceinline__ void quad_exact_rows_par_i(const uint32_t rows, const uint32_t r, const uint32_t nb, const uint32_t arow, const uint32_t pA, const uint32_t pB,
                                                      const uint32_t cntA, const uint32_t cntB, const uint32_t ls, const uint64_t c0, const uint64_t c1, const uint64_t c2,
                                                      const uint64_t c3, uint32_t* s_bits_w, const ParkArgs* a) {
  const KSnap& sn = *a->sn;
  const uint32_t lane = threadIdx.x & 63u, k = lane & 15u, g = lane >> 4;
  const bool on = (rows >> g) & 1u;
  const ExactChain ec = exact_chain(*a->chain);
  uint32_t* bits = s_bits_w + g * (2u * sn.J);
  const bool hasA = on && (ls & 1u), hasB = on && (ls & 2u);
  if (hasA) atomicOr(&bits[pA >> 5], 1u << (pA & 31u));
  if (hasB) atomicOr(&bits[pB >> 5], 1u << (pB & 31u));
  // the next candidate of this lane, over its four words in ascending order (0 with v = false when it has none left)
  auto next = [&](uint64_t (&c)[4], bool& v) -> uint32_t {
    const uint32_t ws = c[0] ? 0u : c[1] ? 1u : c[2] ? 2u : 3u;
    const uint64_t cur = ws == 0u ? c[0] : ws == 1u ? c[1] : ws == 2u ? c[2] : c[3];
    v = cur != 0ull;
    const uint32_t j = v ? (uint32_t)__builtin_ctzll(cur) : 0u;
    const uint64_t rest = cur & (cur - 1ull);
    c[0] = ws == 0u ? rest : c[0]; c[1] = ws == 1u ? rest : c[1]; c[2] = ws == 2u ? rest : c[2]; c[3] = ws == 3u ? rest : c[3];
    return v ? (4u * k + ws) * 64u + j : 0u;
  };
  constexpr int U = 4;
  uint32_t qmin = 0u, qmax = 0u;
  if (ec.has_q) {
    uint32_t mn = 0xFFFFFFFFu, mx = 0u;
    uint64_t c[4] = {on ? c0 : 0ull, on ? c1 : 0ull, on ? c2 : 0ull, on ? c3 : 0ull};
    while (__any((c[0] | c[1] | c[2] | c[3]) != 0ull)) {
      uint32_t q[U];
      bool v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) q[u] = sn.queue[next(c, v[u])];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        mn = (v[u] && q[u] < mn) ? q[u] : mn;
        mx = (v[u] && q[u] > mx) ? q[u] : mx;
      }
    }
    mn = dpp_min_u32<0xB1, 0xf>(mn); mn = dpp_min_u32<0x4E, 0xf>(mn); mn = dpp_min_u32<0x141, 0xf>(mn); mn = dpp_min_u32<0x140, 0xf>(mn);
    mx = ~mx;                                                         // (maximum = complement of the minimum of the complements)
    mx = dpp_min_u32<0xB1, 0xf>(mx); mx = dpp_min_u32<0x4E, 0xf>(mx); mx = dpp_min_u32<0x141, 0xf>(mx); mx = dpp_min_u32<0x140, 0xf>(mx);
    qmin = mn; qmax = ~mx;
  }
  const double qden = (double)(qmax - qmin), nbd = (double)nb;
  const LW* thl = (const LW*)((const uint8_t*)sn.blob + SnapOff<LW>::thl) + (size_t)(on ? arow : 128u) * 128u;
  auto tier_of = [&](LW th, LW tl_, uint32_t p) -> uint32_t {
    return (uint32_t)((((uint64_t)th >> (p >> 6)) & 1ull) << 1) | (uint32_t)(((uint64_t)tl_ >> (p >> 6)) & 1ull);
  };
  // the listed candidates' totals, by the lanes that hold them (once: the rounds only choose among them)
  double t0 = -__builtin_inf(), t1 = -__builtin_inf();
  bool vA = hasA, vB = hasB;
  if (__any(hasA || hasB)) {
    const uint32_t p0 = hasA ? pA : 0u, p1 = hasB ? pB : 0u;
    const uint32_t q0 = sn.queue[p0], q1 = sn.queue[p1];
    const double kv0 = sn.kv[p0], kv1 = sn.kv[p1];
    uint32_t tier0 = 0u, tier1 = 0u;
    if (HAS_L) {
      tier0 = tier_of(thl[(p0 & 63u) * 2u], thl[(p0 & 63u) * 2u + 1u], p0);
      tier1 = tier_of(thl[(p1 & 63u) * 2u], thl[(p1 & 63u) * 2u + 1u], p1);
    }
    const double sp0 = nb != 0u ? clamp01((double)cntA / nbd) : 0.0, sp1 = nb != 0u ? clamp01((double)cntB / nbd) : 0.0;
    t0 = exact_total(ec, qmin, qmax, qden, q0, kv0, tier0, sp0);
    if (__any(hasB)) t1 = exact_total(ec, qmin, qmax, qden, q1, kv1, tier1, sp1);
  }
  const uint32_t tk = TOPK ? a->topk : 1u;
  uint32_t reported = kNoPod;                                         // TOPK: the unlisted pod this lane has added to the bits (lane k: round k's)
  for (uint32_t round = 0; round < tk; ++round) {
    double best = -__builtin_inf();
    uint32_t bidx = kNoPod;
    uint64_t c[4] = {on ? c0 : 0ull, on ? c1 : 0ull, on ? c2 : 0ull, on ? c3 : 0ull};
    while (__any((c[0] | c[1] | c[2] | c[3]) != 0ull)) {
      uint32_t q[U], hw[U], pp[U];
      double kv[U];
      LW th[U], tl_[U];
      bool v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t p = next(c, v[u]);
        pp[u] = p;
        q[u] = sn.queue[p];
        kv[u] = sn.kv[p];
        hw[u] = bits[p >> 5];
        if (HAS_L) { th[u] = thl[(p & 63u) * 2u]; tl_[u] = thl[(p & 63u) * 2u + 1u]; }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t p = pp[u];
        const bool take = v[u] && !((hw[u] >> (p & 31u)) & 1u);
        const double t = exact_total(ec, qmin, qmax, qden, q[u], kv[u], HAS_L ? tier_of(th[u], tl_[u], p) : 0u, 0.0);
        if (take && (t > best || (t == best && p < bidx))) { best = t; bidx = p; }
      }
    }
    if (vA && (t0 > best || (t0 == best && pA < bidx))) { best = t0; bidx = pA; }
    if (vB && (t1 > best || (t1 == best && pB < bidx))) { best = t1; bidx = pB; }
    // argmax over the row: (total desc, pod asc)
    double wmax = best;
    wmax = vmax_f64(wmax, dpp_f64<0xB1, 0xf>(wmax));
    wmax = vmax_f64(wmax, dpp_f64<0x4E, 0xf>(wmax));
    wmax = vmax_f64(wmax, dpp_f64<0x141, 0xf>(wmax));
    wmax = vmax_f64(wmax, dpp_f64<0x140, 0xf>(wmax));
    uint32_t widx = best == wmax ? bidx : kNoPod;
    widx = dpp_min_u32<0xB1, 0xf>(widx); widx = dpp_min_u32<0x4E, 0xf>(widx); widx = dpp_min_u32<0x141, 0xf>(widx); widx = dpp_min_u32<0x140, 0xf>(widx);
    if (on && k == 0u) {
      const bool none = widx == kNoPod;
      a->out_pick[(size_t)r * tk + round] = none ? -1 : (int32_t)widx;
      if (a->out_score) a->out_score[(size_t)r * tk + round] = none ? 0.0 : wmax;
    }
    if constexpr (TOPK) {                                             // the winner leaves the pool
      const bool listedA = vA && pA == widx, listedB = vB && pB == widx;
      const uint32_t was_listed = row16_any(listedA || listedB, lane);
      if (listedA) vA = false;
      if (listedB) vB = false;
      if (on && widx != kNoPod && was_listed == 0u && k == (round & 15u)) {
        atomicOr(&bits[widx >> 5], 1u << (widx & 31u));
        reported = widx;
      }
      wave_lds_fence();
    }
  }
  wave_lds_fence();
  if (hasA) bits[pA >> 5] = 0u;
  if (hasB) bits[pB >> 5] = 0u;
  if (reported != kNoPod) bits[reported >> 5] = 0u;
}

template <typename LW, bool HAS_L, bool TOPK = false>
__device__ __attribute__((noinline)) void quad_exact_rows_par(const uint32_t rows, const uint32_t r, const uint32_t nb, const uint32_t arow, const uint32_t pA, const uint32_t pB,
                                                              const uint32_t cntA, const uint32_t cntB, const uint32_t ls, const uint64_t c0, const uint64_t c1, const uint64_t c2,
                                                              const uint64_t c3, uint32_t* s_bits_w) {
  const ParkArgs pa = park_args_launched();
  quad_exact_rows_par_i<LW, HAS_L, TOPK>(rows, r, nb, arow, pA, pB, cntA, cntB, ls, c0, c1, c2, c3, s_bits_w, &pa);
}
// (3c) The parked rows of one wavefront, four at a time: row g of the wavefront takes entry base + g.  LDS layout and work-list geometry
//      as in pick_quad_body.  INL: everything inlined into the caller -- the resident kernels, where ONE call inside the doorbell loop
//      tripled the kernel's spill code (944 scratch loads against 353) and cost a 16-request batch of dense masks 5-7 us.
template <typename LW, bool HAS_L, bool INL, bool TOPK = false>
__device__ __forceinline__ void quad_park_drain_i(const uint32_t n_x, unsigned char* smem, const ParkArgs* a) {
  const KSnap& sn = *a->sn;
  const uint32_t lane = threadIdx.x & 63u, k = lane & 15u, g = lane >> 4, wpb = blockDim.x >> 6, wave = threadIdx.x >> 6;
  const uint32_t gwave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(a->vblock * wpb + wave)), nwaves = a->vgrid * wpb;
  const uint32_t J = sn.J, bits_dw = J * 2u, cap = a->defer_cap;
  uint32_t* s_bits_all = (uint32_t*)((double*)smem + (size_t)J * 64u + 4u + a->pwn);
  const uint64_t* s_nat = (const uint64_t*)(s_bits_all + (blockDim.x >> 4) * bits_dw);
  uint64_t* s_cn_all = (uint64_t*)s_nat + 192;
  uint64_t* s_cn = s_cn_all + (wave * 4u + g) * J;
  const uint32_t* my_xr = a->defer_list + ((size_t)nwaves + gwave) * cap;
  const uint32_t* my_xs = a->defer_list + (size_t)nwaves * cap * 2u + (size_t)gwave * cap * 32u;
  const __amdgpu_buffer_rsrc_t rmk = __builtin_amdgcn_make_buffer_rsrc((void*)a->cand_mask, 0, (int)((size_t)a->n_reqs * J * 8u), 0x00020000);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // (this wavefront's own stores have reached the L2)
  for (uint32_t base = 0; base < n_x; base += 4u) {
    const bool have = base + g < n_x;
    const uint32_t xi = have ? base + g : base;
    const uint32_t rr = __hip_atomic_load(&my_xr[xi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t e0 = __hip_atomic_load(&my_xs[(size_t)xi * 32u + k * 2u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t e1 = __hip_atomic_load(&my_xs[(size_t)xi * 32u + k * 2u + 1u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t xarow = (e1 >> 14) & 0xFFu, xnb = (e1 >> 22) & 0x7Fu;           // (the row's header, as the loop saw it)
    // the row's candidate words again: mask & active, as in the loop
    const uint32_t mo = (rr * J + 4u * k) * 8u;
    const u32x4_t mk0 = __builtin_amdgcn_raw_buffer_load_b128(rmk, (int)mo, 0, 0), mk1 = __builtin_amdgcn_raw_buffer_load_b128(rmk, (int)(mo + 16u), 0, 0);
    uint64_t cw[4] = {u64_of(mk0.x, mk0.y), u64_of(mk0.z, mk0.w), u64_of(mk1.x, mk1.y), u64_of(mk1.z, mk1.w)};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t w = 4u * k + (uint32_t)i;
      cw[i] = w < J ? cw[i] & s_nat[w] : 0ull;
    }
    const unsigned long long hm = __ballot(have);
    const uint32_t rows = (uint32_t)(hm & 1ull) | (uint32_t)((hm >> 15) & 2ull) | (uint32_t)((hm >> 30) & 4ull) | (uint32_t)((hm >> 45) & 8ull);
    if (TOPK || (rows & (rows - 1u))) {                                // two rows or more (fallback lists: always): each by its own 16 lanes, side by side
      if constexpr (INL) quad_exact_rows_par_i<LW, HAS_L, TOPK>(rows, rr, xnb, xarow, e0 & 0xFFFFu, e0 >> 16, e1 & 0x3Fu, (e1 >> 6) & 0x3Fu, (e1 >> 12) & 3u,
                                                                cw[0], cw[1], cw[2], cw[3], s_bits_all + wave * 4u * bits_dw, a);
      else quad_exact_rows_par<LW, HAS_L, TOPK>(rows, rr, xnb, xarow, e0 & 0xFFFFu, e0 >> 16, e1 & 0x3Fu, (e1 >> 6) & 0x3Fu, (e1 >> 12) & 3u,
                                                cw[0], cw[1], cw[2], cw[3], s_bits_all + wave * 4u * bits_dw);
      continue;
    }
    if constexpr (!TOPK) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {                                      // one row: the whole wavefront on it, its candidate words through LDS
      const uint32_t w = 4u * k + (uint32_t)i;
      if (w < J) s_cn[w] = cw[i];
    }
    wave_lds_fence();
    if constexpr (INL) quad_exact_rows_i<LW, HAS_L>(rows, rr, xnb, xarow, e0 & 0xFFFFu, e0 >> 16, e1 & 0x3Fu, (e1 >> 6) & 0x3Fu, (e1 >> 12) & 3u,
                                                    s_cn_all + (size_t)wave * 4u * J, s_bits_all + wave * 4u * bits_dw, a);
    else quad_exact_rows<LW, HAS_L>(rows, rr, xnb, xarow, e0 & 0xFFFFu, e0 >> 16, e1 & 0x3Fu, (e1 >> 6) & 0x3Fu, (e1 >> 12) & 3u,
                                    s_cn_all + (size_t)wave * 4u * J, s_bits_all + wave * 4u * bits_dw);
    }
  }
}

// (3d) ... behind ONE small call from the loop's kernel: the argument block is put together here, not in the caller's frame
template <typename LW, bool HAS_L, bool TOPK>
__device__ __attribute__((noinline)) void quad_park_drain_launched(const uint32_t n_x, unsigned char* smem) {
  const ParkArgs pa = park_args_launched();
  quad_park_drain_i<LW, HAS_L, false, TOPK>(n_x, smem, &pa);
}
template <typename LW, bool HAS_L, bool TOPK, typename RA>
__device__ __forceinline__ void quad_park_drain_resident(const uint32_t n_x, unsigned char* smem, const RA* ra, const uint32_t bufset, const uint32_t n_reqs, const uint32_t topk) {
  ParkArgs pa;
  pa.sn = &ra->sn; pa.chain = &ra->chain; pa.out_pick = ra->buf[bufset].out_pick; pa.out_score = ra->buf[bufset].out_score; pa.cand_mask = ra->buf[bufset].mask;
  pa.defer_list = ra->defer_list; pa.defer_cap = ra->defer_cap; pa.n_reqs = n_reqs; pa.pwn = ra->pwn; pa.vblock = 0u; pa.vgrid = 1u; pa.topk = topk;
  quad_park_drain_i<LW, HAS_L, true, TOPK>(n_x, smem, &pa);
}

// LEARN (single picks only): the kernel also leaves one word per request for the post-route index update that follows the pick
// (index_insert_picks_kernel: `learn`) -- bits 0..7 = m, the leading blocks of the request it found in the index; bits 8..23 = pick + 1
// (so that the update does not have to fetch the pick from wherever the caller wanted it: pinned host memory on the staged paths);
// bit 31 = the picked pod is on the (common) pod list of all m, i.e. those m (hash, pod) pairs are in the index already and the update
// only has to refresh their stamps.  0 = nothing to tell (a request it deferred, or one without a pick): the update then reads picks[r]
// and takes the whole path.  Costs the pick one 4-byte store per request; saves the update two of the three line look-ups of every
// known pair (1 Mi per closed-loop step of a 64k x 32-block batch).
// The body of pick_quad_kernel -- everything but the end of the one-launch form -- as a function: the kernel below is one caller, the
// RESIDENT small-batch kernel (pick_resident_kernel: one workgroup behind a doorbell, tables already in LDS, request rows in pinned
// host memory) the other.  vblock / vgrid = this workgroup's place among the workgroups that share the batch.  Returns the number of
// requests this WAVEFRONT deferred (wave-uniform): they are in its segment of the work list.
template <typename LW, bool HAS_L, bool P_FIRST, bool MASKED, bool TOPK, bool LEARN, bool RESIDENT = false, typename RA = void>
__device__ __forceinline__ uint32_t pick_quad_body(const uint32_t vblock, const uint32_t vgrid, unsigned char* smem, const KSnap& sn, const KIndex& ix, const KTail& tl,
                                                   const uint8_t* __restrict__ reqs, uint32_t stride, uint32_t n_reqs, uint32_t pwn, const uint64_t* __restrict__ cand_mask,
                                                   int32_t* __restrict__ out_pick, double* __restrict__ out_score, unsigned long long* __restrict__ stats,
                                                   uint32_t* __restrict__ defer_cnt, uint32_t* __restrict__ defer_list, uint32_t defer_cap,
                                                   uint32_t* __restrict__ defer_total, uint32_t topk, uint32_t* __restrict__ learn_out, const uint32_t xbar_off,
                                                   const RA* = nullptr) {
  static_assert(!(LEARN && TOPK), "learn words go with single picks");
  double* s_base = (double*)smem;
  double* s_lw = s_base + (size_t)sn.J * 64u;
  double* s_pterm = s_lw + 4;
  uint32_t* s_bits_all = (uint32_t*)(s_pterm + pwn);          // [waves][4 rows][J * 2] dwords: one bit per pod ("listed")
  const uint32_t bits_dw = sn.J * 2u;
  // MASKED: the three natural-layout sets of the snapshot (active pods, pods at the minimum / maximum queue depth: [3][64] u64) and,
  // per row, the request's candidates (mask & active: [J] u64)
  uint64_t* s_nat = (uint64_t*)(s_bits_all + (blockDim.x >> 4) * bits_dw);
  uint64_t* s_cn_all = s_nat + 192;
  // The wavefront's CROSSBAR scratch (512 bytes at byte offset xbar_off of the workgroup's LDS, 16-byte aligned: quad_xbar_off): the two
  // pieces of a bucket that hold meta dwords -- 32 bytes per quad -- written by lanes j = 0, 1 and read back by all four lanes at the
  // dword the match code names.  One region serves every probe step: LDS operations of one wavefront execute in order.
  unsigned char* s_xbar = smem + xbar_off + (threadIdx.x >> 6) * 512u + (((uint32_t)threadIdx.x & 63u) >> 2) * 32u;     // this quad's 32 bytes
  const int lane = (int)(threadIdx.x & 63u);
  const uint32_t k = (uint32_t)lane & 15u, g = (uint32_t)lane >> 4, gsh = (uint32_t)lane & 48u;
  const uint32_t q = ((uint32_t)lane >> 2) & 3u, j = (uint32_t)lane & 3u, j16 = j * 16u;
  const uint32_t wpb = blockDim.x >> 6;
  const uint32_t gwave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(vblock * wpb + (threadIdx.x >> 6)));
  const uint32_t nwaves = (uint32_t)__builtin_amdgcn_readfirstlane((int)(vgrid * wpb));
  const uint32_t nblk = (n_reqs + 3u) >> 2;                          // blocks of four requests
  const bool idle = gwave >= nblk;                                   // (more wavefronts than blocks: it still helps staging the tables)
  uint32_t* bits = s_bits_all + ((threadIdx.x >> 6) * 4u + g) * bits_dw;
  uint64_t* s_cn = s_cn_all + ((threadIdx.x >> 6) * 4u + g) * sn.J;
  const __amdgpu_buffer_rsrc_t rmk = __builtin_amdgcn_make_buffer_rsrc((void*)cand_mask, 0, MASKED ? (int)((size_t)n_reqs * sn.J * 8u) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void*)ix.keys, 0, (int)((ix.slots + 2u) * 8u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsn = __builtin_amdgcn_make_buffer_rsrc((void*)sn.blob, 0, (int)sn.blob_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void*)reqs, 0, (int)(n_reqs * stride), 0x00020000);
  // (the set table sits right behind the slots' lists: line ix.slots + 4 + i; the kernel reads ONLY set lines -- the identity of a hit's pod
  // set comes with its bucket line, kSidSets)
  const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc((void*)ix.lists, 0, (int)((ix.slots + 4u + ix.sets_mask + 1u) * 64u), 0x00020000);
  const uint32_t set_line0 = ix.slots + 4u;
  const uint32_t hwords = (stride - 8u) / 8u;                        // hash words per request row (>= 1: the host checks)
  const uint32_t hw0 = hwords < kKeysPerProbe ? hwords : kKeysPerProbe;
  // byte offsets inside a row of the two hashes this lane loads (clamped into the row; unused keys are masked by nb)
  const uint32_t hoff0 = (1u + (k < hw0 ? k : hw0 - 1u)) * 8u, hoff1 = (1u + (16u + k < hw0 ? 16u + k : hw0 - 1u)) * 8u;
  const uint32_t bp_addr = (gsh + q) * 4u;                           // ds_bpermute address of lane (g, k = q): key 4i + q sits 4 (i & 3) lanes on
  const uint32_t bmask = (ix.slots / kBucket) - 1u;
  uint32_t n_def = 0;                                                // requests this wavefront deferred (wave-uniform)
  uint32_t acc_hits = 0, acc_look = 0;                               // probe statistics (lane 0 of each row)
  uint32_t* my_list = defer_list + (size_t)gwave * defer_cap;
  // MASKED single picks of a launched kernel: the rows that need their own QUEUE normalisers are scored by this wavefront when its loop is
  // over (quad_exact_rows) -- their request indices and, per lane of the row, the listed pods and matched counts are parked behind the work
  // lists: [n_segs][cap] indices, then [n_segs][cap][16] lanes x 8 bytes {pA | pB << 16, cntA | cntB << 8 | listed-candidate bits << 16}
  uint32_t n_x = 0;                                                  // rows parked (wave-uniform)
  // (computed where they are needed -- rarely -- from my_list: two pointers less to keep through the loop)
  auto my_xr = [&]() -> uint32_t* { return my_list + (size_t)nwaves * defer_cap; };
  auto my_xs = [&]() -> uint32_t* { return my_list + (size_t)defer_cap * (2u * (size_t)nwaves + 31u * (size_t)gwave); };
#if EPPK_QUAD_PREFETCH > 0
  uint32_t pf_sink = 0, pf_prev = 0;                                 // landing registers of the row prefetches (never read)
  const uint32_t pf_lim = n_reqs * stride - 4u;
  const uint32_t pf_lane = (uint32_t)lane * 64u < 4u * stride ? (uint32_t)lane * 64u : 4u * stride - 4u;
#endif

  struct Row { uint64_t hdr, h0, h1; };                              // landing registers of a block's request rows (lane k: hashes k, 16 + k)
  struct Probe {                                                     // the key gather of a block, quad-transposed: step i = key 4i + q
    uint32_t hlo[8], hhi[8], bkt[8];                                 //   the key and its home bucket
    u32x4_t w[8];                                                    //   landing registers: words 2j, 2j + 1 of that bucket
  };
  auto issue_row = [&](uint32_t blk, Row& r_) {
    uint32_t r = (blk < nblk ? blk : nblk - 1u) * 4u + g;
    r = r < n_reqs ? r : n_reqs - 1u;
    const uint32_t roff = r * stride;
    r_.hdr = buffer_load_u64(rq, roff, 0u);
    r_.h0 = buffer_load_u64(rq, roff + hoff0, 0u);
    r_.h1 = buffer_load_u64(rq, roff + hoff1, 0u);
  };
  // hash + home bucket of step i's key from the lanes that loaded them, then the 16-byte piece of that bucket
  // (`first_only`: every quad of the row fetches the step's FIRST key -- key 4i, quad 0's -- instead of its own: the same line four times, one L2 request)
  auto fetch_step = [&](const Row& r_, uint32_t b0, uint32_t b1, Probe& pb, auto ic, bool first_only = false) {
    constexpr int i = decltype(ic)::value;
    const int a = (int)((first_only ? gsh * 4u : bp_addr) + 16u * (uint32_t)(i & 3));
    pb.hlo[i] = (uint32_t)__builtin_amdgcn_ds_bpermute(a, (int)(uint32_t)(i < 4 ? r_.h0 : r_.h1));
    pb.hhi[i] = (uint32_t)__builtin_amdgcn_ds_bpermute(a, (int)(uint32_t)((i < 4 ? r_.h0 : r_.h1) >> 32));
    pb.bkt[i] = (uint32_t)__builtin_amdgcn_ds_bpermute(a, (int)(i < 4 ? b0 : b1));
  };
  // The pipelined gather covers the first 17 keys -- steps 0..3 in full, of step 4 only key 16 (quad 0's) -- : the walk ends at the first
  // miss, and a request whose first 17 blocks are all cached is the exception: it fetches keys 17..31 on demand (one more round trip)
  // instead of every request paying bucket lines it never looks at.  (Rounds 2-5 gathered 20 ahead: three lines per request of a
  // workload whose shared prefixes are 16 blocks long, 13 % of a cold index's HBM traffic.)
  constexpr int kAhead = 5;
  constexpr uint32_t kAheadKeys = 17u;
  auto issue_keys = [&](const Row& r_, Probe& pb) {
    const uint32_t b0 = home_bucket(r_.h0, ix.shift), b1 = home_bucket(r_.h1, ix.shift);
    fetch_step(r_, b0, b1, pb, std::integral_constant<int, 0>{});
    fetch_step(r_, b0, b1, pb, std::integral_constant<int, 1>{});
    fetch_step(r_, b0, b1, pb, std::integral_constant<int, 2>{});
    fetch_step(r_, b0, b1, pb, std::integral_constant<int, 3>{});
    fetch_step(r_, b0, b1, pb, std::integral_constant<int, 4>{}, /*first_only*/ true);
#ifdef EPPK_DBGQ_NO_BKT     // timing experiment only (wrong results): no key-bucket loads, every key a miss
#pragma unroll
    for (int i = 0; i < kAhead; ++i) pb.w[i] = (u32x4_t)(0u);
    return;
#endif
#pragma unroll
    for (int i = 0; i < kAhead; ++i) pb.w[i] = __builtin_amdgcn_raw_buffer_load_b128(rk, (int)(pb.bkt[i] * (kBucket * 8u) + j16), 0, 0);
  };
  // a key absent from its overflowed home bucket: the following buckets (rare; plain loads)
  auto walk = [&](uint64_t h, uint32_t b) -> uint32_t {
#pragma unroll 1
    for (uint32_t n = 0; n < bmask; ++n) {
      b = (b + 1u) & bmask;
      const uint64_t* kb = ix.keys + (size_t)b * kBucket;
#pragma unroll 1
      for (uint32_t i = kKeySub0; i < kBucket; ++i)
        if (kb[i] == h) return b * kBucket + i;
      if (!(kb[0] & 1ull)) break;
    }
    return kNotFound;
  };
  // {hi, lo} LoRA tier lane words of pod p for adapter row `arow`: ONE load out of the interleaved planes
  auto load_tier_pair = [&](uint32_t arow, uint32_t p, LW& th, LW& tl_) {
    const uint32_t woff = (arow * 64u + (p & 63u)) * 2u * (uint32_t)sizeof(LW);
    if constexpr (sizeof(LW) == 8) {
      const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rsn, (int)woff, (int)SnapOff<LW>::thl, 0);
      th = u64_of(v.x, v.y); tl_ = u64_of(v.z, v.w);
    } else if constexpr (sizeof(LW) == 4) {
      const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(rsn, (int)woff, (int)SnapOff<LW>::thl, 0);
      th = v.x; tl_ = v.y;
    } else {
      const uint32_t v = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsn, (int)woff, (int)SnapOff<LW>::thl, 0);
      th = (LW)(v & 0xFFFFu); tl_ = (LW)(v >> 16);
    }
  };
  auto row16 = [&](unsigned long long m_) -> uint32_t { return (uint32_t)(m_ >> gsh) & 0xFFFFu; };   // this row's slice of a wavefront mask
  // v | (v of another lane): DPP controls quad_perm [1,0,3,2] = 0xB1, [2,3,0,1] = 0x4E, row_ror:4 = 0x124, row_ror:8 = 0x128
  auto or_dpp = [](uint32_t v, auto ctrl) { return v | (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, decltype(ctrl)::value, 0xf, 0xf, true); };

  // Block `blk`: its rows in `cur`, its key gather in `pb`; issues the key gather of blk + nwaves (rows in `nxt`, into `pb`) and
  // the rows of blk + 2 nwaves (into `cur`).
  auto process = [&](uint32_t blk, Row& cur, Row& nxt, Probe& pb) {
    const uint32_t r = blk * 4u + g;
    const bool live = r < n_reqs;
    int32_t adapter = (int32_t)(uint32_t)cur.hdr;
    uint32_t nb = (uint32_t)(cur.hdr >> 32);
    const bool badh = nb > hwords || adapter < -1 || adapter >= (int32_t)EPPK_MAX_ADAPTERS;   // pick_fast_kernel reports the row
    if (badh) { adapter = -1; nb = 0u; }
    const uint32_t arow = (HAS_L && adapter >= 0) ? (uint32_t)adapter : 128u;
    const uint32_t nbc = nb < kKeysPerProbe ? nb : kKeysPerProbe;
    // reserved hashes 0 / ~0 among the probed keys: deferred (lane k looks at keys k and 16 + k)
    const bool rsv = (k < nbc && (cur.h0 + 1ull) <= 1ull) || (16u + k < nbc && (cur.h1 + 1ull) <= 1ull);
#if !EPPK_QUAD_PIPE_KEYS
    issue_keys(cur, pb);      // (not pipelined: gathered and consumed right here; fewer live registers, more wavefronts per SIMD)
#endif
    // ---- finish the probe: found / absent and the SET ID of key 4i + q in all four lanes of its quad
    // The pieces of a bucket (kBucket): j = 0 {flags, meta 3, meta 4, meta 5}, j = 1 {meta 6, meta 7, key 3}, j = 2 {key 4, key 5},
    // j = 3 {key 6, key 7}.  Per step a lane produces a 4-bit CODE: 0 = its piece does not hold the key, else 8 | u, u = the key's word
    // index in the bucket - 2 (1..5) = the DWORD of the bucket that holds its meta: bits 0..1 of u = the component of that dword in its
    // lane's piece, bit 2 = the lane (j = 0 / 1).  The codes of all steps are packed into one register and OR-reduced over the quad ONCE
    // (two DPP ops for the whole probe instead of two per step); found = bit 3.
    uint32_t sid[8];
    uint32_t codes = 0u, hdr = 0u;
    const uint32_t codeA = 8u | (2u * j - 2u), codeB = 8u | (2u * j - 1u);   // the codes of this lane's first / second word (j >= 2 / j >= 1)
    auto match_step = [&](auto ic, uint32_t& cd) {
      constexpr int i = decltype(ic)::value;
      const uint64_t h = u64_of(pb.hlo[i], pb.hhi[i]);
      const bool c0 = j >= 2u && u64_of(pb.w[i].x, pb.w[i].y) == h;  // (words 0..2 of a bucket are flags and meta dwords)
      const bool c1 = j >= 1u && u64_of(pb.w[i].z, pb.w[i].w) == h;
      const uint32_t code = c1 ? codeB : (c0 ? codeA : 0u);
      cd |= code << (4 * i);
      hdr |= pb.w[i].x;                                               // flags bit 0 ("a key of this bucket lives further on"): lanes j == 0
    };
    auto quad_or = [&](uint32_t v) {
      v = or_dpp(v, std::integral_constant<int, 0xB1>{});
      return or_dpp(v, std::integral_constant<int, 0x4E>{});
    };
    auto sid_of = [&](auto ic) {                                      // (unspecified where the key is absent: only hits are used)
      constexpr int i = decltype(ic)::value;
      // The key's meta dword sits in the piece of lane j = 0 (bucket dwords 1..3) or j = 1 (dwords 4, 5): dword u of the bucket, u = the low
      // three bits of the code.  Through the LDS: lanes 0 and 1 park their pieces, every lane of the quad reads dword u back -- two LDS
      // operations and two vector instructions per step.  (In registers it took a tree of byte permutes with selectors made from the
      // code's bits, two quad broadcasts and a final select: 14 vector instructions per step, the largest single item of a kernel that is
      // bound by vector issue: profiles/r06_quad_valu.txt.)
      // (no fence between them: the LDS operations of ONE wavefront execute in issue order, the compiler keeps accesses that may alias in
      //  program order, and nobody else touches this wavefront's 512 bytes; the read's result is waited for where it is first used)
      if (j < 2u) *(u32x4_t*)(s_xbar + j16) = pb.w[i];
      sid[i] = *(const uint32_t*)(s_xbar + ((codes >> (4 * i)) & 7u) * 4u);   // (the whole meta dword: its top byte -- the stamp tag -- is masked off where ids are compared)
    };
    // found bits of this quad's keys (bit 4i = step i) -> bit 4i + q = key 4i + q -> the whole row's keys in every lane
    auto row_found = [&](uint32_t cd) {
      uint32_t w_ = (cd >> 3) & 0x11111111u;
      w_ <<= q;
      w_ = or_dpp(w_, std::integral_constant<int, 0x124>{});
      w_ = or_dpp(w_, std::integral_constant<int, 0x128>{});
      return w_ & (nbc >= 32u ? 0xFFFFFFFFu : ((1u << nbc) - 1u));
    };
    match_step(std::integral_constant<int, 0>{}, codes);
    match_step(std::integral_constant<int, 1>{}, codes);
    match_step(std::integral_constant<int, 2>{}, codes);
    match_step(std::integral_constant<int, 3>{}, codes);
    match_step(std::integral_constant<int, 4>{}, codes);
    codes &= q == 0u ? 0xFFFFFFFFu : 0xFFF0FFFFu;                     // (step 4 was gathered for key 16 alone: the other quads looked at ITS bucket)
    codes = quad_or(codes);
    sid_of(std::integral_constant<int, 0>{});
    sid_of(std::integral_constant<int, 1>{});
    sid_of(std::integral_constant<int, 2>{});
    sid_of(std::integral_constant<int, 3>{});
    sid[4] = sid[5] = sid[6] = sid[7] = 0u;
    uint32_t W = row_found(codes);
    uint32_t m = (uint32_t)__builtin_ctzll(~(unsigned long long)W);   // leading hits of this row's request (<= nbc)
    // (step 4 -- keys 16..19 -- gives its ids only where some row of the wavefront has more than 16 hits: not in a batch of new requests
    //  behind a 16-block prefix, the BASELINE workload)
    bool have4 = __any(m > 16u);
    if (have4) sid_of(std::integral_constant<int, 4>{});
    bool all8 = false;
    uint32_t Wx = 0u;                                                 // hits found further on in a bucket chain (walk below): not in `codes`
    // Two rare steps, in a loop because either can make the other necessary: (1) a row whose first 20 keys are all hits needs steps 5..7;
    // (2) a row whose first missing key sits in an overflowed bucket follows the chain -- and may arrive at 20 hits that way.  (Until round 6
    // step (1) ran once, ahead of (2): a returning request with a displaced key among its first 20, in a wavefront whose other rows
    // stopped short of 20, was scored with 20 matched blocks instead of 32 -- right pick, low score; found by the `revisit` bench leg.)
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
      const bool more = !all8 && __any(m == kAheadKeys && nbc > kAheadKeys);
      if (pass == 1 && !more) break;
      if (__builtin_expect(more, 0)) {   // keys 17..31 on demand: step 4 once more, for every quad's own key, and steps 5..7
        const uint32_t b1 = home_bucket(cur.h1, ix.shift);
        fetch_step(cur, 0u, b1, pb, std::integral_constant<int, 4>{});
        fetch_step(cur, 0u, b1, pb, std::integral_constant<int, 5>{});
        fetch_step(cur, 0u, b1, pb, std::integral_constant<int, 6>{});
        fetch_step(cur, 0u, b1, pb, std::integral_constant<int, 7>{});
#pragma unroll
        for (int i = kAhead - 1; i < 8; ++i) pb.w[i] = __builtin_amdgcn_raw_buffer_load_b128(rk, (int)(pb.bkt[i] * (kBucket * 8u) + j16), 0, 0);
        uint32_t cd2 = 0u;
        match_step(std::integral_constant<int, 4>{}, cd2);
        match_step(std::integral_constant<int, 5>{}, cd2);
        match_step(std::integral_constant<int, 6>{}, cd2);
        match_step(std::integral_constant<int, 7>{}, cd2);
        codes = (codes & 0x0000FFFFu) | quad_or(cd2);
        const uint32_t keep4 = sid[4];                                // (key 16 may be a hit the chain walk of the first pass found: not in `codes`)
        sid_of(std::integral_constant<int, 4>{});
        if ((Wx >> (16u + q)) & 1u) sid[4] = keep4;
        have4 = true;
        sid_of(std::integral_constant<int, 5>{});
        sid_of(std::integral_constant<int, 6>{});
        sid_of(std::integral_constant<int, 7>{});
        W = row_found(codes) | Wx;
        m = (uint32_t)__builtin_ctzll(~(unsigned long long)W);
        all8 = true;
      }
      if (__builtin_expect(__any(j == 0u && (hdr & 1u)), 0)) {
        // The first missing key sits in an OVERFLOWED bucket: it may live in a later bucket (rare: 0.06 % of the buckets at the
        // recommended sizing, but a displaced key of a popular prefix is looked up by every request of its group).  Lane (q = m & 3,
        // j = 0) walks the chain; a key found there is a hit like any other: its set id goes to its quad, its bit into W, and the
        // next first miss is examined in turn.
        uint32_t ovf = 0u;
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (i < kAhead || all8) ovf |= (pb.w[i].x & 1u) << i;
        bool pend = m < nbc && q == (m & 3u) && j == 0u && ((ovf >> (m >> 2)) & 1u);
        while (__any(pend)) {
          uint32_t sf = 0u;                                                // 0 = absent, else 0x80000000 | the set id of the key found further on
          if (pend) {
            const uint64_t hh = *(const uint64_t*)(reqs + (size_t)(r < n_reqs ? r : n_reqs - 1u) * stride + 8u + (size_t)m * 8u);
            const uint32_t s = walk(hh, home_bucket(hh, ix.shift));
            if (s != kNotFound) sf = 0x80000000u | (((const uint32_t*)ix.keys)[meta_dword(s)] & kSidMask);
          }
          sf = quad_or(sf);                                               // to the four lanes of the key's quad
#pragma unroll
          for (int i = 0; i < 8; ++i)
            if (sf != 0u && (m >> 2) == (uint32_t)i) sid[i] = sf & kSidMask;
          uint32_t add = sf != 0u ? 1u << (m & 31u) : 0u;                  // to the whole row
          add = or_dpp(add, std::integral_constant<int, 0x124>{});
          add = or_dpp(add, std::integral_constant<int, 0x128>{});
          Wx |= add;
          W |= add;
          m = (uint32_t)__builtin_ctzll(~(unsigned long long)W);
          pend = add != 0u && m < nbc && q == (m & 3u) && j == 0u && ((ovf >> (m >> 2)) & 1u);
        }
      }
    }
    if (__builtin_expect(!have4 && __any(m > 16u), 0)) {             // a chain walk took a row past 16 hits: step 4's ids after all (not over the walk's own)
      const uint32_t keep = sid[4];
      sid_of(std::integral_constant<int, 4>{});
      if ((Wx >> (16u + q)) & 1u) sid[4] = keep;
      have4 = true;
    }
    unsigned long long badm = __ballot(badh || rsv || (m == kKeysPerProbe && nb > kKeysPerProbe));
    // ---- the pod sets of the hits, from their SET IDS (no line is fetched to compare them)
    // Common case: every hit carries the same id -- the blocks of a shared prefix are cached together -- and ONE set line serves the
    // request.  Second case, the prefix scorer's own (0602-.../README.md:101-112): a request that COMES BACK finds its shared blocks on
    // the group's pods (id A) and its own tail blocks on the one pod it was routed to (a single-pod id b): two ids, the second of which
    // needs no line at all.  matched[p] = cA [p in A] + cb [p == b].  Anything else -- three sets, two lists, an id-less hit -- is deferred.
    const bool any_hit = m > 0u;
    uint32_t sid0 = sid[0];
    {   // the id of hit 0: quad 0's step 0, broadcast along the row
      uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp((int)sid0, (int)sid0, 0x114, 0xf, 0xE, false);   // row_shr:4 into quads 1..3
      t = (uint32_t)__builtin_amdgcn_update_dpp((int)t, (int)t, 0x118, 0xf, 0xC, false);                 // row_shr:8 into quads 2..3
      sid0 = q == 0u ? sid0 : t;
    }
    sid0 = any_hit ? sid0 & kSidMask : 0u;
    // (integer arithmetic in vector registers throughout: booleans here would live in scalar register pairs, of which the kernel has none to spare)
    uint32_t dif = 0u;
#pragma unroll
    for (int i = 0; i < 4; ++i) dif |= 4u * (uint32_t)i + q < m ? sid[i] ^ sid0 : 0u;   // step i's key of this quad is one of the m leading hits, else: no opinion
    if (have4) dif |= 16u + q < m ? sid[4] ^ sid0 : 0u;               // (wave-uniform)
    if (all8) {                                                       // (wave-uniform: steps 5..7 were fetched)
#pragma unroll
      for (int i = kAhead; i < 8; ++i) dif |= 4u * (uint32_t)i + q < m ? sid[i] ^ sid0 : 0u;
    }
    dif &= kSidMask;                                                  // (the top byte of a meta dword is its stamp tag)
    uint32_t sidA = sid0, pod_b = kNoPod, cb = 0u;                    // the listed set; the single pod of the second set and how many hits name it
    uint32_t bad2 = sid0 == kSidNone ? 1u : 0u;                       // hit 0 without an id (a later one: the second id below, kSidNone among them)
    if (__builtin_expect(__any(dif != 0u), 0)) {
      // the second id of the row: every differing id must be the same one (row minimum == row maximum)
      uint32_t lmin = 0xFFFFFFFFu, lmax = 0u;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const uint32_t x = 4u * (uint32_t)i + q < m ? sid[i] & kSidMask : sid0;
        lmin = (x != sid0 && x < lmin) ? x : lmin;
        lmax = (x != sid0 && x > lmax) ? x : lmax;
      }
      lmin = dpp_min_u32<0xB1, 0xf>(lmin); lmin = dpp_min_u32<0x4E, 0xf>(lmin); lmin = dpp_min_u32<0x141, 0xf>(lmin); lmin = dpp_min_u32<0x140, 0xf>(lmin);
      lmax = ~lmax;                                                    // (maximum = complement of the minimum of the complements)
      lmax = dpp_min_u32<0xB1, 0xf>(lmax); lmax = dpp_min_u32<0x4E, 0xf>(lmax); lmax = dpp_min_u32<0x141, 0xf>(lmax); lmax = dpp_min_u32<0x140, 0xf>(lmax);
      lmax = ~lmax;
      uint32_t cx = 0u;                                                // hits of the row that carry the second id
#pragma unroll
      for (int i = 0; i < 8; ++i) cx += (uint32_t)__builtin_popcount(row16(__ballot(j == 0u && 4u * (uint32_t)i + q < m && (sid[i] & kSidMask) == lmin)));
      if (lmin != 0xFFFFFFFFu) {                                       // this row has a second id
        bad2 |= (lmin != lmax || lmin == kSidNone || (sid0 >= kSidSets && lmin >= kSidSets)) ? 1u : 0u;      // three sets, a hit without an id, or two lists
        if (lmin < kSidSets) { pod_b = lmin; cb = cx; }                         // {A or a single pod} + single pod b
        else { sidA = lmin; pod_b = sid0; cb = m - cx; }                        // hit 0 names the single pod, the others the list
      }
    }
    badm |= __ballot(any_hit && bad2 != 0u);
    const uint32_t cA = m - cb;
    // ---- the ONE set line of the request (a single-pod id needs none: the line is made up in registers)
    const bool lineA = sidA >= kSidSets && sidA != kSidNone;
    u32x4_t L0;
#ifdef EPPK_DBGQ_NO_LISTS   // timing experiment only (wrong results): no list loads, six pseudo pods per piece
    L0.x = 0x00020001u + (sidA & 1u); L0.y = 0x00040003u; L0.z = 0xFFFFFFFFu; L0.w = 8u;
#else
    L0 = __builtin_amdgcn_raw_buffer_load_b128(rl, (int)((set_line0 + (lineA ? sidA - kSidSets : 0u)) * 64u + j16), 0, 0);
#endif
#ifdef EPPK_DBGQ_NO_TOP     // timing experiment only (wrong results): no top-table loads
    const double top_t = -1.0 - (double)arow;
    const uint32_t top_p = k + arow;
#else
    const u32x4_t te = __builtin_amdgcn_raw_buffer_load_b128(rsn, (int)(arow * 256u + k * 16u), (int)SnapOff<LW>::top16, 0);   // entry k: {T, pod}
    double top_t = __hiloint2double((int)te.y, (int)te.x);            // (TOPK: a row whose pool runs dry loads the table's next 16 entries into these)
    uint32_t top_p = te.z;
#endif
    u32x4_t mk0 = (u32x4_t)(0u), mk1 = (u32x4_t)(0u);
    if (MASKED) {     // words 4k .. 4k+3 of the request's candidate row (a read past the last row returns zeros: buffer range)
      const uint32_t mo = ((r < n_reqs ? r : n_reqs - 1u) * sn.J + 4u * k) * 8u;
      mk0 = __builtin_amdgcn_raw_buffer_load_b128(rmk, (int)mo, 0, 0);
      mk1 = __builtin_amdgcn_raw_buffer_load_b128(rmk, (int)(mo + 16u), 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- the listed pods (as soon as the set line is there): lane (q, j) takes id 4q + j, and id 16 + 4q + j when the list is that long
    if (!lineA) { L0.x = k == 0u ? (0xFFFF0000u | (sidA & 0xFFFFu)) : 0xFFFFFFFFu; L0.y = 0xFFFFFFFFu; L0.z = 0xFFFFFFFFu; }    // {the one pod}: id 0 = piece 0, low half of dword 0
    const uint32_t idsh = 16u * (q & 1u);
    const uint32_t idA = (((q & 2u) ? L0.y : L0.x) >> idsh) & 0xFFFFu;
    uint32_t idB = q < 2u ? (L0.z >> idsh) & 0xFFFFu : kListNone;
    // the single pod b of a second set: where it is on the list its lane counts cA + cb hits, else the row's last lane takes it as a
    // candidate of its own (ids 16 + 4q + j exist for q < 2 only: lane 15's second id is free)
    const bool b_listed = row16(__ballot(idA == pod_b || idB == pod_b)) != 0u;      // (pod_b = kNoPod: never)
    if (k == 15u && pod_b != kNoPod && !b_listed) idB = pod_b;
    bool lsA = any_hit && idA < sn.n_pods, lsB = any_hit && idB < sn.n_pods;
    const uint32_t pA = lsA ? idA : 0u, pB = lsB ? idB : 0u;
    const uint32_t cntA = idA == pod_b ? cA + cb : cA;
    const uint32_t cntB = idB == pod_b ? (b_listed ? cA + cb : cb) : cA;
    LW thA = 0, tlA = 0;
#ifndef EPPK_DBGQ_NO_TIER   // (defined: timing experiment only, wrong results: no tier-word loads)
    if (HAS_L && lsA) load_tier_pair(arow, pA, thA, tlA);             // (lanes without a listed pod stay out of the gather)
#endif
    __builtin_amdgcn_sched_barrier(0);
    // ---- next stages, queued BEHIND everything this block still waits for: key gather of the next block, rows of the one
    //      after, L2 prefetch of a later one
#if EPPK_QUAD_PIPE_KEYS
    if (!RESIDENT || blk + nwaves < nblk) issue_keys(nxt, pb);
#endif
    if (!RESIDENT || blk + 2u * nwaves < nblk) issue_row(blk + 2u * nwaves, cur);
#if EPPK_QUAD_PREFETCH > 0
    {   // one lane per 64-byte sector of the four (contiguous) rows; the value is never used.  Branch-free (past the end: the
        // last block again; lanes beyond the rows: their last sector again): a conditional landing register would need a copy,
        // and a copy of a landing register waits for every load in flight
      const uint32_t bp = blk + (uint32_t)EPPK_QUAD_PREFETCH * nwaves;
      const uint32_t off = (bp < nblk ? bp : nblk - 1u) * 4u * stride + pf_lane;
      pf_sink ^= pf_prev;                   // (consumes the PREVIOUS iteration's prefetch, long landed: keeps every load alive without a wait)
      pf_prev = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rq, (int)(off < pf_lim ? off : pf_lim), 0, 0);
    }
#endif
    __builtin_amdgcn_sched_barrier(0);
    bool no_cand = false;
    // PARK (MASKED): a row that cannot be scored from base[] and the top table ALONE -- its candidates miss
    // a snapshot-wide QUEUE extreme, or none of the table's 64 entries is a candidate outside its list -- is not deferred: every
    // candidate is evaluated in full by quad_exact_rows when the loop is over.  softm = those rows.
    constexpr bool PARK = MASKED && EPPK_QUAD_PARK != 0;
    unsigned long long softm = 0ull;
    uint32_t xe1 = 0u;                                                // this lane's matched counts and listed-candidate bits, for quad_exact_rows
    if (MASKED) {
      // this lane's four candidate words: mask & active (words beyond J are no pods)
      uint64_t cw[4] = {u64_of(mk0.x, mk0.y), u64_of(mk0.z, mk0.w), u64_of(mk1.x, mk1.y), u64_of(mk1.z, mk1.w)};
      bool anyc = false, hmin = false, hmax = false;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint32_t w = 4u * k + (uint32_t)i;
        cw[i] = w < sn.J ? cw[i] & s_nat[w] : 0ull;
        anyc = anyc || cw[i] != 0ull;
        hmin = hmin || (cw[i] & s_nat[64u + (w & 63u)]) != 0ull;
        hmax = hmax || (cw[i] & s_nat[128u + (w & 63u)]) != 0ull;
        if (w < sn.J) s_cn[w] = cw[i];
      }
      no_cand = row16(__ballot(anyc)) == 0u;                          // fail closed: EPPK_NO_PICK (stored below)
      // base[] and the top tables embed the snapshot-wide QUEUE normalisers: they apply iff the candidates contain a pod at the
      // minimum and one at the maximum queue depth; otherwise the request's own normalisers are needed (exact evaluation: deferred)
      unsigned long long exm = 0ull;
      if (sn.lead_queue) exm = __ballot(!no_cand && (row16(__ballot(hmin)) == 0u || row16(__ballot(hmax)) == 0u));
      if constexpr (PARK) softm = exm; else badm |= exm;
      wave_lds_fence();
      auto is_cand = [&](uint32_t p_) -> bool { return (s_cn[p_ >> 6] >> (p_ & 63u)) & 1ull; };
      lsA = lsA && is_cand(pA);
      lsB = lsB && is_cand(pB);
      // (one register to the end of the block; parked in the wavefront's LDS scratch instead it cost the loop 1 us per 64k batch more)
      //  bits 0-5 / 6-11 matched counts (<= 32), 12 / 13 "is a listed candidate", 14-21 the adapter's table row, 22-28 the request's blocks (< 64)
      if constexpr (PARK) xe1 = cntA | (cntB << 6) | ((lsA ? 1u : 0u) << 12) | ((lsB ? 1u : 0u) << 13) | (arow << 14) | (nb << 22);
    }
    // ---- evaluate: binary64 adds in chain order (pick_fast_kernel: pod_total)
    // (the prefix term of a listed pod: clamp01(matched / n) * w for matched = the hits whose set holds it -- m for every listed pod of the
    //  common case; cA, cb or cA + cb where a second, single-pod set is in play)
    const uint32_t prow = nb * sn.pterm_ld;
    auto total_of = [&](uint32_t p, LW th, LW tl_, uint32_t cnt) -> double {
      double lterm = 0.0;
      if (HAS_L) {
        const uint32_t jb = p >> 6;
        lterm = s_lw[(uint32_t)(((th >> jb) & 1) << 1) | (uint32_t)((tl_ >> jb) & 1)];
      }
      return eval_total<HAS_L, true, P_FIRST>(s_base[p], lterm, s_pterm[prow + cnt]);
    };
    const double tA = total_of(pA, thA, tlA, cntA);
    double best = lsA ? tA : -__builtin_inf();
    uint32_t bidx = lsA ? pA : kNoPod;
    double tB_keep = -__builtin_inf();                                // (TOPK: the second listed pod of the lane stays a candidate of its own)
    const bool anyB = __any(lsB);
    if (anyB) {                                                       // a list of more than 16 pods
      LW thB = 0, tlB = 0;
      if (HAS_L) load_tier_pair(arow, pB, thB, tlB);
      const double tB = total_of(pB, thB, tlB, cntB);
      tB_keep = tB;
      if (lsB && (tB > best || (tB == best && pB < bidx))) { best = tB; bidx = pB; }
    }
    // ---- "listed" bitmap of the row (LDS): set, look the table entries up, clear
    if (lsA) atomicOr(&bits[pA >> 5], 1u << (pA & 31u));
    if (anyB && lsB) atomicOr(&bits[pB >> 5], 1u << (pB & 31u));
    wave_lds_fence();
    const bool tpv = top_p != kNoPod;
    const uint32_t tq = tpv ? top_p : 0u;
    bool tok = tpv && !((bits[tq >> 5] >> (tq & 31u)) & 1u);         // entry k exists and is not listed
    if (MASKED) tok = tok && ((s_cn[tq >> 6] >> (tq & 63u)) & 1ull);  // ... and is a candidate of this request
    wave_lds_fence();
    if (!TOPK && !MASKED) {                                           // (TOPK, and MASKED single picks, keep the bitmap a little longer: refills look entries up)
      if (lsA) bits[pA >> 5] = 0u;
      if (anyB && lsB) bits[pB >> 5] = 0u;
    }
    uint32_t tvr = row16(__ballot(tpv));
    if constexpr (TOPK) {
      // ---- ordered fallbacks: topk rounds of a row argmax over every lane's best remaining candidate
      bool vA = lsA, vB = lsB, vT = tok;
      const uint32_t tk = topk;
      uint32_t tbase = 0u;                                            // first table entry of the window this row's lanes hold
      for (uint32_t round = 0; round < tk; ++round) {
        // The table's pool is empty but the table goes on (with candidate masks half of a window's entries are no candidates, and every
        // round may take one): the row loads the NEXT 16 entries (of 64) and looks them up in the listed bitmap and the candidate row --
        // a request used to be deferred to the dense route for that, 20-odd us each.  Beyond entry 64: deferred after all.
        for (int refill = 0; refill < 3; ++refill) {
          const bool dry = row16(__ballot(vT)) == 0u && tvr == 0xFFFFu && !no_cand && tbase < 48u;   // (the table holds 64 entries)
          if (!__any(dry)) break;
          if (dry) {
            tbase += 16u;
            top_t = __longlong_as_double((long long)buffer_load_u64(rsn, (tbase + k) * 8u, SnapOff<LW>::topv + arow * 512u));
            top_p = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsn, (int)((tbase + k) * 4u), (int)(SnapOff<LW>::topi + arow * 256u), 0);
            const bool ex = top_p != kNoPod;
            const uint32_t tq2 = ex ? top_p : 0u;
            vT = ex && !((bits[tq2 >> 5] >> (tq2 & 31u)) & 1u);
            if (MASKED) vT = vT && ((s_cn[tq2 >> 6] >> (tq2 & 63u)) & 1ull);
          }
          const uint32_t tv2 = row16(__ballot(top_p != kNoPod));
          if (dry) tvr = tv2;                                         // (a full window: the table may go on)
        }
        {
          const unsigned long long drym = __ballot(row16(__ballot(vT)) == 0u && tvr == 0xFFFFu && !no_cand && sn.n_pods > tbase + 16u);   // still dry and entries beyond the window exist
          if constexpr (PARK) softm |= drym; else badm |= drym;
        }
        double lb = vA ? tA : -__builtin_inf();
        uint32_t lp = vA ? pA : kNoPod;
        if (vB && (tB_keep > lb || (tB_keep == lb && pB < lp))) { lb = tB_keep; lp = pB; }
        if (vT && (top_t > lb || (top_t == lb && top_p < lp))) { lb = top_t; lp = top_p; }
        double wm = lb;
        wm = vmax_f64(wm, dpp_f64<0xB1, 0xf>(wm));
        wm = vmax_f64(wm, dpp_f64<0x4E, 0xf>(wm));
        wm = vmax_f64(wm, dpp_f64<0x141, 0xf>(wm));
        wm = vmax_f64(wm, dpp_f64<0x140, 0xf>(wm));
        uint32_t wi = lb == wm ? lp : kNoPod;
        wi = dpp_min_u32<0xB1, 0xf>(wi);
        wi = dpp_min_u32<0x4E, 0xf>(wi);
        wi = dpp_min_u32<0x141, 0xf>(wi);
        wi = dpp_min_u32<0x140, 0xf>(wi);
        const bool none = wi == kNoPod || no_cand;
        if (k == 0u && live) {                                        // (stored right away; a row that ends up deferred is rewritten by the work-list pass)
          out_pick[(size_t)r * tk + round] = none ? -1 : (int32_t)wi;
          if (out_score) out_score[(size_t)r * tk + round] = none ? 0.0 : wm;
        }
        if (vA && pA == wi) vA = false;                               // the winner leaves the pool
        if (vB && pB == wi) vB = false;
        if (vT && top_p == wi) vT = false;
      }
      wave_lds_fence();
      if (lsA) bits[pA >> 5] = 0u;
      if (anyB && lsB) bits[pB >> 5] = 0u;
    }
    double wmax = best;
    uint32_t widx = kNoPod;
    bool from_table = false;                                          // (LEARN: the pick is the table's candidate, not a listed pod)
    if constexpr (!TOPK) {
    // ---- argmax over the row: (total desc, pod asc)
    wmax = best;
    wmax = vmax_f64(wmax, dpp_f64<0xB1, 0xf>(wmax));
    wmax = vmax_f64(wmax, dpp_f64<0x4E, 0xf>(wmax));
    wmax = vmax_f64(wmax, dpp_f64<0x141, 0xf>(wmax));
    wmax = vmax_f64(wmax, dpp_f64<0x140, 0xf>(wmax));
    widx = best == wmax ? bidx : kNoPod;
    widx = dpp_min_u32<0xB1, 0xf>(widx);
    widx = dpp_min_u32<0x4E, 0xf>(widx);
    widx = dpp_min_u32<0x141, 0xf>(widx);
    widx = dpp_min_u32<0x140, 0xf>(widx);
    // ---- best pod outside the list: the first table entry that is not listed
    uint32_t okr = row16(__ballot(tok));
    uint32_t e = (uint32_t)__builtin_ctz(okr | 0x10000u);             // 16: none among the 16 entries
    if constexpr (MASKED) {
      // With candidate masks the first 16 entries of the adapter's table often hold no candidate at all (a 1/8-density mask: one request
      // in eight; they used to be deferred to the work-list pass, whose slowest request -- 20-odd us -- then ended the launch): the
      // row walks on through the table's 64 entries, 16 at a time, as the fallback rounds above do (round 5).
      uint32_t tbase = 0u;
      for (int refill = 0; refill < 3; ++refill) {
        const bool dry = e == 16u && tvr == 0xFFFFu && !no_cand && tbase < 48u;
        if (!__any(dry)) break;
        if (dry) {
          tbase += 16u;
          top_t = __longlong_as_double((long long)buffer_load_u64(rsn, (tbase + k) * 8u, SnapOff<LW>::topv + arow * 512u));
          top_p = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsn, (int)((tbase + k) * 4u), (int)(SnapOff<LW>::topi + arow * 256u), 0);
          const bool ex = top_p != kNoPod;
          const uint32_t tq2 = ex ? top_p : 0u;
          tok = ex && !((bits[tq2 >> 5] >> (tq2 & 31u)) & 1u) && ((s_cn[tq2 >> 6] >> (tq2 & 63u)) & 1ull);
        }
        const uint32_t tv2 = row16(__ballot(top_p != kNoPod)), ok2 = row16(__ballot(tok));
        if (dry) { tvr = tv2; okr = ok2; e = (uint32_t)__builtin_ctz(okr | 0x10000u); }
      }
      wave_lds_fence();
      if (lsA) bits[pA >> 5] = 0u;
      if (anyB && lsB) bits[pB >> 5] = 0u;
      const unsigned long long drym = __ballot(e == 16u && tvr == 0xFFFFu && !no_cand && sn.n_pods > tbase + 16u);   // still dry and entries beyond the window exist
      if constexpr (PARK) softm |= drym; else badm |= drym;
    } else {
      badm |= __ballot(e == 16u && tvr == 0xFFFFu && !no_cand);       // all 16 exist and are listed (or no candidates): the rest of the table is needed
    }
    const uint32_t src = gsh + (e & 15u);
    double cand_t = __hiloint2double(__shfl(__double2hiint(top_t), (int)src), __shfl(__double2loint(top_t), (int)src));
    uint32_t cand_p = (uint32_t)__shfl((int)top_p, (int)src);
    if (e == 16u) { cand_t = -__builtin_inf(); cand_p = kNoPod; }
    if (cand_t > wmax || (cand_t == wmax && cand_p < widx)) { wmax = cand_t; widx = cand_p; from_table = true; }
    }
    // ---- store, or defer
    bool xdone = false;                                               // PARK: the row's pick will be stored by quad_exact_rows
    if constexpr (PARK) {
      if (__builtin_expect(softm != 0ull, 0)) {
        const bool want = row16(softm) != 0u && row16(badm) == 0u;    // (a row that is deferred for another reason anyway: the work-list pass sorts it all out)
        const unsigned long long wm = __ballot(want);
        const uint32_t rows = (uint32_t)(wm & 1ull) | (uint32_t)((wm >> 15) & 2ull) | (uint32_t)((wm >> 30) & 4ull) | (uint32_t)((wm >> 45) & 8ull);
        if (want) {
          const uint32_t xi = n_x + (uint32_t)__builtin_popcount(rows & ((1u << g) - 1u));
          uint32_t* e_ = my_xs() + (size_t)xi * 32u + k * 2u;
          e_[0] = pA | (pB << 16);
          e_[1] = xe1;
          if (k == 0u) my_xr()[xi] = r;
        }
        n_x += (uint32_t)__builtin_popcount(rows);
        xdone = want;
        badm |= softm;
      }
    }
    const bool gbad = row16(badm) != 0u;
    const bool lead = k == 0u && live;
    const unsigned long long dm = __ballot(lead && gbad && !xdone);
    if (lead) {      // (one store instruction for pick and score -- lanes 0 / 1 / 2 of a row writing pick / score halves -- was measured: no gain)
      if (MASKED && xdone) {                                          // (stored already; the update that follows a LEARN pick takes the whole path)
        if constexpr (LEARN) learn_out[r] = 0u;
        acc_hits += m;
        acc_look += (m + 1u < nb) ? m + 1u : nb;
      } else if (!gbad) {
        if constexpr (!TOPK) {
          const bool none = widx == kNoPod || no_cand;
          out_pick[r] = none ? -1 : (int32_t)widx;
          if (out_score) out_score[r] = none ? 0.0 : wmax;
          // (bit 31: the pick is in the set of ALL m hits -- every listed pod of the common case; with a second, single-pod set only that
          //  pod, and only where the list holds it too)
          if constexpr (LEARN) learn_out[r] = none ? 0u : (((from_table || m == 0u || (pod_b != kNoPod && !(widx == pod_b && b_listed))) ? 0u : 0x80000000u) | ((widx + 1u) << 8) | m);
        }
        if (!no_cand) {                                                 // (like pick_fast_kernel: a request without candidates is not counted)
          acc_hits += m;
          acc_look += (m + 1u < nb) ? m + 1u : nb;
        }
      } else {
        __hip_atomic_store(&my_list[n_def + (uint32_t)__builtin_popcountll(dm & ((1ull << lane) - 1ull))], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (through to memory: TAIL)
        if constexpr (LEARN) learn_out[r] = 0u;
      }
    }
    n_def += (uint32_t)__builtin_popcountll(dm);
  };

  Row qa, qb;
  Probe pb;
  issue_row(gwave, qa);                                              // the first rows come from HBM: in flight while the tables are staged
  if (!RESIDENT || gwave + nwaves < nblk) issue_row(gwave + nwaves, qb);   // (RESIDENT: rows cross PCIe -- no load that nobody waits for)
  if constexpr (!RESIDENT) {                                         // (RESIDENT: the tables are in LDS already and stay there between doorbells)
#ifndef EPPK_DBGQ_NO_STAGE  // (defined: timing experiment only, wrong results: base[] is not staged)
    for (uint32_t i = threadIdx.x; i < sn.J * 32u; i += blockDim.x) ((double2*)s_base)[i] = ((const double2*)sn.base)[i];   // (16 bytes per load: every
                                                                     // vector memory instruction costs the CU's address unit ~16 clocks)
#endif
    if (threadIdx.x == 0u) { s_lw[0] = tl.lw[0]; s_lw[1] = tl.lw[1]; s_lw[2] = tl.lw[2]; s_lw[3] = tl.lw[3]; }   // (no dynamic index into the argument struct)
    for (uint32_t i = threadIdx.x; i < pwn / 2u; i += blockDim.x) ((double2*)s_pterm)[i] = ((const double2*)sn.pterm)[i];
    if ((pwn & 1u) && threadIdx.x == 0u) s_pterm[pwn - 1u] = sn.pterm[pwn - 1u];
    for (uint32_t i = threadIdx.x; i < (blockDim.x >> 4) * bits_dw; i += blockDim.x) s_bits_all[i] = 0u;
    if (MASKED)
      for (uint32_t i = threadIdx.x; i < 192u; i += blockDim.x) s_nat[i] = sn.nat[i];
    __syncthreads();
  }
  if (idle) {
    if (lane == 0) __hip_atomic_store(&defer_cnt[gwave], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (!idle) {
#if EPPK_QUAD_PIPE_KEYS
    issue_keys(qa, pb);
#endif
    for (uint32_t blk = gwave; blk < nblk; blk += 2u * nwaves) {
      process(blk, qa, qb, pb);
      if (blk + nwaves >= nblk) break;
      process(blk + nwaves, qb, qa, pb);
    }
    if constexpr (MASKED && EPPK_QUAD_PARK != 0) {
      // the parked rows: a function of its own that is handed the kernel's arguments by address (launched: where they lie in the kernarg
      // segment, re-read from there) -- nothing is kept alive through the loop for it but the count
      // (RESIDENT: the caller does that, where its work-list pass is -- the count comes back in the upper half of the result; a call in
      //  here cost a 16-request batch of dense masks 7 us of spills through scratch that the doorbell's acquire had just invalidated)
      if constexpr (!RESIDENT)
        if (__builtin_expect(n_x != 0u, 0)) quad_park_drain_launched<LW, HAS_L, TOPK>(n_x, smem);
    }
#if EPPK_QUAD_PREFETCH > 0
    asm volatile("" ::"v"(pf_sink ^ pf_prev));     // (the prefetch loads must not be dead-code eliminated)
#endif
    if (lane == 0) {
      __hip_atomic_store(&defer_cnt[gwave], n_def, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (n_def) atomicAdd(defer_total, n_def);
    }
  }
  if (stats && gwave < kStatSlots) {
    const uint32_t hs = (uint32_t)__builtin_amdgcn_readlane((int)acc_hits, 0) + (uint32_t)__builtin_amdgcn_readlane((int)acc_hits, 16) +
                        (uint32_t)__builtin_amdgcn_readlane((int)acc_hits, 32) + (uint32_t)__builtin_amdgcn_readlane((int)acc_hits, 48);
    const uint32_t ls = (uint32_t)__builtin_amdgcn_readlane((int)acc_look, 0) + (uint32_t)__builtin_amdgcn_readlane((int)acc_look, 16) +
                        (uint32_t)__builtin_amdgcn_readlane((int)acc_look, 32) + (uint32_t)__builtin_amdgcn_readlane((int)acc_look, 48);
    if (lane == 0 && (hs | ls)) { stats[4 + 2 * gwave] += hs; stats[5 + 2 * gwave] += ls; }
  }
  if constexpr (RESIDENT && MASKED && EPPK_QUAD_PARK != 0) return n_def | (n_x << 16);    // (at most 64 requests per wavefront)
  return n_def;
}

template <typename LW, bool HAS_L, bool P_FIRST, bool MASKED = false, bool TOPK = false, bool TAIL = false, bool LEARN = false>
__global__ __launch_bounds__(EPPK_QUAD_MAX_THREADS, EPPK_QUAD_WAVES) void pick_quad_kernel(KSnap sn, KIndex ix, KTail tl, const uint8_t* __restrict__ reqs,
                                                                 uint32_t stride, uint32_t n_reqs, uint32_t pwn, const uint64_t* __restrict__ cand_mask,
                                                                 int32_t* __restrict__ out_pick, double* __restrict__ out_score,
                                                                 unsigned long long* __restrict__ stats,
                                                                 uint32_t* __restrict__ defer_cnt, uint32_t* __restrict__ defer_list, uint32_t defer_cap,
                                                                 uint32_t* __restrict__ defer_total, uint32_t* __restrict__ defer_total_next, uint32_t topk,
                                                                 uint32_t* __restrict__ done_ctr, uint32_t* __restrict__ report, KChain tail_chain,
                                                                 uint32_t* __restrict__ learn_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (blockIdx.x == 0 && threadIdx.x == 0) *defer_total_next = 0u;   // the counter of this buffer set's NEXT launch (nobody reads it before)
  const uint32_t n_def = pick_quad_body<LW, HAS_L, P_FIRST, MASKED, TOPK, LEARN>(blockIdx.x, gridDim.x, smem, sn, ix, tl, reqs, stride, n_reqs, pwn, cand_mask, out_pick, out_score,
                                                                                 stats, defer_cnt, defer_list, defer_cap, defer_total, topk, learn_out,
                                                                                 quad_xbar_off(sn.J, pwn, blockDim.x >> 6, MASKED));
  (void)tail_chain; (void)done_ctr; (void)report; (void)n_def;
  if constexpr (TAIL) {
    // the end of the one-launch form -- the work-list pass over what THIS workgroup deferred (rarely anything), the arrival counters and
    // the report -- lives in a function that is never inlined: the hot loop above is compiled as if it were not there
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wavefront's part of the work list, its count and its share of the total have
                                                          // reached memory (agent-scope stores and atomics, acknowledged)
    if (__syncthreads_or((int)(n_def != 0u))) quad_tail_pass<LW, HAS_L, P_FIRST, MASKED, TOPK>(smem);
    __syncthreads();
    if (threadIdx.x == 0u) quad_tail_report();
  }
}

// ---- RESIDENT pick kernel: the latency path of small batches (opt-in: EPPK_RESIDENT=1) -------------------------------------------
// What a per-request caller hands over -- pkg/lwepp/handlers/request.go:141-163 calls Pick once per stream message; the reference's
// design point is 10-1000 QPS (docs/proposals/006-scheduler/README.md:133) -- is batches of 1-64 requests, and such a batch is all
// latency: of the 20 us a 16-request batch took host-observed in round 3, 13 were the launch and the completion signal.  This kernel
// is launched ONCE and stays: one workgroup (a wavefront per request, 16 at a time) polls a doorbell word in pinned host memory; the
// host writes the request rows into the context's pinned staging buffer, the count and then the doorbell; the workgroup scores the
// batch with pick_fast_kernel's own body (same picks, same scores), writes picks and scores into the pinned result buffers, and
// raises the completion word the host is polling.  Doorbell -> answer round trip of an empty batch: 2.7 us, with one row read over
// PCIe 3.9 us (scripts/micro/doorbell.hip -> profiles/r04_micro_doorbell.txt).
//   * Nothing cached before the doorbell may be trusted -- the rows were rewritten by the host, the index and the snapshot may have
//     been updated by other kernels, and there is no kernel boundary to do it for us: system-scope acquire (vector caches) plus
//     s_dcache_inv (the request headers and the argument block are read with scalar loads) behind every doorbell; system-scope
//     release in front of the completion word.
//   * The kernel leaves by itself after `max_idle_polls` polls without a doorbell (~50 ms), so that a host that has died, a device-wide
//     synchronise of somebody else, or a caller that simply stopped cannot leave a spinning workgroup behind; the library starts it again
//     with the next small batch.  kResQuit in the doorbell = leave now (eppk_destroy, and in front of every device-wide wait of the
//     library's own).
//   * It holds one CU (16 wavefronts x 128 VGPRs): the persistent pick kernels of the same context size their grids for one CU fewer.
struct alignas(64) ResidentCtl {   // pinned host memory; the two directions in cache lines of their own
  uint32_t bell;                // host -> device: sequence number of the batch to score (monotonic, never 0), or kResQuit
  uint32_t n_reqs;              //                 the batch's request count: the upper half of the SAME 8-byte word (one store, one load)
  uint32_t pad0[14];
  uint32_t done;                // device -> host: sequence number of the last batch whose results are in the pinned buffers
  uint32_t state;               //                 kResRunning while the kernel polls, kResExited when it has left
  uint32_t updated;             //                 LEARN units: sequence number of the last batch whose post-route index update is in the index
  uint32_t pad1[13];
};
constexpr uint32_t kResQuit = 0xFFFFFFFFu, kResRunning = 1u, kResExited = 2u;
// The upper half of the doorbell word: bits 0..7 the batch's request count, 8..11 k (entries per request: 1 = the pick), 12..13 the
// buffer set the rows / masks / results of this batch live in (0 = the context's staging buffers, 1 + s = staging set s of the
// pipelined host path).
constexpr uint32_t kResBufSets = 3u;
__host__ __device__ __forceinline__ constexpr uint32_t res_bell_hi(uint32_t n, uint32_t k, uint32_t bufset) { return n | (k << 8) | (bufset << 12); }
struct ResidentBuf { const uint8_t* reqs; const uint64_t* mask; int32_t* out_pick; double* out_score; };   // pinned host memory as the device addresses it (null: never used)
struct ResidentArgs {           // device memory; rewritten by the host only between two doorbells (the kernel reads it behind each)
  KSnap sn; KIndex ix; KTail tl;
  ResidentBuf buf[kResBufSets];
  KChain chain;                   // the whole weighted chain: the exact evaluation of a MASKED request whose candidates miss a QUEUE extreme (work-list pass)
  uint32_t stride, pwn;
  uint32_t gen, lds_bytes;        // gen changes whenever the block is rewritten (a publish): the workgroup stages the snapshot's tables into LDS again
  uint32_t* defer_cnt; uint32_t* defer_list; uint32_t* defer_total; uint32_t defer_cap, pad;   // QUAD form: the workgroup's work list (one segment per wavefront)
  // LEARN units: the post-route update index[hash[r][i]] U= {pick[r]} is applied by the resident workgroup itself, right behind the answer
  uint8_t* rows_copy;             // device copy of the batch's request rows (the caller may refill the pinned rows as soon as it has the picks)
  uint32_t* learn;                // learn words of the batch (pick_quad_kernel<..., LEARN>)
  uint64_t* keys_w; void* bitmaps_w; uint32_t* lists_w; uint32_t* rstamps; unsigned long long* ixc; uint32_t* status; const void* act;
  uint32_t* sort_wl; uint32_t sort_cap;      // a work list of the unit's own for the lists to canonicalise (SortWl layout)
  uint32_t limit, epoch, max_blocks, max_pods, pad2;
  uint32_t* set_ctl;              // the set table's counters (SetTab::ctl); the table itself: ix.sets_mask + 1 lines behind the slots' lists (lists_w)
};

// MASKED / TOPK (QUAD form only): the variants a dispatcher issues beside plain picks -- a batch with candidate masks (the subset filter,
// pkg/lwepp/handlers/request.go:104-133), ordered fallbacks (PickResult.Fallbacks, handlers/server.go:72-77) -- each a kernel of its own
// behind a doorbell of its own (the library starts the one a call needs; an idle one leaves by itself).
// LEARN (QUAD form, single picks): what eppk_pick_stage_begin(EPPK_PICK_LEARN) issues.  The workgroup first copies the batch's rows
// from pinned host memory into device memory (one PCIe round trip, which the pick body then does not pay again: it reads the copy),
// answers the picks, and then applies the post-route index update ITSELF (resident_learn_update, behind the index maintenance code
// below): budget, one index_insert_one per (block, pick) pair, re-sort of the lists it touched -- the three launches of the
// launched path, without a launch.  The next doorbell of this unit is looked at only behind the update; every other reader or writer of the
// index waits for the `updated` word (eppk.hip: resident_drain).
template <typename LW>
__device__ __noinline__ void resident_learn_update(const ResidentArgs* a, const ResidentBuf& rb, uint32_t n);

template <typename LW, bool HAS_L, bool P_FIRST, bool QUAD, bool MASKED = false, bool TOPK = false, bool LEARN = false>
__global__ __launch_bounds__(EPPK_FAST_MAX_THREADS, 1) void pick_resident_kernel(ResidentCtl* ctl, const ResidentArgs* __restrict__ args, uint32_t seen, unsigned long long max_idle_polls) {
  static_assert(QUAD || !(MASKED || TOPK || LEARN), "the variants exist for the quad form");
  static_assert(!(LEARN && TOPK), "learn words go with single picks");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ uint32_t s_seq, s_n;
  // Everything the doorbell wavefront does alone is behind a WAVE-UNIFORM condition (a scalar branch).  Written as `threadIdx.x == 0`
  // the compiler rotated the loop so that thread 0's part -- completion store, then the polling -- became an outer loop around an inner
  // one in which the other 63 lanes of its wavefront ran ahead through the barriers with a stale sequence number: the workgroup never
  // answered its first doorbell (round 4, found with progress marks in the control block).
  const bool bell_wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) == 0;
  const bool lane0 = (threadIdx.x & 63u) == 0u;
  uint32_t staged_gen = 0u;                                 // generation of the tables in LDS (0 = none: the host's generations start at 1)
  for (;;) {
    if (bell_wave) {
      // doorbell and request count are ONE aligned 8-byte word, written by the host with one store and read here with one load: a
      // second read of host memory behind the doorbell would be another ~1.2 us round trip over PCIe
      uint32_t v = seen, n_now = 0u;
      for (unsigned long long polls = 0; polls < max_idle_polls; ++polls) {
        const unsigned long long w = __hip_atomic_load((const unsigned long long*)&ctl->bell, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
        v = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)w);
        n_now = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(w >> 32));
        if (v != seen) break;
        __builtin_amdgcn_s_sleep(2);
      }
      if (lane0) {
        s_seq = v == seen ? kResQuit : v;                   // (idle for too long: leave; the library starts the kernel again when it needs it)
        s_n = n_now;
      }
    }
#ifdef EPPK_RESIDENT_STAMPS      // measurement build only: 100 MHz timestamps of the stages of a doorbell in the control block's spare words
    const unsigned long long ts0 = wall_clock64();
#endif
    __syncthreads();
    const uint32_t seq = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_seq), n_word = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_n);
    const uint32_t n = n_word & 0xFFu, kk = (n_word >> 8) & 0xFu, bufset = (n_word >> 12) & 3u;
    if (seq == kResQuit) break;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");           // system scope: the vector caches forget what they held before the doorbell
    __builtin_amdgcn_s_dcache_inv();                        // ... and the scalar cache (request headers, the argument block)
#ifdef EPPK_RESIDENT_STAMPS
    const unsigned long long ts1 = wall_clock64();
#endif
    if constexpr (QUAD) {
      // The form for batches of 8 requests and more (the library keeps one resident workgroup of each form and rings the one that suits
      // the batch: a request's latency is its chain of dependent loads -- row over PCIe, buckets, lists, tables -- and pick_quad_kernel's
      // body has one link more, the tier words of the listed pods: 11.9 against 11.0 us for one request, but a flat 12 us up to 16
      // requests, 13.4 against 15.5 for 32 and 19.4 against 27.9 for 64; scripts/res_sweep.py).
      // pick_quad_kernel's body -- four requests per wavefront, a third of the instructions per request -- and, for what it defers
      // (differing or overflowed lists, reserved hashes, an exhausted table ...), the work-list form of pick_fast_kernel's body over
      // this workgroup's own segments, exactly as pick_quad_kernel<TAIL> ends.  The two bodies share the front of the LDS layout
      // (base | lw | pterm) and both keep everything behind it all-zero between two requests, so the tables are staged ONCE per
      // snapshot generation, here, and both bodies are told that they are there.
      const ResidentArgs* a = args;
      const uint32_t gen = a->gen;
      if (gen != staged_gen) {
        double* s_base = (double*)smem;
        double* s_lw = s_base + (size_t)a->sn.J * 64u;
        double* s_pterm = s_lw + 4;
        for (uint32_t i = threadIdx.x; i < a->sn.J * 64u; i += blockDim.x) s_base[i] = a->sn.base[i];
        if (threadIdx.x < 4u) s_lw[threadIdx.x] = threadIdx.x == 0u ? a->tl.lw[0] : threadIdx.x == 1u ? a->tl.lw[1] : threadIdx.x == 2u ? a->tl.lw[2] : a->tl.lw[3];
        for (uint32_t i = threadIdx.x; i < a->pwn; i += blockDim.x) s_pterm[i] = a->sn.pterm[i];
        uint32_t* rest = (uint32_t*)(s_pterm + a->pwn);
        const uint32_t n_rest = (a->lds_bytes - (uint32_t)((unsigned char*)rest - smem)) / 4u;
        for (uint32_t i = threadIdx.x; i < n_rest; i += blockDim.x) rest[i] = 0u;
        if constexpr (MASKED) {   // the snapshot's three natural-layout sets, where pick_quad_body<MASKED> keeps them: behind the "listed" bits of all rows
          __syncthreads();
          uint64_t* s_nat = (uint64_t*)(rest + (blockDim.x >> 4) * (a->sn.J * 2u));
          for (uint32_t i = threadIdx.x; i < 192u; i += blockDim.x) s_nat[i] = a->sn.nat[i];
        }
        __syncthreads();
        staged_gen = gen;
      }
      ResidentBuf rb = a->buf[bufset];
      if constexpr (LEARN) {
        // the rows into device memory (8-byte words, the whole batch at once: every load in flight together), and the pick reads them there
        const unsigned long long* src = (const unsigned long long*)rb.reqs;
        unsigned long long* dst = (unsigned long long*)a->rows_copy;
        const uint32_t n_words = n * (a->stride / 8u);
        for (uint32_t i = threadIdx.x; i < n_words; i += blockDim.x) dst[i] = __builtin_nontemporal_load(src + i);
        __syncthreads();
        rb.reqs = a->rows_copy;
#ifdef EPPK_RESIDENT_STAMPS
        if (bell_wave && lane0) ctl->pad1[3] = (uint32_t)(wall_clock64() - ts0);      // bell seen -> rows copied
#endif
      }
      uint32_t n_def = pick_quad_body<LW, HAS_L, P_FIRST, MASKED, TOPK, LEARN, /*RESIDENT*/ true>(
          0u, 1u, smem, a->sn, a->ix, a->tl, rb.reqs, a->stride, n, a->pwn, MASKED ? rb.mask : nullptr, rb.out_pick, rb.out_score, nullptr, a->defer_cnt, a->defer_list, a->defer_cap,
          a->defer_total, TOPK ? kk : 1u, LEARN ? a->learn : nullptr, a->lds_bytes - (blockDim.x >> 6) * 512u,       // (crossbar scratch: the last 512 bytes per wavefront of the allocation)
          (const void*)nullptr);
      // MASKED single picks: the rows this wavefront PARKED (their candidates miss a QUEUE extreme -- what a subset filter of a few endpoints
      // does to nearly every request --, or the adapter's table holds none of them) are scored by the wavefront now, every candidate in full
      if constexpr (MASKED && EPPK_QUAD_PARK != 0) {
        const uint32_t n_x = n_def >> 16;
        n_def &= 0xFFFFu;
        if (n_x != 0u) quad_park_drain_resident<LW, HAS_L, TOPK, ResidentArgs>(n_x, smem, a, bufset, n, TOPK ? kk : 1u);
      }
      // ONE barrier ends the common case: picks and scores released to host memory, this wavefront's segment of the work list in device
      // memory (the system-scope release covers both), and the barrier that tells the doorbell wavefront "everybody is through" also
      // asks "did anybody defer?".  Only then the work-list pass, and a second release + barrier behind it.
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
      if (__syncthreads_or((int)(n_def != 0u))) {
        const KChain no_chain{};
        KWork wk;
        wk.cnt = a->defer_cnt; wk.list = a->defer_list; wk.total = a->defer_total; wk.report = a->defer_total; wk.cap = a->defer_cap; wk.n_segs = blockDim.x >> 6;
        // (MASKED: the quad layout keeps the snapshot's natural sets and the rows' candidate words where the fast body expects its all-zero
        // histogram: the rare pass over deferred requests stages its own layout -- "tables staged" false -- and the next doorbell stages the
        // quad layout again)
        pick_fast_body<LW, 6, HAS_L, true, P_FIRST, MASKED, /*BIG*/ true, /*GEN*/ false, TOPK, /*WL*/ true, /*RESIDENT*/ true>(
            0u, 1u, 0u, /*tables staged*/ !MASKED, smem, a->sn, a->ix, a->tl, rb.reqs, a->stride, n, a->pwn, MASKED ? rb.mask : nullptr, MASKED ? a->chain : no_chain, rb.out_pick,
            rb.out_score, nullptr, TOPK ? kk : 1u, wk);
        if constexpr (MASKED) staged_gen = 0u;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        __syncthreads();
      }
    } else {
      const ResidentArgs* a = args;
      const KChain no_chain{};
      const KWork no_work{};
      const uint32_t gen = a->gen;
      const ResidentBuf rb = a->buf[bufset];
      (void)kk;
      pick_fast_body<LW, 6, HAS_L, true, P_FIRST, /*MASKED*/ false, /*BIG*/ true, /*GEN*/ false, /*TOPK*/ false, /*WL*/ false, /*RESIDENT*/ true>(
          0u, 1u, 0u, /*tables staged*/ gen == staged_gen, smem, a->sn, a->ix, a->tl, rb.reqs, a->stride, n, a->pwn, nullptr, no_chain, rb.out_pick, rb.out_score, nullptr, 1u,
          no_work);
      staged_gen = gen;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");         // this wavefront's picks and scores are in host memory ...
      __syncthreads();
    }
#ifdef EPPK_RESIDENT_STAMPS
    const unsigned long long ts2 = wall_clock64();
#endif
#ifdef EPPK_RESIDENT_STAMPS
    if (bell_wave && lane0) {
      const unsigned long long ts3 = wall_clock64();
      ctl->pad1[0] = (uint32_t)(ts1 - ts0); ctl->pad1[1] = (uint32_t)(ts2 - ts1); ctl->pad1[2] = (uint32_t)(ts3 - ts2);   // (ts2: behind body, release and barrier)
    }
#endif
    if (bell_wave) {
      if (lane0) __hip_atomic_store(&ctl->done, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);   // ... (every wavefront's, behind the barrier) before the answer is
    }
    if constexpr (LEARN) {
      const ResidentArgs* a = args;
      ResidentBuf rb = a->buf[bufset];
      rb.reqs = a->rows_copy;
      resident_learn_update<LW>(a, rb, n);
      // (the update's stores and atomics are at agent scope; the word tells the HOST that they have all been issued and acknowledged)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __syncthreads();
      if (bell_wave) {
        if (lane0) __hip_atomic_store(&ctl->updated, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      // the quad layout's all-zero regions are untouched by the update (it uses static LDS of its own)
    }
    seen = seq;
  }
  if (bell_wave) {
    if (lane0) __hip_atomic_store(&ctl->state, kResExited, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// Does the pod set of `slot` (wave-uniform) contain this lane's pod p?  A listed set: its ids, broadcast one by one, against p;
// an overflowed one: p's bit in the dense row.
template <typename LW>
__device__ __forceinline__ uint32_t set_has(const KIndex& ix, uint32_t slot, uint32_t p, int lane) {
  const uint32_t lv = ix.lists_all[(size_t)slot * kListDwords + list_lane_dw(lane)];
  const uint32_t cnt = (uint32_t)__builtin_amdgcn_readlane((int)lv, (int)kListCap);
  if (cnt > kListCap) return (uint32_t)((((const LW*)ix.bitmaps)[(size_t)slot * 64u + (p & 63u)] >> (p >> 6)) & 1);
  uint32_t has = 0;
  for (uint32_t j = 0; j < cnt; ++j) {
    const uint32_t dwv = (uint32_t)__builtin_amdgcn_readlane((int)lv, (int)j);
    const uint32_t id = ((j >> 2) & 1u) ? (dwv >> 16) : (dwv & 0xFFFFu);
    has |= (uint32_t)(id == p);
  }
  return has;
}

// ---- CANDIDATE-MAJOR pick kernel (MASKED batches with few candidates: what a subset filter leaves, request.go:104-133) -----
// The fast kernel's masked routes cost O(pods) per request (the mask row is transposed, every pod's counter planes are built, and
// a candidate set that misses the snapshot-wide QUEUE extremes -- almost every small one -- takes the exact dense evaluation:
// 291 us for 64k requests with ~30 candidates each).  This kernel costs O(candidates): one candidate per lane (64 at a time),
// scored with the whole chain in chain order under the request's own QUEUE normalisers (SEMANTICS.md §2: they range over the
// candidates); matched[p] is one word of the dense row per hit; no top tables, no LDS.  Same results as the fast / generic
// kernels for any mask; the caller picks it when candidates are few (eppk_pick_batch_candidates_device, eppk_pick_batch_subset).
// (Tried first as a route INSIDE the fast kernel: its registers -- two binary64 divisions on top of the pipeline's landing
// registers -- spilled 60 VGPRs into the per-request loop and doubled the time of every masked batch; as a second loop of that
// kernel +30..50 %; as a called function, or at 3 waves per SIMD, worse still; as a second launch behind every masked launch
// +14 us even when it had nothing to do.  All measured, round 2.)
template <typename LW>
__global__ __launch_bounds__(256, 6) void pick_cands_kernel(KSnap sn, KIndex ix, KChain ch, const uint8_t* __restrict__ reqs, uint32_t stride, uint32_t n_reqs,
                                                         const uint64_t* __restrict__ cand_mask, int32_t* __restrict__ out_pick,
                                                         double* __restrict__ out_score, uint32_t tk) {
  const int lane = (int)(threadIdx.x & 63u);
  const uint32_t hwords = (stride - 8u) / 8u;
  const uint32_t wave0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
  const uint32_t nw = (gridDim.x * blockDim.x) >> 6;
  const __amdgpu_buffer_rsrc_t rk = ix.small ? __builtin_amdgcn_make_buffer_rsrc((void*)ix.bitmaps, 0, (int)ix.table_bytes, 0x00020000)
                                             : __builtin_amdgcn_make_buffer_rsrc((void*)ix.keys, 0, (int)((ix.slots + 2u) * 8u), 0x00020000);
  const uint32_t keys_off = ix.small ? ix.keys_off : 0u;
  bool has_q = false;
  for (uint32_t i = 0; i < ch.n; ++i) has_q |= ch.kind[i] == 1u;
  const uint32_t ki = (uint32_t)lane >> 1;
  const uint32_t hw0 = hwords < kKeysPerProbe ? hwords : kKeysPerProbe;
  for (uint32_t r = wave0; r < n_reqs; r += nw) {
    const uint8_t* row = reqs + (size_t)r * stride;
    // everything that depends on r alone is requested first: header, the lane pair's hash, the candidate row
    const uint64_t hdr = *(const uint64_t __attribute__((address_space(4)))*)row;
    const uint64_t hraw = hw0 ? ((const uint64_t*)(row + 8))[ki < hw0 ? ki : hw0 - 1u] : 0ull;
    const uint64_t cn = ((uint32_t)lane < sn.J) ? (cand_mask[(size_t)r * sn.J + (uint32_t)lane] & sn.nat[lane]) : 0ull;
    int32_t adapter = (int32_t)(uint32_t)hdr;
    uint32_t nb = (uint32_t)(hdr >> 32);
    bool bad = false;
    if (nb > hwords || adapter < -1 || adapter >= (int32_t)EPPK_MAX_ADAPTERS) {      // not scored + sticky flag (SEMANTICS.md §7)
      bad = true; adapter = -1; nb = 0u;
      if (lane == 0) atomicOr(sn.status, kStatusBadRow);
    }
    const uint32_t arow = adapter >= 0 ? (uint32_t)adapter : 128u;
    // the request's leading hits: the pair probe of the fast kernel (32 keys at a time, two lanes per key)
    const bool use_index = ix.slots != 0u && nb != 0u;
    const uint32_t nchunk = nb < kKeysPerProbe ? nb : kKeysPerProbe;
    ReqRegs q;
    q.hdr = 0;
    q.h = (use_index && ki < nchunk) ? hraw : 0ull;
    q.kw[0] = q.kw[1] = make_uint4(0, 0, 0, 0);
    q.bkt = 0;
    if (use_index) {
      pair_probe_prepare(ix, q);
      pair_probe_issue(rk, keys_off, q, lane);
    }
    // candidates (while the keys are in flight): lane w holds word w of the row (pods 64w .. 64w+63), holes and pods beyond the
    // snapshot already removed
    const uint32_t pc = (uint32_t)__builtin_popcountll(cn);
    uint32_t incl = pc;
#pragma unroll
    for (uint32_t dd = 1; dd < 64u; dd <<= 1) {
      const uint32_t t = (uint32_t)__shfl_up((int)incl, (int)dd);
      if ((uint32_t)lane >= dd) incl += t;
    }
    const uint32_t excl = incl - pc;
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    const uint32_t nchunks = (total + 63u) >> 6;
    // candidate number g (in ascending pod order) -> pod: the lane whose prefix range holds g, then the (g - excl)-th set bit of its word
    auto candidate = [&](uint32_t g, bool v) -> uint32_t {
      uint32_t owner = 0;                                              // = the number of lanes whose inclusive count is <= g
#pragma unroll
      for (uint32_t step = 32u; step; step >>= 1) {                    // binary search: incl is non-decreasing over the lanes
        const uint32_t probe_l = owner + step - 1u;
        if ((uint32_t)__shfl((int)incl, (int)probe_l) <= g) owner += step;
      }
      const uint32_t ow = owner < 64u ? owner : 63u;
      const uint32_t oex = (uint32_t)__shfl((int)excl, (int)ow);
      uint64_t w = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(cn >> 32), (int)ow) << 32) | (uint32_t)__shfl((int)(uint32_t)cn, (int)ow);
      if (!v) return 0u;
      for (uint32_t n = g - oex; n; --n) w &= w - 1ull;
      return ow * 64u + (uint32_t)__builtin_ctzll(w);
    };
    struct Facts { uint32_t p, q; bool v; double kvu; LW th, tl; };
    auto facts_of = [&](uint32_t c) -> Facts {
      Facts f;
      const uint32_t g = c * 64u + (uint32_t)lane;
      f.v = g < total;
      f.p = candidate(g, f.v);
      f.q = sn.queue[f.p];
      f.kvu = sn.kv[f.p];
      f.th = ((const LW*)sn.thi_t)[(size_t)arow * 64u + (f.p & 63u)];
      f.tl = ((const LW*)sn.tlo_t)[(size_t)arow * 64u + (f.p & 63u)];
      return f;
    };
    const Facts f0 = facts_of(0u);                                     // (the only pass when a request has at most 64 candidates)
    uint32_t m0 = 0, slot_eff = ix.slots + 2u;
    if (use_index) m0 = pair_probe_finish(ix, q, nchunk, lane, slot_eff);
    // QUEUE normalisers over ALL candidates
    uint32_t qmin = 0, qmax = 0;
    if (has_q) {
      uint32_t mn = f0.v ? f0.q : 0xFFFFFFFFu, mx = f0.v ? f0.q : 0u;
      for (uint32_t c = 1; c < nchunks; ++c) {
        const Facts f = facts_of(c);
        if (f.v) { mn = f.q < mn ? f.q : mn; mx = f.q > mx ? f.q : mx; }
      }
      qmin = wave_min_u32(mn);
      qmax = ~wave_min_u32(~mx);
    }
    const double qden = (double)(qmax - qmin);
    uint32_t rep[EPPK_MAX_TOPK];                                       // pods already reported (ordered fallbacks)
#pragma unroll
    for (int i = 0; i < (int)EPPK_MAX_TOPK; ++i) rep[i] = kNoPod;
    for (uint32_t round = 0; round < tk; ++round) {
      double best = -__builtin_inf();
      uint32_t bidx = kNoPod;
      for (uint32_t c = 0; c < nchunks; ++c) {
        const Facts f = c == 0u ? f0 : facts_of(c);
        const uint32_t p = f.p;
        bool v = f.v;
#pragma unroll
        for (int i = 0; i < (int)EPPK_MAX_TOPK; ++i) v = v && p != rep[i];
        // matched[p]: is this lane's candidate in the pod set of every leading hit? (a listed set: its ids against p, one by one; an
        // overflowed one: one word of its dense row)
        uint32_t cnt = 0;
        for (uint32_t k = 0; k < m0; ++k) {                            // (lane pairs beyond m0 hold the all-zero row)
          const uint32_t sk = (uint32_t)__builtin_amdgcn_readlane((int)slot_eff, (int)(2u * (k < kKeysPerProbe ? k : 0u)));
          cnt += set_has<LW>(ix, sk, p, lane);
        }
        if (__builtin_expect(m0 == kKeysPerProbe && nb > kKeysPerProbe, 0)) {       // hashes beyond the first 32 (every earlier key hit)
          bool stop = false;
          for (uint32_t b0 = kKeysPerProbe; b0 < nb && !stop; b0 += 64u) {
            const uint32_t i = b0 + (uint32_t)lane;
            const bool act = i < nb;
            const uint32_t slot = probe(ix, act ? ((const uint64_t*)(row + 8))[i] : 0ull, act);
            const unsigned long long found = __ballot(slot != kNotFound);
            const uint32_t chunk = (nb - b0) < 64u ? (nb - b0) : 64u;
            const uint32_t m = (~found == 0ull) ? 64u : (uint32_t)__builtin_ctzll(~found);
            for (uint32_t k = 0; k < m; ++k) {
              const uint32_t sk = (uint32_t)__builtin_amdgcn_readlane((int)slot, (int)k);
              cnt += set_has<LW>(ix, sk, p, lane);
            }
            if (m < chunk) stop = true;
          }
        }
        const uint32_t tier = (uint32_t)(((f.th >> (p >> 6)) & 1) << 1) | (uint32_t)((f.tl >> (p >> 6)) & 1);
        double t = 0.0;
        for (uint32_t i = 0; i < ch.n; ++i) {        // (the same expressions as masked_exact / the generic kernel: SEMANTICS.md §3)
          double sc;
          switch (ch.kind[i]) {
            case 1u: sc = (qmax == qmin) ? 1.0 : (double)(qmax - f.q) / qden; break;
            case 2u: sc = 1.0 - f.kvu; break;
            case 3u: sc = tier == 3u ? 1.0 : tier == 2u ? 0.8 : tier == 1u ? 0.6 : 0.0; break;
            default: sc = nb ? (double)cnt / (double)nb : 0.0; break;
          }
          t = t + clamp01(sc) * ch.w[i];
        }
        double cb = v ? t : -__builtin_inf();
        uint32_t ci = v ? p : kNoPod;
        wave_argmax_dpp(cb, ci);
        if (cb > best || (cb == best && ci < bidx)) { best = cb; bidx = ci; }
      }
      const bool none = bidx == kNoPod || bad;
      if (lane == 0) {
        out_pick[(size_t)r * tk + round] = none ? -1 : (int32_t)bidx;
        if (out_score) out_score[(size_t)r * tk + round] = none ? 0.0 : best;
      }
#pragma unroll
      for (int i = 0; i < (int)EPPK_MAX_TOPK; ++i)        // (static indices: rep stays in registers)
        if ((uint32_t)i == round) rep[i] = bidx;
    }
  }
}

// ---- GENERIC pick kernel -----------------------------------------------------------------------
// Any chain order (duplicates allowed), optional candidate mask; every scorer evaluated per pair in
// chain order.  Slower; it is both the fallback for non-canonical chains / masked batches and an
// independent on-device statement of SEMANTICS.md.
// LDS: queue[J*64] u32 | kv[J*64] f64 | per wave pw[pwn] f64 (raw ratios cnt/n).
// TOPK > 1: ordered fallbacks (PickResult.Fallbacks, handlers/server.go:72-77; 004-…/README.md:73): the `topk` (<= TOPK) best
// candidates per request under (total desc, index asc) go to out_pick[r*topk + i] / out_score[r*topk + i], padded with
// EPPK_NO_PICK / 0.0 when the request has fewer candidates.  Every lane keeps a sorted list of its TOPK best pods; the wave
// then merges the lists head by head (topk argmax rounds).
template <typename LW, int NPL, bool MASKED, int TOPK>
__global__ __launch_bounds__(512) void pick_generic_kernel(KSnap sn, KIndex ix, KChain ch, const uint8_t* __restrict__ reqs,
                                                           uint32_t stride, uint32_t n_reqs, uint32_t pwn,
                                                           const uint64_t* __restrict__ cand_mask,
                                                           int32_t* __restrict__ out_pick, double* __restrict__ out_score,
                                                           unsigned long long* __restrict__ stats, uint32_t topk) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double* s_kv = (double*)smem;
  uint32_t* s_q = (uint32_t*)(s_kv + (size_t)sn.J * 64u);
  const int lane = (int)(threadIdx.x & 63u);
  const uint32_t wib = threadIdx.x >> 6;
  const uint32_t wpb = blockDim.x >> 6;
  double* s_pw = (double*)(s_q + (size_t)sn.J * 64u) + (size_t)wib * pwn;
  LW* s_scr = (LW*)((double*)(s_q + (size_t)sn.J * 64u) + (size_t)wpb * pwn) + (size_t)wib * 64u;   // set_from_list's scratch: 64 lane words per wavefront

  for (uint32_t i = threadIdx.x; i < sn.J * 64u; i += blockDim.x) { s_kv[i] = sn.kv[i]; s_q[i] = sn.queue[i]; }
  s_scr[lane] = (LW)0;
  __syncthreads();

  bool has_q = false, has_l = false, has_p = false;
  for (uint32_t k = 0; k < ch.n; ++k) {
    has_q |= ch.kind[k] == 1u; has_l |= ch.kind[k] == 3u; has_p |= ch.kind[k] == 4u;
  }
  const LW valid = (LW)(valid_word<LW>(sn.n_pods, lane) & ((const LW*)sn.act_t)[lane]);
  unsigned long long w_hits = 0, w_lookups = 0;

  const uint32_t gwave = blockIdx.x * wpb + wib;
  const uint32_t nwaves = gridDim.x * wpb;
  for (uint32_t r = gwave; r < n_reqs; r += nwaves) {
    const uint8_t* row = reqs + (size_t)r * stride;
    int32_t adapter = __builtin_amdgcn_readfirstlane(((const int32_t*)row)[0]);
    uint32_t nb = (uint32_t)__builtin_amdgcn_readfirstlane(((const int32_t*)row)[1]);
    if (__builtin_expect(nb > (stride - 8u) / 8u || adapter < -1 || adapter >= (int32_t)EPPK_MAX_ADAPTERS, 0)) {
      // out-of-range row on a *_device entry point: not scored (EPPK_NO_PICK), flagged (eppk_launch_status)
      if (lane == 0) {
        for (uint32_t i = 0; i < topk; ++i) { out_pick[(size_t)r * topk + i] = -1; if (out_score) out_score[(size_t)r * topk + i] = 0.0; }
        atomicOr(sn.status, kStatusBadRow);
      }
      continue;
    }

    LW cand = valid;
    if (MASKED) cand &= transpose_mask<LW>(cand_mask + (size_t)r * sn.J, sn.J, lane);

    LW c[NPL];
#pragma unroll
    for (int k = 0; k < NPL; ++k) c[k] = 0;
    if (has_p) {
      for (uint32_t k = 0; k < ch.n; ++k) {
        if (ch.kind[k] != 4u) continue;   // every PREFIX entry performs (and is billed) its own walk; counts are identical
        LW cc[NPL];
#pragma unroll
        for (int q = 0; q < NPL; ++q) cc[q] = 0;
        const uint32_t hits = prefix_walk<LW, NPL>(ix, (const uint64_t*)(row + 8), nb, lane, cc, s_scr);
#pragma unroll
        for (int q = 0; q < NPL; ++q) c[q] = cc[q];
        w_hits += hits;
        w_lookups += (hits + 1u < nb) ? hits + 1u : nb;
      }
      wave_lds_fence();
      for (uint32_t cnt = (uint32_t)lane; cnt <= nb; cnt += 64u) s_pw[cnt] = nb ? (double)cnt / (double)nb : 0.0;
      wave_lds_fence();
    }

    LW thi = 0, tlo = 0;
    if (has_l) {
      const uint32_t arow = adapter >= 0 ? (uint32_t)adapter : 128u;
      thi = ((const LW*)sn.thi_t)[(size_t)arow * 64u + (uint32_t)lane];
      tlo = ((const LW*)sn.tlo_t)[(size_t)arow * 64u + (uint32_t)lane];
    }

    // QUEUE normalisers over the request's candidates
    uint32_t qmin = sn.qrange[0], qmax = sn.qrange[1];
    if (MASKED && has_q) {
      uint32_t mn = 0xFFFFFFFFu, mx = 0u;
      for (uint32_t j = 0; j < sn.J; ++j) {
        if ((cand >> j) & 1) {
          const uint32_t q = s_q[j * 64u + (uint32_t)lane];
          mn = q < mn ? q : mn;
          mx = q > mx ? q : mx;
        }
      }
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) {
        const uint32_t omn = (uint32_t)__shfl_xor((int)mn, off), omx = (uint32_t)__shfl_xor((int)mx, off);
        mn = omn < mn ? omn : mn;
        mx = omx > mx ? omx : mx;
      }
      qmin = mn; qmax = mx;
    }
    const double qden = (double)(qmax - qmin);

    double bt[TOPK];       // this lane's best pods so far, (total desc, index asc)
    uint32_t bp[TOPK];
#pragma unroll
    for (int i = 0; i < TOPK; ++i) { bt[i] = -__builtin_inf(); bp[i] = kNoPod; }
    for (uint32_t j = 0; j < sn.J; ++j) {
      const uint32_t p = j * 64u + (uint32_t)lane;
      const uint32_t tier = (uint32_t)(((thi >> j) & 1) << 1) | (uint32_t)((tlo >> j) & 1);
      uint32_t cnt = 0;
#pragma unroll
      for (int k = 0; k < NPL; ++k) cnt |= (uint32_t)((c[k] >> j) & 1) << k;
      double t = 0.0;
      for (uint32_t k = 0; k < ch.n; ++k) {
        double s;
        switch (ch.kind[k]) {
          case 1u: s = (qmax == qmin) ? 1.0 : (double)(qmax - s_q[p]) / qden; break;
          case 2u: s = 1.0 - s_kv[p]; break;
          case 3u: s = tier == 3u ? 1.0 : tier == 2u ? 0.8 : tier == 1u ? 0.6 : 0.0; break;
          default: s = s_pw[cnt]; break;
        }
        t = t + clamp01(s) * ch.w[k];
      }
      const bool ok = (cand >> j) & 1;
      if (ok) {              // sorted insert; pods arrive in ascending index, so an equal total stays behind the earlier pod
#pragma unroll
        for (int i = TOPK - 1; i >= 0; --i) {
          const bool gt = t > bt[i];
          if (i + 1 < TOPK && gt) { bt[i + 1] = bt[i]; bp[i + 1] = bp[i]; }
          bool place = gt;
          if (i > 0) place = gt && !(t > bt[i - 1]);
          if (place) { bt[i] = t; bp[i] = p; }
        }
      }
    }
    if (TOPK == 1) {
      double best = bt[0];
      uint32_t bidx = bp[0];
      wave_argmax(best, bidx);
      if (lane == 0) {
        const bool none = bidx == kNoPod;
        out_pick[r] = none ? -1 : (int32_t)bidx;
        if (out_score) out_score[r] = none ? 0.0 : best;
      }
    } else {
      for (uint32_t i = 0; i < topk; ++i) {
        double best = bt[0];
        uint32_t bidx = bp[0];
        wave_argmax_dpp(best, bidx);
        if (lane == 0) {
          const bool none = bidx == kNoPod;
          out_pick[(size_t)r * topk + i] = none ? -1 : (int32_t)bidx;
          if (out_score) out_score[(size_t)r * topk + i] = none ? 0.0 : best;
        }
        if (bidx != kNoPod && (uint32_t)lane == (bidx & 63u)) {   // the owner pops its head
#pragma unroll
          for (int q = 0; q + 1 < TOPK; ++q) { bt[q] = bt[q + 1]; bp[q] = bp[q + 1]; }
          bt[TOPK - 1] = -__builtin_inf();
          bp[TOPK - 1] = kNoPod;
        }
      }
    }
  }
  if (stats && lane == 0 && (w_hits | w_lookups) && gwave < kStatSlots) {
    stats[4 + 2 * gwave] += w_hits;
    stats[5 + 2 * gwave] += w_lookups;
  }
}

// ---- on-device prompt hashing (SEMANTICS.md §4; 0602-…/README.md:99) -------------------------------------
// One thread per request walks its prompt block by block: h[i] = XXH64(block_i || LE64(h[i-1])), seed 0, and
// writes the complete request row {adapter, n_blocks, h[0..)}.  Blocks are 8-byte multiples (block_chars % 8
// == 0, prompts 8-byte aligned), so every read is an aligned u64 and the XXH64 tail has no 4-/1-byte steps.
// The chain is sequential per request and independent across requests; each lane streams its own prompt
// (64 B per block = one sector per lane per step).  Bound: HBM read of the prompt bytes.
namespace xxh {
constexpr uint64_t P1 = 0x9E3779B185EBCA87ull, P2 = 0xC2B2AE3D27D4EB4Full, P3 = 0x165667B19E3779F9ull,
                   P4 = 0x85EBCA77C2B2AE63ull, P5 = 0x27D4EB2F165667C5ull;
__device__ __forceinline__ uint64_t rotl(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
__device__ __forceinline__ uint64_t round1(uint64_t acc, uint64_t in) { return rotl(acc + in * P2, 31) * P1; }
__device__ __forceinline__ uint64_t merge(uint64_t h, uint64_t acc) { return (h ^ round1(0, acc)) * P1 + P4; }
}  // namespace xxh

// XXH64(seed 0) of nw 8-byte words at `w` followed by the single word `last` (total length 8*(nw+1) bytes).
__device__ __forceinline__ uint64_t xxh64_words_plus(const uint64_t* w, uint32_t nw, uint64_t last) {
  using namespace xxh;
  const uint32_t total = nw + 1u;                 // words
  const uint64_t len = (uint64_t)total * 8u;
  auto word = [&](uint32_t i) -> uint64_t { return i < nw ? w[i] : last; };
  uint64_t h;
  uint32_t i = 0;
  if (total >= 4u) {
    uint64_t v1 = P1 + P2, v2 = P2, v3 = 0, v4 = 0ull - P1;
    for (; i + 4u <= total; i += 4u) {
      v1 = round1(v1, word(i));
      v2 = round1(v2, word(i + 1));
      v3 = round1(v3, word(i + 2));
      v4 = round1(v4, word(i + 3));
    }
    h = rotl(v1, 1) + rotl(v2, 7) + rotl(v3, 12) + rotl(v4, 18);
    h = merge(h, v1); h = merge(h, v2); h = merge(h, v3); h = merge(h, v4);
  } else {
    h = P5;
  }
  h += len;
  for (; i < total; ++i) h = rotl(h ^ round1(0, word(i)), 27) * P1 + P4;
  h = (h ^ (h >> 33)) * P2;
  h = (h ^ (h >> 29)) * P3;
  return h ^ (h >> 32);
}

#ifdef EPPK_MAIN_UNIT   // non-template kernels are defined once, in eppk.hip (the pick units include this header too)
__global__ void hash_prompts_kernel(const uint8_t* __restrict__ prompts, uint64_t prompt_stride, const uint32_t* __restrict__ prompt_len,
                                    const uint64_t* __restrict__ seeds, const int32_t* __restrict__ adapters, uint32_t n_reqs,
                                    uint32_t block_chars, uint32_t max_blocks, uint8_t* __restrict__ rows, uint32_t stride) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_reqs) return;
  const uint64_t* p = (const uint64_t*)(prompts + (size_t)r * prompt_stride);
  const uint32_t wpb = block_chars / 8u;
  uint32_t nblk = prompt_len[r] / block_chars;
  if (nblk > max_blocks) nblk = max_blocks;
  uint64_t prev = seeds[r];
  uint64_t* out = (uint64_t*)(rows + (size_t)r * stride);
  for (uint32_t b = 0; b < nblk; ++b) {
    prev = xxh64_words_plus(p + (size_t)b * wpb, wpb, prev);
    out[1 + b] = prev;
  }
  for (uint32_t b = nblk; b < max_blocks; ++b) out[1 + b] = 0ull;
  out[0] = (uint64_t)(uint32_t)adapters[r] | ((uint64_t)nblk << 32);
}
#endif

// ---- subset filter for a batch (request.go:104-133; include/eppk.h "the subset filter for a whole batch") ---------------------
// at / av: open-addressing table of the published endpoints' fingerprints (built on the host at eppk_snapshot_set_addresses:
// two entries per pod -- address, address + port -- at most a quarter full; av = pod + 1, 0 = empty; equal fingerprints sit in
// one probe run: several pods may share an address).  A wavefront per request: lane e looks entry e up and ORs the pods it
// finds into the request's mask row in LDS; the row goes out with one coalesced store.
#ifdef EPPK_MAIN_UNIT
__global__ __launch_bounds__(256) void subset_masks_kernel(const uint64_t* __restrict__ at, const uint32_t* __restrict__ av, uint32_t tmask,
                                                           const uint64_t* __restrict__ keys, const uint32_t* __restrict__ off, uint32_t n_reqs,
                                                           uint32_t n_pods, uint64_t* __restrict__ out) {
  __shared__ unsigned long long s_row[4][64];
  const uint32_t lane = threadIdx.x & 63u, wib = threadIdx.x >> 6;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  const uint32_t J = (n_pods + 63u) / 64u;
  unsigned long long* row = s_row[wib];
  for (uint32_t r = wave; r < n_reqs; r += nwaves) {
    row[lane] = 0ull;
    wave_lds_fence();
    bool all = false;
    const uint32_t e0 = off[r], e1 = off[r + 1];
    for (uint32_t eb = e0; eb < e1; eb += 64u) {
      const uint32_t e = eb + lane;
      if (e < e1) {
        const uint64_t lo = keys[2 * (size_t)e], hi = keys[2 * (size_t)e + 1];
        if ((lo | hi) == 0ull) all = true;                       // the "no filter" entry
        else {
          uint32_t t = (uint32_t)lo & tmask;
          for (uint32_t n = 0; n <= tmask; ++n) {
            const uint32_t v = av[t];
            if (v == 0u) break;
            if (at[2 * (size_t)t] == lo && at[2 * (size_t)t + 1] == hi) atomicOr(&row[(v - 1u) >> 6], 1ull << ((v - 1u) & 63u));
            t = (t + 1u) & tmask;
          }
        }
      }
    }
    wave_lds_fence();
    const bool any_all = __any(all);
    if (lane < J) {
      unsigned long long w = any_all ? ~0ull : row[lane];
      if (lane == J - 1u && (n_pods & 63u)) w &= (1ull << (n_pods & 63u)) - 1ull;     // no bits beyond the snapshot
      out[(size_t)r * J + lane] = w;
    }
    wave_lds_fence();
  }
}
#endif

// ---- snapshot producer: raw pod rows -> the device layout, on the device ------------------------
// A publish is one H2D copy of the raw 64-byte rows plus three small launches; every value is computed with the same
// binary64 operations, in the same order, as SEMANTICS.md §2 prescribes (so the fused terms stay bit-exact).

// (0) one workgroup: minimum / maximum queue depth over the active pods (holes excluded) -> qr[0], qr[1]  (0, 0 when none)
#ifdef EPPK_MAIN_UNIT
__global__ __launch_bounds__(1024) void snap_qrange_kernel(const eppk_pod_row* __restrict__ rows, uint32_t n_pods, uint32_t* __restrict__ qr) {
  __shared__ uint32_t s_mn[16], s_mx[16];
  uint32_t mn = 0xFFFFFFFFu, mx = 0u;
  bool any = false;
  for (uint32_t p = threadIdx.x; p < n_pods; p += blockDim.x)
    if (!(rows[p].flags & EPPK_POD_INACTIVE)) {
      const uint32_t q = rows[p].queue;
      mn = q < mn ? q : mn; mx = q > mx ? q : mx; any = true;
    }
  for (int off = 32; off >= 1; off >>= 1) {
    const uint32_t omn = (uint32_t)__shfl_xor((int)mn, off), omx = (uint32_t)__shfl_xor((int)mx, off);
    mn = omn < mn ? omn : mn; mx = omx > mx ? omx : mx;
  }
  (void)any;
  if ((threadIdx.x & 63u) == 0u) { s_mn[threadIdx.x >> 6] = mn; s_mx[threadIdx.x >> 6] = mx; }
  __syncthreads();
  if (threadIdx.x == 0u) {
    for (uint32_t w = 1; w < (blockDim.x >> 6); ++w) { mn = s_mn[w] < mn ? s_mn[w] : mn; mx = s_mx[w] > mx ? s_mx[w] : mx; }
    if (mn > mx) { mn = 0u; mx = 0u; }         // no active pod
    qr[0] = mn; qr[1] = mx;
  }
}

// Assumed load (SEMANTICS.md §2b): queue[pick] += 1 for every routed request of an epoch; picks[r * k] is request r's pick.
__global__ void assumed_bump_kernel(eppk_pod_row* __restrict__ rows, const int32_t* __restrict__ picks, uint32_t n, uint32_t k, uint32_t n_pods) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const int32_t p = picks[(size_t)r * k];
  if (p >= 0 && (uint32_t)p < n_pods) atomicAdd(&rows[p].queue, 1u);
}

// Picker "random-top-k" (SEMANTICS.md §3b): entry (splitmix64(seed + (r+1)*golden) mod n) of request r's ordered fallback list.
// r0 = index of the first request of this launch within its batch.
__global__ void random_select_kernel(const int32_t* __restrict__ tp, const double* __restrict__ ts, uint32_t n, uint32_t k, uint64_t seed, uint32_t r0,
                                     int32_t* __restrict__ out_pick, double* __restrict__ out_score) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  uint32_t cnt = 0;
  while (cnt < k && tp[(size_t)r * k + cnt] >= 0) ++cnt;
  if (cnt == 0u) { out_pick[r] = -1; if (out_score) out_score[r] = 0.0; return; }
  uint64_t z = seed + ((uint64_t)(r0 + r) + 1ull) * 0x9E3779B97F4A7C15ull;
  z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
  z ^= z >> 27; z *= 0x94D049BB133111EBull;
  z ^= z >> 31;
  const uint32_t j = (uint32_t)(z % (uint64_t)cnt);
  out_pick[r] = tp[(size_t)r * k + j];
  if (out_score) out_score[r] = ts[(size_t)r * k + j];
}
#endif

// (1) thread per pod: fused leading pod-only terms base[p] (chain order), raw gauges for the generic kernel
// `lead` = the pod-only scorers in front of the first LORA / PREFIX (folded into base[p]); `postc` = the pod-only scorers
// behind it (n <= 2): their products clamp01(s) * w go to post0[p] / post1[p], one array each (they are added one by one).
#ifdef EPPK_MAIN_UNIT
__global__ void snap_terms_kernel(const eppk_pod_row* __restrict__ rows, uint32_t n_pods, uint32_t np64, const uint32_t* __restrict__ qr,
                                  KChain lead, KChain postc, double* __restrict__ base, double* __restrict__ post0, double* __restrict__ post1,
                                  uint32_t* __restrict__ queue, double* __restrict__ kv) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= np64) return;
  const uint32_t qmin = qr[0], qmax = qr[1];
  double t = 0.0, k = 0.0, pp[2] = {0.0, 0.0};
  uint32_t q = 0;
  if (p < n_pods) {
    q = rows[p].queue;
    k = rows[p].kv_util;
    auto score = [&](uint32_t kind) { return kind == 1u ? ((qmax == qmin) ? 1.0 : (double)(qmax - q) / (double)(qmax - qmin)) : 1.0 - k; };
    for (uint32_t i = 0; i < lead.n; ++i) t = t + clamp01(score(lead.kind[i])) * lead.w[i];
    for (uint32_t i = 0; i < postc.n && i < 2u; ++i) pp[i] = clamp01(score(postc.kind[i])) * postc.w[i];
  }
  base[p] = t; post0[p] = pp[0]; post1[p] = pp[1]; queue[p] = q; kv[p] = k;
}
#endif

// (2) thread per (adapter row a, lane l): lane-transposed LoRA tier planes (row 128 = base model: in no set)
//       hi = active | free, lo = active | (~free & waiting)  ->  tier = 2*hi + lo (SEMANTICS.md §3 LORA);
//     row 129 of the launch builds the lane words of the pods at the minimum / maximum queue depth instead.
template <typename LW>
__global__ void snap_planes_kernel(const eppk_pod_row* __restrict__ rows, uint32_t n_pods, uint32_t J, const uint32_t* __restrict__ qr,
                                   LW* __restrict__ thi, LW* __restrict__ tlo, LW* __restrict__ qmin_t, LW* __restrict__ qmax_t,
                                   LW* __restrict__ act_t, uint64_t* __restrict__ nat) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t a = t >> 6, l = t & 63u;
  const uint32_t qmin = qr[0], qmax = qr[1];
  if (a == 130u) {           // natural-layout words: nat[0][l] active pods 64l .. 64l+63, nat[1][l] those at qmin, nat[2][l] at qmax
    uint64_t ac = 0, mn = 0, mx = 0;
    for (uint32_t b = 0; b < 64u; ++b) {
      const uint32_t p = l * 64u + b;
      if (p >= n_pods) break;
      const eppk_pod_row& r = rows[p];
      if (r.flags & EPPK_POD_INACTIVE) continue;
      ac |= 1ull << b;
      if (r.queue == qmin) mn |= 1ull << b;
      if (r.queue == qmax) mx |= 1ull << b;
    }
    nat[l] = ac; nat[64u + l] = mn; nat[128u + l] = mx;
    return;
  }
  if (a > 129u) return;
  LW hi = 0, lo = 0, ac = 0;
  for (uint32_t j = 0; j < J; ++j) {
    const uint32_t p = j * 64u + l;
    if (p >= n_pods) break;
    const eppk_pod_row& r = rows[p];
    if (r.flags & EPPK_POD_INACTIVE) continue;        // a hole: in no set (qmin / qmax range over the active pods only)
    ac |= (LW)((LW)1 << j);
    if (a == 129u) {
      if (r.queue == qmin) hi |= (LW)((LW)1 << j);
      if (r.queue == qmax) lo |= (LW)((LW)1 << j);
    } else {
      const uint32_t loaded = (uint32_t)(__popcll(r.active[0]) + __popcll(r.active[1]) + __popcll(r.waiting[0]) + __popcll(r.waiting[1]));
      const bool freeslot = loaded < r.max_lora;
      const bool act = a < 128u && ((r.active[a >> 6] >> (a & 63u)) & 1ull);
      const bool wai = a < 128u && ((r.waiting[a >> 6] >> (a & 63u)) & 1ull);
      if (act || freeslot) hi |= (LW)((LW)1 << j);
      if (act || (!freeslot && wai)) lo |= (LW)((LW)1 << j);
    }
  }
  if (a == 129u) { qmin_t[l] = hi; qmax_t[l] = lo; act_t[l] = ac; }
  else {
    thi[(size_t)a * 64u + l] = hi; tlo[(size_t)a * 64u + l] = lo;
    LW* thl = tlo + 129u * 64u;                     // the interleaved copy right behind the lo planes (SnapOff<LW>::thl)
    thl[((size_t)a * 64u + l) * 2u] = hi; thl[((size_t)a * 64u + l) * 2u + 1u] = lo;
  }
}

// (3) workgroup per adapter row: the 64 best pods by T_a[p] = base[p] (+ lw[tier(a,p)]) under (T desc, p asc) -- the exact
//     total of every pod without a prefix match (FAST pick kernel).  64 rounds of a block-wide argmax over T staged in LDS.
template <typename LW>
__global__ __launch_bounds__(256) void snap_top_kernel(const double* __restrict__ base, const LW* __restrict__ thi, const LW* __restrict__ tlo,
                                                       uint32_t n_pods, uint32_t np64, uint32_t has_l, KTail tl,
                                                       const double* __restrict__ post0, const double* __restrict__ post1,
                                                       const LW* __restrict__ act, double* __restrict__ topv, uint32_t* __restrict__ topi) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double* sT = (double*)smem;                         // [np64]
  __shared__ double red_t[4];
  __shared__ uint32_t red_p[4];
  const uint32_t a = blockIdx.x, tid = threadIdx.x;
  TopEntry* top16 = (TopEntry*)((uint8_t*)topv + SnapOff<LW>::top16);   // (topv is the head of the snapshot blob: SnapOff<LW>::topv == 0)
  if ((!has_l && a != 128u) || n_pods == 0u) {        // without a LoRA scorer only the base row is read
    if (tid < 64u) { topv[(size_t)a * 64u + tid] = -__builtin_inf(); topi[(size_t)a * 64u + tid] = kNoPod; }
    if (tid < 16u) top16[(size_t)a * 16u + tid] = TopEntry{-__builtin_inf(), kNoPod, 0u};
    return;
  }
  for (uint32_t p = tid; p < np64; p += blockDim.x) {
    double t = -__builtin_inf();
    if (p < n_pods && ((act[p & 63u] >> (p >> 6)) & 1)) {     // (holes never enter a table: T = -inf)
      t = base[p];
      double lterm = 0.0;
      if (has_l) {
        const uint32_t l = p & 63u, j = p >> 6;
        const uint32_t tier = (uint32_t)(((thi[(size_t)a * 64u + l] >> j) & 1) << 1) | (uint32_t)((tlo[(size_t)a * 64u + l] >> j) & 1);
        lterm = tier_term(tl, tier);
      }
      // the chain behind base[], in order, WITHOUT its PREFIX entry (a pod without a prefix match adds +-0.0 there: identity)
      for (uint32_t i = 0; i < tl.n_tail; ++i) {
        const uint32_t kd = tl.kind[i];
        if (kd == 0u) t = t + lterm;
        else if (kd == 2u) t = t + post0[p];
        else if (kd == 3u) t = t + post1[p];
      }
    }
    sT[p] = t;
  }
  __syncthreads();
  uint32_t n_act = 0;                                 // active pods (every thread counts: 64 lane words)
  for (uint32_t l = 0; l < 64u; ++l) n_act += (uint32_t)__builtin_popcountll((unsigned long long)(LW)(act[l] & valid_word<LW>(n_pods, (int)l)));
  const uint32_t K = n_act < 64u ? n_act : 64u;
  for (uint32_t k = 0; k < 64u; ++k) {
    if (k >= K) {                                     // fewer than 64 active pods: pad
      if (tid == 0) {
        topv[(size_t)a * 64u + k] = -__builtin_inf(); topi[(size_t)a * 64u + k] = kNoPod;
        if (k < 16u) top16[(size_t)a * 16u + k] = TopEntry{-__builtin_inf(), kNoPod, 0u};
      }
      continue;
    }
    double best = -__builtin_inf();
    uint32_t bi = kNoPod;
    for (uint32_t p = tid; p < np64; p += blockDim.x) {   // ascending p, strict >: the lowest index among equal T
      const double v = sT[p];
      if (v > best) { best = v; bi = p; }
    }
    wave_argmax_dpp(best, bi);
    if ((tid & 63u) == 0u) { red_t[tid >> 6] = best; red_p[tid >> 6] = bi; }
    __syncthreads();
    if (tid == 0) {
      double bt = red_t[0];
      uint32_t bp = red_p[0];
      for (uint32_t w = 1; w < (blockDim.x >> 6); ++w)
        if (red_t[w] > bt || (red_t[w] == bt && red_p[w] < bp)) { bt = red_t[w]; bp = red_p[w]; }
      topv[(size_t)a * 64u + k] = bt;
      topi[(size_t)a * 64u + k] = bp;
      if (k < 16u) top16[(size_t)a * 16u + k] = TopEntry{bt, bp, 0u};
      sT[bp] = -__builtin_inf();                      // taken
    }
    __syncthreads();
  }
}

// ---- prefix index maintenance (0602-…/README.md:101-108) -----------------------------------------
//
// LISTS FIRST (round 3).  A key's pod set lives in ONE of two places:
//   * its short list (64 bytes: at most kListCap 16-bit pod ids, ascending, + the count) while it has at most kListCap members --
//     the dense row of such a slot is ALL ZERO and nobody touches it;
//   * its dense row (64 * sizeof(LW) bytes, one bit per pod) once a 25th pod arrived: count > kListCap ("overflowed"), the ids in
//     the list are then unspecified.  A row that shrinks back to kListCap members returns to its list (and is zeroed).
// Invariants (index_selfcheck_kernel checks them all): an EMPTY word (and a bucket header, the zero row) has an empty list -- count 0,
// every id 0xFFFF -- and an all-zero row; a TOMBSTONE has an all-zero row and a list line whose positions from 1 on are 0xFFFF
// (position 0 and the count may be what its previous occupant left: round 4, "stamps as tags" below); a present key has a non-empty
// set; list ids are unique, below 64 * bits(LW), strictly ascending after every entry point of the library; unused id positions and the
// three spare dwords are 0xFFFF.
// Why: the post-route update of a 64k x 32-block batch makes ~1 Mi NEW keys, each a single-pod set.  With the row as the arbiter
// ("did my atomicOr set the bit?") a new key touched four random 64-byte HBM lines (bucket, stamp, row word, list); random-line
// atomics run at 20 G lines/s on this GPU (scripts/micro/linermw.hip), so the update could not go below ~205 us per Mi keys and the
// closed loop sat at 143 M decisions/s.  Now the LIST is the arbiter: an id is appended by a 32-bit compare-and-swap on the dword
// that holds the first free position, so two inserters of one (key, pod) pair cannot both land it, and the row is not touched at
// all: three lines per new key (bucket CAS, list CAS, stamp store), and an eviction victim costs its list and its key.

// A 64-byte line -> 16 dwords by four 16-byte loads.  COHERENT: the loads bypass the non-coherent caches (the vector L1 of the CU,
// and the L2 of this XCD for lines another XCD may have changed during this kernel) -- what every RETRY of the insert protocol uses;
// the first look at a bucket or a list may be an ordinary cached load (half the cost: profiles/r03_d_micro_insert_variants.txt), because
// a stale line only ever shows an EARLIER state of this launch (words and ids appear, nothing disappears while an insert kernel runs)
// and every decision taken on it is confirmed by a compare-and-swap, which returns the truth.
template <bool COHERENT = true>
__device__ __forceinline__ void load_line16(const uint32_t* p, uint32_t (&d)[16]) {
  u32x4_t q0, q1, q2, q3;
  if constexpr (COHERENT) {
    asm volatile("global_load_dwordx4 %0, %4, off sc0 sc1\n\tglobal_load_dwordx4 %1, %4, off offset:16 sc0 sc1\n\t"
                 "global_load_dwordx4 %2, %4, off offset:32 sc0 sc1\n\tglobal_load_dwordx4 %3, %4, off offset:48 sc0 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3) : "v"(p) : "memory");
  } else {
    asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:16\n\t"
                 "global_load_dwordx4 %2, %4, off offset:32\n\tglobal_load_dwordx4 %3, %4, off offset:48\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3) : "v"(p) : "memory");
  }
  d[0] = q0.x; d[1] = q0.y; d[2] = q0.z; d[3] = q0.w; d[4] = q1.x; d[5] = q1.y; d[6] = q1.z; d[7] = q1.w;
  d[8] = q2.x; d[9] = q2.y; d[10] = q2.z; d[11] = q2.w; d[12] = q3.x; d[13] = q3.y; d[14] = q3.z; d[15] = q3.w;
}
__device__ __forceinline__ void store_line16(uint32_t* p, const uint32_t (&d)[16]) {
  u32x4_t* P = (u32x4_t*)p;
  const u32x4_t q0 = {d[0], d[1], d[2], d[3]}, q1 = {d[4], d[5], d[6], d[7]}, q2 = {d[8], d[9], d[10], d[11]}, q3 = {d[12], d[13], d[14], d[15]};
  P[0] = q0; P[1] = q1; P[2] = q2; P[3] = q3;
}
// id number j of a list: dword and half of the 16-dword line (list_pos(j) = 8 * (j & 3) + (j >> 2) as a u16 index)
__host__ __device__ __forceinline__ constexpr uint32_t list_dw(uint32_t j) { return 4u * (j & 3u) + (j >> 3); }
__host__ __device__ __forceinline__ constexpr uint32_t list_hi(uint32_t j) { return (j >> 2) & 1u; }
__device__ __forceinline__ uint32_t list_id(const uint32_t (&d)[16], uint32_t j) {      // (j: compile-time constant after unrolling)
  return list_hi(j) ? (d[list_dw(j)] >> 16) : (d[list_dw(j)] & 0xFFFFu);
}
constexpr uint32_t kListBusy = 0x80000000u;   // in the count dword, only while index_lists_sort_kernel works on the list

// Set pod's bit in the row of `slot` (an OVERFLOWED set); true iff this call set it.
template <typename LW>
__device__ __forceinline__ bool bitmap_set(void* bitmaps, uint32_t slot, uint32_t pod) {
  const uint32_t lane = pod & 63u, j = pod >> 6;
  if constexpr (sizeof(LW) == 8) {
    unsigned long long* w = (unsigned long long*)bitmaps + (size_t)slot * 64u + lane;
    if ((__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> j) & 1ull) return false;
    return !((atomicOr(w, 1ull << j) >> j) & 1ull);
  } else if constexpr (sizeof(LW) == 4) {
    unsigned int* w = (unsigned int*)bitmaps + (size_t)slot * 64u + lane;
    if ((__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> j) & 1u) return false;
    return !((atomicOr(w, 1u << j) >> j) & 1u);
  } else {
    const size_t e = (size_t)slot * 64u + lane;           // u16 element index
    unsigned int* w = (unsigned int*)bitmaps + (e >> 1);
    const unsigned int bit = (1u << j) << (16u * (uint32_t)(e & 1u));
    if (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit) return false;
    return !(atomicOr(w, bit) & bit);
  }
}

// Add `pod` to the short list L whose line was read into d (KNOWN_EMPTY: the caller just claimed the key -- the list is in its
// reset state and was not read).  Returns 0 = already listed, 1 = appended at position `pos`, 2 = the list is full of other pods.
// Protocol: ids only ever appear (nothing is removed while an insert kernel runs) and an id goes into the FIRST free position, by a
// compare-and-swap on the dword that holds it (two ids per dword: the CAS changes one half and fails if either half moved).  A
// thread lands its pod at position f only after it has seen every position below f hold some OTHER pod -- in its first read of
// the line or in the value a failed CAS returned -- so no pod is ever listed twice, whatever the interleaving, and the filled
// positions always form a prefix: the count is an atomicMax of (position + 1).
template <bool KNOWN_EMPTY>
__device__ __forceinline__ uint32_t list_add(uint32_t* L, const uint32_t (&d)[16], uint32_t pod, uint32_t& pos) {
  uint32_t j = kListCap, exp = 0xFFFFFFFFu;
  if constexpr (KNOWN_EMPTY) {
    j = 0u;
  } else {
    bool found = false;
#pragma unroll
    for (int jj = (int)kListCap - 1; jj >= 0; --jj) {
      const uint32_t id = list_id(d, (uint32_t)jj);
      if (id == kListNone) { j = (uint32_t)jj; exp = d[list_dw((uint32_t)jj)]; }
      found = found || id == pod;
    }
    if (found) return 0u;
  }
  while (j < kListCap) {
    const uint32_t hi = list_hi(j);
    const uint32_t mine = hi ? (exp >> 16) : (exp & 0xFFFFu);
    if (mine == pod) return 0u;
    if (mine != kListNone) {                               // somebody else's pod sits here by now: on to the next position
      ++j;
      if (j < kListCap) exp = __hip_atomic_load(&L[list_dw(j)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      continue;
    }
    const uint32_t des = hi ? ((exp & 0x0000FFFFu) | (pod << 16)) : ((exp & 0xFFFF0000u) | pod);
    const uint32_t seen = atomicCAS(&L[list_dw(j)], exp, des);
    if (seen == exp) {
      pos = j;
      atomicMax(&L[3], j + 1u);
      return 1u;
    }
    exp = seen;                                            // (either half moved: look again)
  }
  return 2u;
}

// The 25th pod of a set: the set moves to its dense row.  The thread that raises the count past kListCap copies the 24 listed
// pods into the row (every position is final by then: the line is read again, coherently); every thread that arrives on a full
// list -- the mover included -- sets its own pod's bit.  Inserts that come later see count > kListCap and go to the row directly.
// Bits 1..7 of a bucket's header word say which of its slots are overflowed (bit 0: the chain continues): the eviction scan, which
// streams the key words anyway, then knows without reading a victim's list line whether a dense row has to be zeroed.
template <typename LW>
__device__ __forceinline__ void list_overflow(uint64_t* keys, uint32_t slots, void* bitmaps, uint32_t* L, uint32_t slot, uint32_t pod) {
  const uint32_t old = atomicMax(&L[3], kListCap + 1u);
  if (old <= kListCap) {
    uint32_t d[16];
    load_line16(L, d);
#pragma unroll
    for (uint32_t jj = 0; jj < kListCap; ++jj) {
      const uint32_t id = list_id(d, jj);
      if ((id >> 6) < 8u * (uint32_t)sizeof(LW)) bitmap_set<LW>(bitmaps, slot, id);
    }
  }
  bitmap_set<LW>(bitmaps, slot, pod);
}

// 24 ids ascending (unused positions 0xFFFF sort to the end): a bitonic network over 32 registers, fully unrolled.
__device__ __forceinline__ void sort_ids(uint32_t (&a)[32]) {
#pragma unroll
  for (uint32_t k = 2; k <= 32u; k <<= 1) {
#pragma unroll
    for (uint32_t j = k >> 1; j > 0u; j >>= 1) {
#pragma unroll
      for (uint32_t i = 0; i < 32u; ++i) {
        const uint32_t l = i ^ j;
        if (l > i) {
          const bool up = (i & k) == 0u;
          const uint32_t lo = a[i] < a[l] ? a[i] : a[l], hi = a[i] < a[l] ? a[l] : a[i];
          a[i] = up ? lo : hi;
          a[l] = up ? hi : lo;
        }
      }
    }
  }
}
// d (a list line) with its ids sorted ascending; the count dword and the spare dwords are left alone
__device__ __forceinline__ void list_sort_line(uint32_t (&d)[16]) {
  uint32_t a[32];
#pragma unroll
  for (uint32_t j = 0; j < 32u; ++j) a[j] = j < kListCap ? list_id(d, j) : kListNone;
  sort_ids(a);
#pragma unroll
  for (uint32_t q = 0; q < 4u; ++q)          // dword 4q + t holds ids j = q + 8t (low half) and q + 8t + 4 (high half), t = 0..2
#pragma unroll
    for (uint32_t t = 0; t < 3u; ++t) d[4u * q + t] = a[q + 8u * t] | (a[q + 8u * t + 4u] << 16);
}
__device__ __forceinline__ void list_reset_line(uint32_t (&d)[16]) {
#pragma unroll
  for (int i = 0; i < 16; ++i) d[i] = i == 3 ? 0u : 0xFFFFFFFFu;
}

// Rebuild the short list of a slot from its dense row, ids ascending (one wavefront; v = this lane's word of the row).  Returns
// the member count; when it exceeds kListCap only the count is written (the set stays in its row).
template <typename LW>
__device__ __forceinline__ uint32_t list_rebuild(uint32_t* lists, uint32_t slot, LW v, uint32_t lane) {
  uint32_t total = (uint32_t)__builtin_popcountll((unsigned long long)v);
  for (uint32_t d = 32; d; d >>= 1) total += (uint32_t)__shfl_xor((int)total, (int)d);
  uint32_t* L = lists + (size_t)slot * kListDwords;
  if (lane < kListDwords) L[lane] = lane == 3u ? total : 0xFFFFFFFFu;
  if (total > kListCap || total == 0u) return total;
  __threadfence();                                          // the 2-byte stores below land on top of the reset line
  uint32_t base = 0;
  for (uint32_t j = 0; j < 8u * (uint32_t)sizeof(LW); ++j) {          // pod = j * 64 + lane: ascending = bit index first, lane second
    const unsigned long long b = __ballot((v >> j) & 1);
    if (b == 0ull) continue;
    if ((v >> j) & 1) {
      const uint32_t p = base + (uint32_t)__builtin_popcountll(b & ((1ull << lane) - 1ull));
      ((uint16_t*)L)[list_pos(p)] = (uint16_t)(j * 64u + lane);
    }
    base += (uint32_t)__builtin_popcountll(b);
  }
  return total;
}

#ifdef EPPK_MAIN_UNIT
// every list empty: count 0, ids 0xFFFF
__global__ void lists_fill_kernel(uint32_t* lists, size_t n_dwords) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_dwords; i += (size_t)gridDim.x * blockDim.x)
    lists[i] = (i & (kListDwords - 1u)) == 3u ? 0u : 0xFFFFFFFFu;
}
#endif

// ---- stamps as TAGS, set ids beside them: the META dword of a key (round 4: tags in a header word; round 6: protocol v5) ------------
// A key's stamp (SEMANTICS.md 6a: the index epoch of its last insert) lives in the bucket line every look-up of the key loads anyway,
// as the top byte of the key's meta dword (kBucket): 0 = no valid pod list behind this word (empty, tombstone, or a claim in progress),
// else 1 + (stamp - 1) % 255.  The low 24 bits are the key's SET ID (kSidSets).
//   * The claimer of a new key writes {tag, pod} with ONE 4-byte store, once its list store has been acknowledged: tag != 0 stays the
//     "ready" signal of the insert protocol, and a single-pod set is known from the bucket line alone.
//   * A restamp is a plain BYTE store of the top byte (every writer of a launch writes the same value); a set that changes gets its
//     id bits raised to kSidNone by an atomic OR and its slot onto the work list of index_canon_kernel, which runs behind every update
//     launch and gives the set its id again: byte stores and the 32-bit atomics do not disturb each other.
// Ages: age = (tag(epoch) - tag) mod 255 is the true age of a live key as long as that is at most 254 epochs, which
// eppk_index_advance_epoch enforces (SEMANTICS.md 6a "window": a hash stamped 255 epochs ago or more is evicted by the tick).
// TAG 0 IS THE "NOT READY" SIGNAL, so an eviction victim's list line does not have to be reset: a plain single-pod victim keeps its
// stale line -- {old pod, count 1}, every other position 0xFFFF -- and the next claimer's ONE 16-byte store of {pod, count 1} makes
// it a canonical line again; only victims with a set id >= kSidSets (two pods or more, or the dense row: rare, the hot prefixes) are
// reset in full.  Eviction: 68 -> 40 us per Mi victims (scripts/micro/claimcost2.hip).
// The two reserved rows (hashes 0 / ~0: no bucket, no meta) keep exact u32 stamps (rstamps[2]) and the round-3 list protocol.
constexpr uint32_t kTagMod = 255u;
__host__ __device__ __forceinline__ constexpr uint32_t tag_of_epoch(uint32_t epoch) { return 1u + (epoch - 1u) % kTagMod; }     // epoch >= 1
__device__ __forceinline__ uint32_t tag_age(uint32_t cur_tag, uint32_t tag) { return (cur_tag + kTagMod - tag) % kTagMod; }    // both in 1..255
__device__ __forceinline__ uint32_t* meta_ptr(uint64_t* keys, uint32_t slot) { return (uint32_t*)keys + meta_dword(slot); }
__device__ __forceinline__ uint32_t meta_tag(uint32_t meta) { return meta >> 24; }
__device__ __forceinline__ bool meta_fat(uint32_t meta) { return (meta & kSidMask) >= kSidSets; }      // more than one listed pod (or: was, until index_canon_kernel has looked)
__device__ __forceinline__ void tag_store(uint64_t* keys, uint32_t slot, uint32_t tag) {       // (agent scope, like the atomics around it)
  __hip_atomic_store((uint8_t*)meta_ptr(keys, slot) + 3, (uint8_t)tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t meta_load_coherent(const uint64_t* keys, uint32_t slot) {
  return __hip_atomic_load((const uint32_t*)keys + meta_dword(slot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the meta dword of the slot a scanning lane looks at (lane = word of the table, 64 per step: the three meta words of a bucket sit in
// the first three lanes of its group of eight): k = the word this lane loaded
__device__ __forceinline__ uint32_t meta_of_lane(unsigned long long k, uint32_t lane) {
  const uint32_t sub = lane & (kBucket - 1u);
  const uint32_t d = sub >= kKeySub0 ? sub - 2u : 1u;                  // dword of the bucket (1..5); lanes of meta words: any
  const int src = (int)((lane & ~(kBucket - 1u)) + (d >> 1));
  const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)k, src), hi = (uint32_t)__shfl((int)(uint32_t)(k >> 32), src);
  return (d & 1u) ? hi : lo;
}

// ---- index counters ------------------------------------------------------------------------------------------------
// Live keys, non-empty words, dropped inserts and "evicted by this launch" are SHARDED over kIxShards cache lines
// (ixc[shard * 8 + field]): a same-address atomic serialises at ~12 ns, and a post-pick update of a 64k x 32-block batch bumps
// the counters from 32 768 wavefronts -- on one address that alone took 1.2 ms of the 3 ms the kernel needed (round 2 profile);
// spread over 32 lines it still was half of the kernel (32 768 wavefronts x 2-3 atomics on 32 lines, and every workgroup's
// ix_budget reads those lines meanwhile: scripts/micro/insertbreak.hip, profiles/r02_micro_insertbreak.txt).  Now the insert and
// evict kernels bump them once per WORKGROUP (LDS first) and there are 64 shards.  Readers (host, capacity test) sum the shards.
constexpr uint32_t kIxShards = 64u;
constexpr uint32_t kIxLive = 0u, kIxWords = 1u, kIxDropped = 2u, kIxEvicted = 3u, kIxReserved = 4u;

// The capacity test of an insert launch.  index_budget_kernel (one wavefront, right in front of the insert launch on its stream) sums
// the shards ONCE and leaves the verdict in IxLaunch; every workgroup of the insert launch reads that one read-only line instead of
// gathering the 64 shard lines itself while other workgroups hammer them with their end-of-workgroup atomics (8 192 workgroups x 64
// lines were HALF of the kernel: profiles/r03_h_micro_insert_prologue.txt).  Capacity: at most `limit` (= slots / 2) live keys and
// 3/4 of the words non-empty; `left` = what the launch may still add under both.  Two regimes:
//   safe      every pair of this LAUNCH fits even if each brings a new key: nobody looks at a counter again;
//   per key   the launch as a whole might not fit: `left` is dealt out over the 64 shards' kIxReserved words (share(s) = left / 64,
//             the remainder one each to the first shards: the shares add up to `left` exactly) and a thread BOOKS its key -- one
//             atomic on a shard that still has room, starting with its workgroup's -- before it claims a word for it (and gives the
//             booking back when the key turns out to be there already).  The launch admits exactly `left` new keys, the rest is
//             dropped and counted: EPPK_ERR_INDEX_FULL stays exact for a caller that fills a table to the brim.  One sharded atomic
//             per NEW key -- the per-key path used to READ all the counters per key, 4.1 ms instead of 0.56 ms per Mi new keys for
//             every launch into a table more than half-way to its limit (profiles/r02_micro_insertbreak.txt).
//             (Booking in bulk per workgroup -- the keys its first look found missing -- was tried: pairs of one key sit in many
//             lanes, the bookings over-state the need several times over and crowd out real keys near the limit.)
struct IxLaunch {                 // written by index_budget_kernel, read-only while the insert kernel runs
  long long left;                 // min(limit - live, 3/4 slots - words) when the launch started
  uint32_t safe, pad;
};
__device__ __forceinline__ long long ix_share(const IxLaunch* il, uint32_t shard) {
  return il->left <= 0 ? 0ll : il->left / (long long)kIxShards + ((long long)shard < il->left % (long long)kIxShards ? 1ll : 0ll);
}
// One key, in per-key mode: book it on the first shard (from `start` on) that has room.  kIxShards as result: no room anywhere.
__device__ __forceinline__ uint32_t ix_book_one(unsigned long long* ixc, const IxLaunch* il, uint32_t start) {
  for (uint32_t t = 0; t < kIxShards; ++t) {
    const uint32_t sh = (start + t) & (kIxShards - 1u);
    const long long share = ix_share(il, sh);
    if (share <= 0) continue;
    if ((long long)__hip_atomic_load(&ixc[sh * 8u + kIxReserved], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= share) continue;
    const long long before = (long long)atomicAdd(&ixc[sh * 8u + kIxReserved], 1ull);
    if (before + 1 <= share) return sh;
    atomicAdd(&ixc[sh * 8u + kIxReserved], (unsigned long long)(0ll - 1ll));
  }
  return kIxShards;
}

#ifdef EPPK_MAIN_UNIT
__global__ void index_budget_kernel(unsigned long long* ixc, uint32_t limit, uint32_t slots, unsigned long long n_items, IxLaunch* out) {
  const uint32_t l = threadIdx.x;          // one wavefront, lane = shard
  unsigned long long lv = __hip_atomic_load(&ixc[l * 8u + kIxLive], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  unsigned long long wd = __hip_atomic_load(&ixc[l * 8u + kIxWords], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  ixc[l * 8u + kIxReserved] = 0ull;
  for (int off = 32; off >= 1; off >>= 1) { lv += __shfl_xor((long long)lv, off); wd += __shfl_xor((long long)wd, off); }
  if (l == 0u) {
    const long long live = (long long)lv, words = (long long)wd;     // (sums over the shards: a single shard may read "negative", removals land anywhere)
    long long left = (long long)limit - live;
    const long long left_w = (long long)words_cap(slots) - words;
    if (left_w < left) left = left_w;
    out->left = left;
    out->safe = (live >= 0 && (unsigned long long)live + n_items < (unsigned long long)limit && (unsigned long long)words + n_items < (unsigned long long)words_cap(slots)) ? 1u : 0u;
  }
}
#endif

// Work list of the slots whose pod list an update launch has changed: index_canon_kernel, behind the launch, brings each list back into
// ascending order and gives the set its id again (kSidSets).
// wl[0] / wl[1] = two alternating cursors (a launch appends through one; its canon pass zeroes the other for the next launch),
// wl[2] = "entries were lost" (the pass then walks the whole table), wl[4 ..] = slots.
struct SortWl {
  uint32_t* wl;
  uint32_t cap;       // entries
  uint32_t which;     // cursor of this launch (0 / 1)
};

// ---- the SET TABLE: interned, immutable copies of sorted pod lists (kSidSets) ---------------------------------------------------------
// An open-addressing table of 64-byte lines in the format of a list line (ids ascending, count in dword 3, unused positions 0xFFFF),
// keyed by content: the set id of a list is kSidSets + the line that holds its copy.  A line is written once -- claimed by a
// compare-and-swap on its count dword (0 -> kListBusy), filled, published by storing the count -- and never changes afterwards, so
// whoever holds an id may read its line with ordinary cached loads, in any later launch.  Lines are never freed one by one: when the
// table is half full of (mostly dead) sets the library clears it and has index_canon_kernel intern every listed set again
// (eppk.hip: sets_rebuild).  A set that finds no free line within kSetProbes gets no id (kSidNone): correct, only slower to pick from.
struct SetTab {
  uint32_t* lines;    // [mask + 1][16]
  uint32_t mask;
  uint32_t* ctl;      // [0] lines claimed so far, [1] sets that found no line (device memory)
};
constexpr uint32_t kSetProbes = 32u;
__device__ __forceinline__ uint32_t set_hash(const uint32_t (&d)[16]) {
  uint32_t h = d[3] * kHomeMul;
#pragma unroll
  for (int t = 0; t < 16; ++t)
    if ((t & 3) != 3) { h = (h ^ d[t]) * 0x85EBCA6Bu; h ^= h >> 13; }
  return h;
}
// The id of the canonical list line d (2 .. kListCap ids ascending, count in d[3]).  MUST BE CALLED BY EVERY LANE OF THE WAVEFRONT (`want`
// = this lane has a list): lanes of one wavefront often bring the SAME new set (the blocks of a prefix, listed one after the other), one
// of them claims the line and the others must find it there -- they wait for its count in the loop's own trip structure (the claimer
// fills and publishes inside the trip in which it won, ahead of everybody's next look; a lane spinning in a loop of its own would keep
// its wave-mate from ever getting there: NEXT.md "rules learnt"), and the loop's condition is wave-uniform.
__device__ __forceinline__ uint32_t set_intern(const SetTab& st, const uint32_t (&d)[16], bool want) {
  const uint32_t c = d[3];
  uint32_t pos = set_hash(d) & st.mask, n = 0u, spins = 0u, result = kSidNone;
  bool done = !want;
  while (__any(!done)) {
    uint32_t* S = st.lines + (size_t)pos * kListDwords;
    uint32_t cnt = 1u;
    if (!done) cnt = __hip_atomic_load(&S[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bool won = false;
    if (!done && cnt == 0u) {
      const uint32_t old = atomicCAS(&S[3], 0u, kListBusy);
      won = old == 0u;
      cnt = won ? kListBusy : old;
    }
    if (won) {
#pragma unroll
      for (int t = 0; t < 16; ++t)
        if (t != 3) __hip_atomic_store(&S[t], (t & 3) == 3 ? 0xFFFFFFFFu : d[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the ids have reached memory before the count says so
    if (won) {
      __hip_atomic_store(&S[3], c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      atomicAdd(&st.ctl[0], 1u);
      result = kSidSets + pos;
      done = true;
    }
    if (!done) {
      if (cnt & kListBusy) {                                // somebody is filling this line: look again in the next trip
        if (++spins > (1u << 16)) { done = true; atomicAdd(&st.ctl[1], 1u); }
      } else {
        bool eq = cnt == c;
        if (eq) {
          uint32_t e[16];
          load_line16<true>(S, e);
#pragma unroll
          for (int t = 0; t < 16; ++t)
            if ((t & 3) != 3) eq = eq && e[t] == d[t];
        }
        if (eq) { result = kSidSets + pos; done = true; }
        else {
          pos = (pos + 1u) & st.mask;
          if (++n >= kSetProbes) { done = true; atomicAdd(&st.ctl[1], 1u); }
        }
      }
    }
  }
  return result;
}
// One slot whose list has changed: lock (count dword: compare-and-swap count -> count | kListBusy), sort, intern, unlock, and the id into
// the slot's meta dword.  EVERY LANE OF THE WAVEFRONT calls (set_intern); `active` = this lane has a slot.  A slot can be on the work list
// several times (several appends in one launch): a lane that finds the lock taken leaves the list to its owner; one that comes after the
// owner has finished does it all again, harmlessly (the same line of the set table answers).  Reserved rows (no meta): sorted only.
// `all` = the pass walks the whole table (lost work-list entries, or a rebuild of the set table): `slot` is any word -- meta words,
// empty words and tombstones are skipped, and so are single-pod sets (their id never needs a line).
__device__ __forceinline__ void canon_slot(uint64_t* keys, uint32_t* lists, uint32_t slots, const SetTab& st, uint32_t slot, bool active, bool all) {
  bool act = active;
  if (act && all && slot < slots) {
    const uint64_t k = is_key_word(slot) ? keys[slot] : 0ull;
    act = k != 0ull && k != kTomb && meta_fat(meta_load_coherent(keys, slot));
  }
  uint32_t* L = lists + (size_t)slot * kListDwords;
  uint32_t cnt = act ? __hip_atomic_load(&L[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
  bool mine = act && cnt >= 1u && cnt <= kListCap;          // (0: emptied meanwhile; > kListCap: the dense row, no id; busy bit: another lane is on it)
  if (mine) mine = atomicCAS(&L[3], cnt, cnt | kListBusy) == cnt;
  uint32_t d[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) d[t] = 0u;
  if (mine) {
    load_line16(L, d);
    if (cnt >= 2u) {
      list_sort_line(d);
      d[3] = cnt | kListBusy;
      store_line16(L, d);
    }
    d[3] = cnt;
  }
  const uint32_t sid_set = set_intern(st, d, mine && cnt >= 2u && slot < slots);
  if (mine) {
    __threadfence();
    __hip_atomic_store(&L[3], cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (slot < slots) {
      const uint32_t sid = cnt == 1u ? (d[0] & 0xFFFFu) : sid_set;
      uint32_t* mp = meta_ptr(keys, slot);
      const uint32_t m = __hip_atomic_load(mp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(mp, (m & ~kSidMask) | sid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // (nothing else writes the index while a canon pass runs)
    }
  }
}

// Must be called by EVERY lane of the wavefront (`active` = this lane has a pair): the counters are bumped once per wavefront
// (ballot + popcount).  Capacity: at most `limit` (= slots/2) live keys and 3/4 of the words non-empty.  A launch whose pairs all
// fit (IxLaunch::safe, the common case) inserts without looking at the counters; otherwise every new key is booked before it is
// claimed (IxLaunch, above): the launch admits exactly what was left when it started.
//
// A key goes into the first FREE word (empty, or a tombstone left by an eviction) of its bucket chain -- home bucket, then
// the following buckets for as long as the overflow flags say the chain continues -- but only after the whole chain has
// been searched for the key itself (a tombstone may sit in front of it).  Free words only disappear while an insert kernel
// runs (evictions are separate launches), so every inserter of one key converges on the same word: no duplicates.
// Every insert stamps the key with the index epoch: its TAG in the bucket header (ageing: index_evict_kernel).
// Memory round trips per pair: one for the whole home bucket (its 8 words are loaded together: the key AND its tag), one CAS when the
// key is new, one for the list line (skipped for a key this thread just claimed), one for the list CAS.
// `known_only`: the pick kernel has already seen this pair's key in the index with the picked pod on its list (pick_quad_kernel's
// learn word): all that is left to do is the stamp -- one bucket look-up, no list access.
template <typename LW>
__device__ __forceinline__ void index_insert_one(uint64_t* keys, void* bitmaps, uint32_t* lists, uint32_t* rstamps, uint32_t slots, uint32_t shift,
                                                 uint32_t limit, uint32_t epoch, unsigned long long* ixc, const IxLaunch* il, unsigned long long* s_tmp,
                                                 uint64_t h, uint32_t pod, bool active, const LW* act, const SortWl& sw, uint32_t* status, bool known_only = false) {
  // a hole of the current snapshot has no cache to record: the pair is ignored (SEMANTICS.md §6b; act == null: no snapshot yet)
  if (active && act && !((act[pod & 63u] >> (pod >> 6)) & 1)) active = false;
  uint32_t slot = kNotFound;
  bool newkey = false, newword = false, stop = false;
  const bool reserved_hash = h == 0 || h == kTomb;
  const uint32_t bmask = slots / kBucket - 1u;
  unsigned long long* K = (unsigned long long*)keys;
  uint32_t free_slot = kNotFound;
  unsigned long long free_val = 0ull;
  uint32_t found_tag = 0u;                 // the tag of `slot` as the search saw it (0: unknown / not ready -- looked at again, coherently)
  // One walk of the key's bucket chain: `slot` when the key is there, else the first free word (`free_slot`, holding `free_val`).
  auto search = [&](bool coherent) {
    uint32_t bkt = home_bucket(h, shift);
    free_slot = kNotFound;
    free_val = 0ull;
    found_tag = 0u;
    bool chain_end = false;
    for (uint32_t n = 0; n <= bmask && slot == kNotFound && !chain_end; ++n) {
      unsigned long long* kb = K + (size_t)bkt * kBucket;
      unsigned long long w[kBucket];
      uint32_t q[16];
      {
        if (coherent) load_line16<true>((const uint32_t*)kb, q);
        else load_line16<false>((const uint32_t*)kb, q);
#pragma unroll
        for (int i = 0; i < (int)kBucket; ++i) w[i] = ((unsigned long long)q[2 * i + 1] << 32) | q[2 * i];
      }
#pragma unroll
      for (uint32_t i = kKeySub0; i < kBucket; ++i) {
        if (slot != kNotFound || chain_end) continue;
        const unsigned long long k = w[i];
        if (k == (unsigned long long)h) { slot = bkt * kBucket + i; found_tag = meta_tag(q[i - 2u]); continue; }
        if ((k == 0ull || k == (unsigned long long)kTomb) && free_slot == kNotFound) { free_slot = bkt * kBucket + i; free_val = k; }
        if (k == 0ull) chain_end = true;                      // buckets fill front to back: nothing lives behind an empty word
      }
      if (slot != kNotFound || chain_end) break;
      if (w[0] & 1ull) { bkt = (bkt + 1) & bmask; continue; }   // chain continues
      if (free_slot != kNotFound) break;                      // chain ends here and a tombstone is free
      atomicOr(&kb[0], 1ull);                                 // full bucket, no free word anywhere: extend the chain
      bkt = (bkt + 1) & bmask;
    }
    if (slot == kNotFound && free_slot == kNotFound) stop = true;   // walked the whole table
  };
  // (1) the first look (ordinary cached loads: load_line16)
  if (active && !reserved_hash) search(false);
  if (reserved_hash || slot == kNotFound) known_only = false;      // (the pick kernel saw the key; should it be gone, the pair takes the whole path)
  // (2) the capacity regime (IxLaunch above): safe launch-wide, or every new key booked before it is claimed
  const bool safe = il->safe != 0u;        // (uniform over the launch)
  const uint32_t my_shard = blockIdx.x & (kIxShards - 1u);
  // (3) claim
  uint32_t booked_on = kIxShards;          // per-key regime: the shard this lane's key is booked on
  // The pairs of one new key often sit in neighbouring threads (a block cached on many pods: hundreds of pairs of one hash in a row).
  // In the per-key regime only ONE lane per key and wavefront books and claims; the others look again afterwards and find the key.
  bool follower = false;
  if (!safe) {
    const uint32_t lane = threadIdx.x & 63u;
    const bool needy = active && !reserved_hash && slot == kNotFound && !stop;
    unsigned long long todo = __ballot(needy);
    while (todo) {
      const int first = __builtin_ctzll(todo);
      const uint64_t hf = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(h >> 32), first) << 32) | (uint32_t)__shfl((int)(uint32_t)h, first);
      const unsigned long long same = __ballot(needy && h == hf);
      if (needy && h == hf && (int)lane != first) follower = true;
      todo &= ~same;
    }
  }
#ifndef EPPK_BOOK_TRIES
#define EPPK_BOOK_TRIES 256
#endif
  constexpr uint32_t kBookTries = EPPK_BOOK_TRIES;      // attempts to book a new key in a launch that may not fit as a whole, a short sleep between them
  uint32_t book_tries = 0;
  auto claim = [&]() {
    while (slot == kNotFound && !stop) {
      if (reserved_hash) {
        // The two reserved hashes have rows of their own behind the table (no bucket, no search, no word of the table): the key word
        // is 0 or 1.  They are NOT booked: the capacity limit protects the table's load factor, which these two rows are no part of
        // (they count as live hashes; a table filled to its limit may therefore hold limit + 2).
        const uint32_t rrow = h == 0 ? slots : slots + 1u;
        const unsigned long long was = atomicExch((unsigned long long*)&keys[rrow], 1ull);
        newkey = was == 0ull;
        slot = rrow;
        break;
      }
      if (!safe && booked_on == kIxShards) {                  // the key needs a booking before it may claim a word
        // No room may be a passing state: threads with pairs of the same new key book as well, one claims, the others give their
        // bookings back -- so look for the key again (coherently) and try again before the pair is dropped.  ONE attempt per trip of
        // this loop: the lane that holds the surplus booking may sit in THIS wavefront, and it gives the booking back further down
        // in the loop body -- a lane that spun here for its 64 tries never let it get there, and a launch that filled a table to
        // exactly its limit dropped pairs that fit (fuzz campaign, seeds beyond the suite's: keys == limit).
        booked_on = ix_book_one(ixc, il, my_shard);
        if (booked_on == kIxShards) {
          if (++book_tries >= kBookTries) { stop = true; break; }   // the launch has admitted all it may: dropped
          __builtin_amdgcn_s_sleep(16);
          search(true);
          continue;
        }
      }
      const unsigned long long seen = atomicCAS(&K[free_slot], free_val, (unsigned long long)h);
      if (seen == free_val) { slot = free_slot; newkey = true; newword = free_val == 0ull; }
      else if (seen == (unsigned long long)h) slot = free_slot;
      else search(true);                                    // somebody else took the word for another key -> search again, coherently
      if (slot != kNotFound && !newkey && booked_on != kIxShards) {      // the key is there after all: the booking back AT ONCE (a lane of
        atomicAdd(&ixc[booked_on * 8u + kIxReserved], (unsigned long long)(0ll - 1ll));   // this wavefront may be waiting for it above)
        booked_on = kIxShards;
      }
    }
  };
  if (active && !follower) claim();
  if (follower) {                                           // (after the leaders of this wavefront: the key is there now, as a rule)
    search(true);
    claim();
  }
  if (active) {
    if (booked_on != kIxShards && !newkey) atomicAdd(&ixc[booked_on * 8u + kIxReserved], (unsigned long long)(0ll - 1ll));   // the key was there after all
  }
  const unsigned long long nk = __ballot(newkey), nw = __ballot(newword), dropped = __ballot(active && slot == kNotFound);
  __shared__ unsigned int s_cnt[3];
  if (safe) {              // (uniform over the launch) the common case: nobody reads the counters while the kernel runs
    if (threadIdx.x == 0u) { s_cnt[0] = 0u; s_cnt[1] = 0u; s_cnt[2] = 0u; }
    __syncthreads();
    if ((threadIdx.x & 63u) == 0u) {
      if (nk) atomicAdd(&s_cnt[0], (unsigned int)__builtin_popcountll(nk));
      if (nw) atomicAdd(&s_cnt[1], (unsigned int)__builtin_popcountll(nw));
      if (dropped) atomicAdd(&s_cnt[2], (unsigned int)__builtin_popcountll(dropped));
    }
    __syncthreads();
    if (threadIdx.x == 0u) {
      const uint32_t shard = blockIdx.x & (kIxShards - 1u);
      if (s_cnt[0]) atomicAdd(&ixc[shard * 8u + kIxLive], (unsigned long long)s_cnt[0]);
      if (s_cnt[1]) atomicAdd(&ixc[shard * 8u + kIxWords], (unsigned long long)s_cnt[1]);
      if (s_cnt[2]) atomicAdd(&ixc[shard * 8u + kIxDropped], (unsigned long long)s_cnt[2]);
    }
  } else if ((threadIdx.x & 63u) == 0u && (nk | nw | dropped)) {
    // per-key regime: the counts once per wavefront (the bookings stay: each claimed key is one)
    const uint32_t shard = blockIdx.x & (kIxShards - 1u);
    if (nk) atomicAdd(&ixc[shard * 8u + kIxLive], (unsigned long long)__builtin_popcountll(nk));
    if (nw) atomicAdd(&ixc[shard * 8u + kIxWords], (unsigned long long)__builtin_popcountll(nw));
    if (dropped) atomicAdd(&ixc[shard * 8u + kIxDropped], (unsigned long long)__builtin_popcountll(dropped));
  }
  bool unsorted = false;               // this thread appended behind other ids: the list needs its order back
  const bool have = active && slot != kNotFound;
  const uint32_t cur_tag = tag_of_epoch(epoch);
  // (1) A key this thread claimed a moment ago.  Its word was empty (list line in the reset state) or a tombstone (list line left by
  // the previous occupant: positions 1.. are 0xFFFF by the invariant of tag 0, position 0 and the count are stale): either way the
  // first id and the count go in with ONE plain 16-byte store -- no atomic, nothing read -- and the TAG with a byte store into the
  // bucket header, but only once the list store has been acknowledged: a lane of (2), in any wavefront, reads the list as soon as it
  // sees a tag.  (Random-line atomics cost ~50 us per Mi on this GPU whatever line they hit, stores ~37; the wait costs ~2:
  // scripts/micro/claimcost2.hip N1 / N4.)  Both steps come BEFORE (2) for every lane of the wavefront: a lane of (2) may wait for
  // exactly this tag.  The two reserved rows have no header: exact stamp, and the list's count is their ready signal as in round 3.
  if (have && newkey) {
    uint32_t* L = lists + (size_t)slot * kListDwords;
    const u32x4_t first = {0xFFFF0000u | pod, 0xFFFFFFFFu, 0xFFFFFFFFu, 1u};
    // (agent scope, like the atomics around it.  The s_nop is the wait state the ISA demands between a store of more than 8 bytes
    // and a write of its data registers: the compiler cannot see into the asm, reused the first register at once for the stamp below,
    // and the list got the epoch instead of the pod -- found by tests/test_gpu_group.py)
#ifndef EPPK_DBG_NO_LISTSTORE
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(L), "v"(first) : "memory");
#endif
    if (slot >= slots) __hip_atomic_store(&rstamps[slot - slots], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // (the stores above are acknowledged before any tag goes out and before any lane of this wavefront looks at a list: no cache
  // maintenance -- a __threadfence here, buffer_wbl2 + buffer_inv per wavefront, made the kernel four times slower)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifndef EPPK_DBG_NO_STAMP
  // tag AND set id -- the one pod -- with one 4-byte store (every insert of a launch carries the same epoch)
  if (have && newkey && slot < slots) __hip_atomic_store(meta_ptr(keys, slot), (cur_tag << 24) | pod, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
  // (2) A key that was there.  Its tag first -- from the bucket line the search loaded; 0 = its claimer has not finished (or the
  // search never saw the line: the slot came out of a lost compare-and-swap): look again, coherently, until the tag is there -- the
  // claimer never waits for anybody, so it arrives.  Bounded all the same: a lane that gives up drops its pair and raises a
  // launch-status flag instead of hanging the GPU.  An older tag is brought up to date with a byte store (every writer of the launch
  // writes the same value).  Then the list line and the list protocol (list_add) -- unless the pick kernel has vouched for the pair.
  if (have && !newkey) {
    uint32_t* L = lists + (size_t)slot * kListDwords;
    bool ready = true;
    if (slot < slots) {
      uint32_t tag = found_tag, spins = 0;
      while (tag == 0u && spins < (1u << 20)) { tag = meta_tag(meta_load_coherent(keys, slot)); ++spins; }
      ready = tag != 0u;
#ifndef EPPK_DBG_NO_STAMP
      // An older tag: look again, COHERENTLY, before storing.  The first look was an ordinary cached load, and the copy of a hot bucket in
      // this XCD's L2 keeps showing the old tag to every later thread of the launch: in the first update after an epoch tick all
      // 256 pairs of every hot key stored the byte (1 Mi stores into 4096 lines: that update took 190 us instead of 106,
      // profiles/r04_closed_loop_kernel_stats_before_recheck.csv).  The coherent load sees the first store that has landed.
      if (ready && tag != cur_tag && meta_tag(meta_load_coherent(keys, slot)) != cur_tag) tag_store(keys, slot, cur_tag);
#endif
    }
    if (!ready) atomicOr(status, kStatusIndexStall);
    else if (!known_only) {
      // The list line COHERENTLY, also at the first look.  An ordinary cached load can hit a copy of the line that this XCD's L2 took
      // in earlier in the launch -- through the NEIGHBOURING slot's list: the L2 line is 128 bytes, two list lines -- i.e. from before
      // the claimer's store: the lane then does not see the claimer's pod, and when it brings the same pod it appends it a second time
      // (a duplicate id and count 2 for a one-pod set; found by a fuzz campaign over 3 600 sequences beyond the suite's seeds, three
      // times: scripts/gpu_fuzz_campaign.py).  The tag this lane has seen says the claimer's store is in memory; only a load that goes
      // there is sure to see it.  (Measured in round 3, cached against coherent first looks: no difference in the update's time.)
      uint32_t d[16];
      load_line16<true>(L, d);
      uint32_t res = 0u, pos = 0;
      if (slot >= slots) {                                  // reserved rows: the round-3 protocol (count 0 = the claimer's store is on its way)
        uint32_t spins = 0;
        while (d[3] == 0u && spins < (1u << 20)) { load_line16(L, d); ++spins; }
        if (__hip_atomic_load(&rstamps[slot - slots], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) atomicMax(&rstamps[slot - slots], epoch);
      }
      if (d[3] == 0u) atomicOr(status, kStatusIndexStall);
      else if (d[3] > kListCap) bitmap_set<LW>(bitmaps, slot, pod);         // overflowed: the row is the set
      else res = list_add<false>(L, d, pod, pos);
      if (res == 2u) list_overflow<LW>(keys, slots, bitmaps, L, slot, pod);
      // The set has changed: its id no longer names it.  kSidNone until index_canon_kernel -- behind this launch -- has put the list
      // back in order and interned it (the id is >= kSidSets from here on: the eviction resets such a list line in full).  Reserved
      // rows have no meta: their lists are only re-sorted.
      unsorted = res == 1u;
      if (res != 0u && slot < slots) atomicOr(meta_ptr(keys, slot), kSidMask);
    }
  }
  // the lists to re-sort, appended to the work list once per wavefront
  const unsigned long long um = __ballot(unsorted);
  if (um) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t base = 0;
    if (lane == (uint32_t)__builtin_ctzll(um)) base = atomicAdd(&sw.wl[sw.which], (uint32_t)__builtin_popcountll(um));
    base = (uint32_t)__shfl((int)base, __builtin_ctzll(um));
    if (unsorted) {
      const uint32_t at = base + (uint32_t)__builtin_popcountll(um & ((1ull << lane) - 1ull));
      if (at < sw.cap) sw.wl[4u + at] = slot;
      else sw.wl[2] = 1u;
    }
  }
}

template <typename LW>
__global__ void index_insert_kernel(uint64_t* keys, void* bitmaps, uint32_t* lists, uint32_t* rstamps, uint32_t slots, uint32_t shift, uint32_t limit,
                                    uint32_t epoch, unsigned long long* ixc, const uint64_t* hashes, const uint32_t* pods, uint32_t n,
                                    const LW* act, SortWl sw, uint32_t* status, const IxLaunch* il) {
  __shared__ unsigned long long s_tmp[4];
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = i < n;
  index_insert_one<LW>(keys, bitmaps, lists, rstamps, slots, shift, limit, epoch, ixc, il, s_tmp, active ? hashes[i] : 0ull, active ? pods[i] : 0u, active, act, sw, status);
}

// thread (r, i): append picks[r] to hash i of request r.
// `learn` (nullable): one word per request from pick_quad_kernel<..., LEARN> -- bits 0..7 = m, the number of leading blocks of the
// request whose keys the pick kernel found in the index (the walk of SEMANTICS.md 3 PREFIX), bits 8..23 = pick + 1, bit 31 = the picked
// pod is on the pod list of every one of the m.  Such a pair is in the index already: thread (r, i < m) only brings the key's stamp up
// to date -- one bucket look-up instead of bucket + list (the 1 Mi known pairs of a 64k x 32-block closed-loop step were 38 us of its
// 150).  0 = no information (a request that kernel deferred, or another pick route): picks[r], and every block takes the whole path.
template <typename LW>
__global__ void index_insert_picks_kernel(uint64_t* keys, void* bitmaps, uint32_t* lists, uint32_t* rstamps, uint32_t slots, uint32_t shift, uint32_t limit,
                                          uint32_t epoch, unsigned long long* ixc, const uint8_t* reqs, uint32_t stride,
                                          uint32_t max_blocks, const int32_t* picks, uint32_t n_reqs, uint32_t max_pods, uint32_t* status, const LW* act,
                                          SortWl sw, const IxLaunch* il, const uint32_t* learn) {
  __shared__ unsigned long long s_tmp[4];
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t r = (uint32_t)(t / max_blocks), i = (uint32_t)(t % max_blocks);
  bool active = r < n_reqs, known_only = false;
  int32_t pick = -1;
  uint64_t h = 0;
  uint32_t lw = 0u;
  if (active) {
    if (learn) lw = learn[r];
    pick = (lw & 0x00FFFF00u) ? (int32_t)((lw >> 8) & 0xFFFFu) - 1 : picks[r];
    const uint8_t* row = reqs + (size_t)r * stride;
    const uint32_t nb = ((const uint32_t*)row)[1];
    // a pick beyond the lane words (>= max_pods) would shift into other pods' bits: ignored and flagged, like a row whose block
    // count exceeds the row (eppk_launch_status)
    const bool bad = (pick >= 0 && (uint32_t)pick >= max_pods) || nb > max_blocks;
    if (bad && i == 0u) atomicOr(status, (uint32_t)pick >= max_pods && pick >= 0 ? kStatusBadPick : kStatusBadRow);
    active = !bad && pick >= 0 && i < nb;
    if (active) h = ((const uint64_t*)(row + 8))[i];
    known_only = active && (lw >> 31) != 0u && i < (lw & 0xFFu);
  }
  index_insert_one<LW>(keys, bitmaps, lists, rstamps, slots, shift, limit, epoch, ixc, il, s_tmp, h, (uint32_t)pick, active, act, sw, status, known_only);
}

#ifdef EPPK_MAIN_UNIT
// Behind every launch that changes pod lists (same stream): the lists on the work list go back to ascending order and get their set
// ids again (canon_slot) -- equal SETS are equal IDS (pick_quad_kernel) and equal LINES (the fast kernel's uniform route compares the
// slots' own lists bit for bit).  A lane per work-list entry; lost entries (work list full) or `force_all` (a rebuild of the set table:
// the library has just cleared it): the whole table is walked.  The last workgroup to arrive leaves the set table's counters in pinned
// host memory (`report`: lines in use, sets without a line), where the library reads them before the next update -- no synchronisation.
__global__ void index_canon_kernel(uint64_t* keys, uint32_t* lists, uint32_t slots, SetTab st, uint32_t* wl, uint32_t cap, uint32_t which, uint32_t force_all,
                                   uint32_t* report) {
  const uint32_t n_listed = wl[which] < cap ? wl[which] : cap;
  const bool lost = wl[2] != 0u || force_all != 0u;
  const uint32_t total = lost ? slots + 2u : n_listed;
  for (uint32_t base = blockIdx.x * blockDim.x; base < total; base += gridDim.x * blockDim.x) {      // (uniform per wavefront: canon_slot is called by every lane)
    const uint32_t i = base + threadIdx.x;
    const bool active = i < total;
    const uint32_t slot = !active ? 0u : (lost ? i : wl[4u + i]);
    canon_slot(keys, lists, slots, st, slot, active, lost);
  }
  // the next launch's cursor (nobody reads it before that launch), and the lost flag once every thread of this grid has read it:
  // the flag is only ever set by an update launch, so clearing it from the LAST workgroup to arrive is safe
  __shared__ uint32_t s_last;
  __syncthreads();
  if (threadIdx.x == 0u) {
    if (blockIdx.x == 0u) wl[which ^ 1u] = 0u;
    __threadfence();
    s_last = atomicAdd(&wl[3], 1u) == gridDim.x - 1u ? 1u : 0u;
  }
  __syncthreads();
  if (s_last && threadIdx.x == 0u) {
    wl[2] = 0u; wl[3] = 0u;
    if (report) {
      report[0] = __hip_atomic_load(&st.ctl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      report[1] = __hip_atomic_load(&st.ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}
#endif

// The scanning maintenance kernels below (removal, trim, ageing) walk the table with lane = word, 64 words per step and chunk: the
// three meta words of a bucket sit in the first three lanes of its group of eight (meta_of_lane).
// A slot whose pod set has become empty: the key word turns into a tombstone (reusable by later inserts; reserved rows: presence
// cleared) and its meta dword into 0 (tag 0: "no valid list behind this word") -- the caller has left the list line with
// every position from 1 on at 0xFFFF (reset, or a plain single-pod line), which is all the next claimer relies on.  These kernels run
// alone (no insert in flight): plain stores (neighbouring lanes write other dwords).
__device__ __forceinline__ void slot_bury(uint64_t* keys, uint32_t slots, uint32_t row) {
  keys[row] = row < slots ? kTomb : 0ull;
  if (row < slots) *meta_ptr(keys, row) = 0u;
}
// A listed set that a removal pass has edited (d: the line it stored, count in d[3] >= 1): a single pod is its own id; a longer list loses
// its id and goes onto the work list of the index_canon_kernel that follows the pass.
__device__ __forceinline__ void slot_reid(uint64_t* keys, uint32_t slots, uint32_t row, uint32_t count, uint32_t first_id, const SortWl& sw) {
  if (row >= slots) return;
  uint32_t* mp = meta_ptr(keys, row);
  const uint32_t m = *mp;
  if (count == 1u) { *mp = (m & ~kSidMask) | first_id; return; }
  *mp = m | kSidMask;
  const uint32_t at = atomicAdd(&sw.wl[sw.which], 1u);
  if (at < sw.cap) sw.wl[4u + at] = row;
  else sw.wl[2] = 1u;
}

// Remove pods from every set; a set that becomes empty gets its key tombstoned so that the hot path never meets a present key
// with an empty pod set.  `rm` (nullable): instead of the single `pod`, every pod whose bit is set in the lane-transposed row
// rm[64] (the holes of a snapshot: eppk_snapshot_publish scrubs them out of the index in one pass).
// A wavefront scans 64 slots per step (lane = slot: keys stream in coalesced); a listed set is edited by its lane (removed ids
// become 0xFFFF, the sorting network closes the gaps and keeps the order); overflowed sets are then taken by the whole wavefront,
// one after the other: bits cleared, and a row that is back at kListCap members or fewer returns to its list.
template <typename LW>
__global__ void index_remove_pod_kernel(uint64_t* keys, void* bitmaps, uint32_t* lists, uint32_t slots, uint32_t pod, unsigned long long* ixc,
                                        const LW* rm, SortWl sw) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  const uint32_t total = slots + 2u;
  const LW my_rm = rm ? rm[lane] : (lane == (pod & 63u) ? (LW)((LW)1 << (pod >> 6)) : (LW)0);
  auto removed = [&](uint32_t id) -> bool {
    if (id == kListNone) return false;
    if (!rm) return id == pod;
    return (id >> 6) < 8u * (uint32_t)sizeof(LW) && ((rm[id & 63u] >> (id >> 6)) & 1);
  };
  uint32_t gone = 0;
  for (uint32_t base = wave * 64u; base < total; base += nwaves * 64u) {
    const uint32_t row = base + lane;
    bool present = false;
    const uint64_t k = row < total ? keys[row] : 0ull;
    if (row < total && !(row < slots && !is_key_word(row)))       // (the meta words of a bucket are not keys)
      present = k != 0ull && !(row < slots && k == kTomb);
    bool over = false, emptied = false;
    if (present) {
      uint32_t* L = lists + (size_t)row * kListDwords;
      uint32_t d[16];
      load_line16<false>(L, d);                         // (a launch of its own: nothing else writes the index meanwhile)
      if (d[3] > kListCap) over = true;
      else {
        uint32_t nrm = 0;
#pragma unroll
        for (uint32_t j = 0; j < kListCap; ++j) {
          if (removed(list_id(d, j))) {
            d[list_dw(j)] |= list_hi(j) ? 0xFFFF0000u : 0x0000FFFFu;
            ++nrm;
          }
        }
        if (nrm) {
          list_sort_line(d);
          d[3] -= nrm;
          store_line16(L, d);
          emptied = d[3] == 0u;
          if (!emptied) slot_reid(keys, slots, row, d[3], d[0] & 0xFFFFu, sw);     // the set has changed: its id with it
        }
      }
    }
    unsigned long long om = __ballot(over);
    while (om) {
      const uint32_t v_row = base + (uint32_t)__builtin_ctzll(om);
      om &= om - 1ull;
      LW* w = (LW*)bitmaps + (size_t)v_row * 64u + lane;
      LW v = *w;
      const LW nv = (LW)(v & (LW)~my_rm);
      if (!__any(nv != v)) continue;
      uint32_t members = (uint32_t)__builtin_popcountll((unsigned long long)nv);
      for (uint32_t dd = 32; dd; dd >>= 1) members += (uint32_t)__shfl_xor((int)members, (int)dd);
      if (members <= kListCap) {                       // back to the list (or gone: slot_bury): the row returns to all-zero, the set gets an id again
        list_rebuild<LW>(lists, v_row, nv, lane);
        *w = 0;
        const unsigned long long hm = __ballot(nv != 0);
        const uint32_t first = (uint32_t)__shfl((int)(nv != 0 ? ctz_lw<LW>(nv) * 64u + lane : 0u), hm ? __builtin_ctzll(hm) : 0);   // (the ONE pod when members == 1)
        if (lane == 0 && members >= 1u) slot_reid(keys, slots, v_row, members, first, sw);
        if (members == 0u && v_row == row) emptied = true;
      } else if (nv != v) {
        *w = nv;
      }
    }
    if (emptied) slot_bury(keys, slots, row);
    gone += (uint32_t)__builtin_popcountll(__ballot(emptied));
  }
  if (lane == 0 && gone) atomicAdd(&ixc[(wave & (kIxShards - 1u)) * 8u + kIxLive], (unsigned long long)(0ull - (unsigned long long)gone));
}

// ---- per-pod capacity (SEMANTICS.md §6c; 0602-…/README.md:82: the index mimics the model servers' own LRU-bounded caches) ----
// A pod may be listed under at most `cap` hashes; what exceeds that goes oldest first, at epoch granularity: with
// age(h) = min(63, epoch - stamp(h)) and cutage(p) = the smallest b >= 1 with #{h containing p : age(h) <= b} > cap, pod p is removed
// from every hash of age >= cutage(p) (entries stamped in the current epoch always stay).  Three passes:
//   (1) index_pod_hist_kernel   hist[p][age] += 1 for every pod p of every set (a lane per listed set, the wavefront per overflowed row)
//   (2) index_pod_cut_kernel    thread per pod: cutage[p] (kNoCut = within capacity), then the lane-transposed set `over` of the pods to trim
//   (3) index_pod_trim_kernel   removes the pods with cutage <= age(set) (same shape as index_remove_pod_kernel), tombstones empty sets
constexpr uint32_t kTrimBins = 64u;
constexpr uint32_t kNoCut = 0xFFFFFFFFu;

// age (in epochs, capped at kTrimBins - 1) of the present key in `row`: from the tag in its meta dword, the reserved rows from their exact stamps
__device__ __forceinline__ uint32_t slot_age(uint32_t meta, const uint32_t* rstamps, uint32_t slots, uint32_t row, uint32_t epoch) {
  uint32_t age;
  if (row < slots) {
    const uint32_t tag = meta_tag(meta);
    age = tag ? tag_age(tag_of_epoch(epoch), tag) : 0u;
  } else {
    age = epoch - rstamps[row - slots];
  }
  return age < kTrimBins - 1u ? age : kTrimBins - 1u;
}

template <typename LW>
__global__ void index_pod_hist_kernel(const uint64_t* keys, const void* bitmaps, const uint32_t* lists, const uint32_t* rstamps, uint32_t slots, uint32_t epoch,
                                      uint32_t* hist) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  const uint32_t total = slots + 2u;
  for (uint32_t base = wave * 64u; base < total; base += nwaves * 64u) {
    const uint32_t row = base + lane;
    bool present = false;
    const uint64_t k = row < total ? keys[row] : 0ull;
    if (row < total && !(row < slots && !is_key_word(row)))
      present = k != 0ull && !(row < slots && k == kTomb);
    const uint32_t meta = meta_of_lane(k, lane);
    bool over = false;
    uint32_t age = 0;
    if (present) {
      age = slot_age(meta, rstamps, slots, row, epoch);
      uint32_t d[16];
      load_line16<false>(lists + (size_t)row * kListDwords, d);
      if (d[3] > kListCap) over = true;
      else {
#pragma unroll
        for (uint32_t j = 0; j < kListCap; ++j) {
          const uint32_t id = list_id(d, j);
          if (id != kListNone && (id >> 6) < 8u * (uint32_t)sizeof(LW)) atomicAdd(&hist[(size_t)id * kTrimBins + age], 1u);
        }
      }
    }
    unsigned long long om = __ballot(over);
    while (om) {
      const uint32_t src = (uint32_t)__builtin_ctzll(om);
      om &= om - 1ull;
      const uint32_t v_row = base + src, v_age = (uint32_t)__shfl((int)age, (int)src);
      LW v = ((const LW*)bitmaps)[(size_t)v_row * 64u + lane];
      while (v != 0) {
        const uint32_t j = ctz_lw<LW>(v);
        v = (LW)(v & (LW)(v - 1));
        atomicAdd(&hist[(size_t)(j * 64u + lane) * kTrimBins + v_age], 1u);
      }
    }
  }
}

#ifdef EPPK_MAIN_UNIT
// one workgroup of 1024 threads; over_t: 64 u64 lane words (bit j of word l = pod j*64+l has a cut)
__global__ __launch_bounds__(1024) void index_pod_cut_kernel(const uint32_t* hist, uint32_t max_pods, uint32_t cap, uint32_t* cutage, uint64_t* over_t) {
  for (uint32_t p = threadIdx.x; p < 4096u; p += blockDim.x) {
    uint32_t cut = kNoCut;
    if (p < max_pods) {
      uint64_t cum = 0;
      for (uint32_t b = 0; b < kTrimBins; ++b) {
        cum += hist[(size_t)p * kTrimBins + b];
        if (cum > cap) { cut = b < 1u ? 1u : b; break; }      // (the current epoch's entries always stay)
      }
    }
    cutage[p] = cut;
  }
  __syncthreads();
  if (threadIdx.x < 64u) {
    uint64_t w = 0;
    for (uint32_t j = 0; j < 64u; ++j)
      if (cutage[j * 64u + threadIdx.x] != kNoCut) w |= 1ull << j;
    over_t[threadIdx.x] = w;
  }
}
#endif

template <typename LW>
__global__ void index_pod_trim_kernel(uint64_t* keys, void* bitmaps, uint32_t* lists, const uint32_t* rstamps, uint32_t slots, uint32_t epoch,
                                      const uint32_t* cutage, const uint64_t* over_t, unsigned long long* ixc, unsigned long long* removed, SortWl sw) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  const uint32_t total = slots + 2u;
  const LW over_pods = (LW)over_t[lane];
  uint32_t gone = 0, pairs = 0;
  for (uint32_t base = wave * 64u; base < total; base += nwaves * 64u) {
    const uint32_t row = base + lane;
    bool present = false;
    const uint64_t k = row < total ? keys[row] : 0ull;
    if (row < total && !(row < slots && !is_key_word(row)))
      present = k != 0ull && !(row < slots && k == kTomb);
    const uint32_t meta = meta_of_lane(k, lane);
    bool over = false, emptied = false;
    uint32_t age = 0;
    if (present) {
      age = slot_age(meta, rstamps, slots, row, epoch);
      uint32_t* L = lists + (size_t)row * kListDwords;
      uint32_t d[16];
      load_line16<false>(L, d);                         // (a launch of its own: nothing else writes the index meanwhile)
      if (d[3] > kListCap) over = true;
      else {
        uint32_t nrm = 0;
#pragma unroll
        for (uint32_t j = 0; j < kListCap; ++j) {
          const uint32_t id = list_id(d, j);
          if (id != kListNone && (id >> 6) < 8u * (uint32_t)sizeof(LW) && cutage[id] <= age) {
            d[list_dw(j)] |= list_hi(j) ? 0xFFFF0000u : 0x0000FFFFu;
            ++nrm;
          }
        }
        if (nrm) {
          list_sort_line(d);
          d[3] -= nrm;
          store_line16(L, d);
          emptied = d[3] == 0u;
          pairs += nrm;
          if (!emptied) slot_reid(keys, slots, row, d[3], d[0] & 0xFFFFu, sw);
        }
      }
    }
    unsigned long long om = __ballot(over);
    while (om) {
      const uint32_t src = (uint32_t)__builtin_ctzll(om);
      om &= om - 1ull;
      const uint32_t v_row = base + src, v_age = (uint32_t)__shfl((int)age, (int)src);
      LW* w = (LW*)bitmaps + (size_t)v_row * 64u + lane;
      const LW v = *w;
      LW cand = (LW)(v & over_pods), rmw = 0;
      while (cand != 0) {
        const uint32_t j = ctz_lw<LW>(cand);
        cand = (LW)(cand & (LW)(cand - 1));
        if (cutage[j * 64u + lane] <= v_age) rmw |= (LW)((LW)1 << j);
      }
      if (!__any(rmw != 0)) continue;
      const LW nv = (LW)(v & (LW)~rmw);
      uint32_t nrm = (uint32_t)__builtin_popcountll((unsigned long long)rmw), members = (uint32_t)__builtin_popcountll((unsigned long long)nv);
      for (uint32_t dd = 32; dd; dd >>= 1) { nrm += (uint32_t)__shfl_xor((int)nrm, (int)dd); members += (uint32_t)__shfl_xor((int)members, (int)dd); }
      if (lane == src) pairs += nrm;
      if (members <= kListCap) {
        list_rebuild<LW>(lists, v_row, nv, lane);
        *w = 0;
        const unsigned long long hm = __ballot(nv != 0);
        const uint32_t first = (uint32_t)__shfl((int)(nv != 0 ? ctz_lw<LW>(nv) * 64u + lane : 0u), hm ? __builtin_ctzll(hm) : 0);
        if (lane == 0 && members >= 1u) slot_reid(keys, slots, v_row, members, first, sw);
        if (members == 0u && lane == src) emptied = true;
      } else if (rmw != 0) {
        *w = nv;
      }
    }
    if (emptied) slot_bury(keys, slots, row);
    gone += (uint32_t)__builtin_popcountll(__ballot(emptied));
  }
  for (int off = 32; off >= 1; off >>= 1) pairs += (uint32_t)__shfl_xor((int)pairs, off);
  if (lane == 0) {
    if (gone) atomicAdd(&ixc[(wave & (kIxShards - 1u)) * 8u + kIxLive], (unsigned long long)(0ull - (unsigned long long)gone));
    if (pairs) atomicAdd(removed, (unsigned long long)pairs);
  }
}

// Ageing (SEMANTICS.md §6a; 0602-…/README.md:82 "mimicking a similar cache eviction strategy of the model server (e.g., LRU)"):
// drop every key last stamped before min_epoch -- key tombstoned (reusable by later inserts), tag 0.
// A wavefront scans 64 slots per step and chunk (lane = slot: the key words stream in coalesced, and with them the bucket headers that
// hold the stamps as tags -- there is no stamp array to read).  A LANE per victim: one 8-byte store (tombstone) and one byte store
// (tag 0) into the line the scan has just read.  Its list line is NOT touched when the slot is not FAT: {old pod, count 1} with every
// other position 0xFFFF is exactly what the next claimer's 16-byte store overwrites (see "stamps as tags").  That was a whole random
// 64-byte line per victim: 68 -> 40 us per Mi victims (scripts/micro/claimcost2.hip E0 / E3, tables beyond the Infinity Cache).  FAT
// victims (two or more listed pods, or the dense row: the hot prefixes when they finally age out) get their list line reset in
// full, their row zeroed by the whole wavefront when the set lived there, their flag cleared.
// `keep` = epoch - min_epoch: a key is a victim iff its age exceeds it (-1: every key; >= 254: none -- ages are at most 254, which
// eppk_index_advance_epoch sees to).  The two reserved rows: exact stamps, and the round-3 protocol (list reset in full).
template <typename LW>
__global__ void index_evict_kernel(uint64_t* keys, void* bitmaps, uint32_t* lists, const uint32_t* rstamps, uint32_t slots, uint32_t epoch, uint32_t min_epoch,
                                   unsigned long long* ixc) {
  // kEvictChunks x 64 slots per wavefront and step, every chunk's key words requested before the first is looked at (with one chunk the
  // scan was bound by bytes in flight; measured with 1 / 4 / 8 chunks: profiles/r03_y_evict_chunks.txt).
#ifndef EPPK_EVICT_CHUNKS
#define EPPK_EVICT_CHUNKS 4
#endif
  constexpr uint32_t kEvictChunks = EPPK_EVICT_CHUNKS;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  const uint32_t total = slots + 2u;
  const uint32_t cur_tag = tag_of_epoch(epoch);
  const long long keep = (long long)epoch - (long long)min_epoch;
  uint32_t gone = 0;
  for (uint32_t base0 = wave * 64u * kEvictChunks; base0 < total; base0 += nwaves * 64u * kEvictChunks) {
    uint64_t kk[kEvictChunks];
#pragma unroll
    for (uint32_t u = 0; u < kEvictChunks; ++u) {
      const uint32_t row = base0 + u * 64u + lane;
      kk[u] = row < total ? keys[row] : 0ull;
    }
#pragma unroll
    for (uint32_t u = 0; u < kEvictChunks; ++u) {
      const uint32_t base = base0 + u * 64u;
      if (base >= total) break;                       // (uniform)
      const uint32_t row = base + lane;
      const bool header = row < slots && !is_key_word(row);
      const uint64_t k = kk[u];
      const uint32_t meta = meta_of_lane(k, lane);                      // this slot's meta dword (rows < slots)
      bool victim = false, fat = false;
      if (row < slots) {
        const uint32_t tag = meta_tag(meta);
        victim = !header && k != 0ull && k != kTomb && tag != 0u && (long long)tag_age(cur_tag, tag) > keep;
        fat = meta_fat(meta);
      } else if (row < total) {
        victim = k != 0ull && rstamps[row - slots] < min_epoch;
        fat = true;                                   // (reserved rows: always the full reset)
      }
      gone += (uint32_t)__builtin_popcountll(__ballot(victim));
      bool whole = false;
      if (victim) {
        if (fat) {
          u32x4_t* Lp = (u32x4_t*)(lists + (size_t)row * kListDwords);
          whole = Lp[0].w > kListCap;                 // the set lives in its dense row
          const u32x4_t e0 = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u}, e1 = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
          Lp[0] = e0; Lp[1] = e1; Lp[2] = e1; Lp[3] = e1;
        }
        slot_bury(keys, slots, row);
      }
      unsigned long long vm = __ballot(whole);          // dense sets: the whole row, by the wavefront
      while (vm) {
        const uint32_t v = base + (uint32_t)__builtin_ctzll(vm);
        vm &= vm - 1ull;
        ((LW*)bitmaps)[(size_t)v * 64u + lane] = 0;
      }
    }
  }
  if (lane == 0 && gone) {
    const uint32_t shard = (wave & (kIxShards - 1u)) * 8u;
    atomicAdd(&ixc[shard + kIxLive], (unsigned long long)(0ull - (unsigned long long)gone));
    atomicAdd(&ixc[shard + kIxEvicted], (unsigned long long)gone);   // evicted by this launch (the synchronous entry point zeroes it first)
  }
}

// Diagnostic (eppk_index_selfcheck): counts the slots that break an invariant of the index (the list at the head of this section, and
// "stamps as tags, set ids beside them").  A wavefront per slot: its row word per lane, its list dwords in lanes 0..15.  By state of the key word:
//   empty          meta 0 (no tag, no id), list line in the reset state (count 0, every id 0xFFFF), row all-zero
//   tombstone      meta 0, row all-zero; of the list line only what the next claimer relies on: count <= 1 and every position
//                  from 1 on 0xFFFF (position 0 may still hold the previous occupant's pod)
//   present        tag != 0; a non-empty set: listed (count <= 24: ids valid, unique, strictly ascending, nothing behind the count,
//                  row all-zero) or dense (count > 24: more than 24 members in the row).  Its SET ID: one member => the id IS that pod;
//                  2..24 members => kSidNone or the id of a published line of the set table that equals the list dword for dword;
//                  dense => kSidNone
//   reserved rows  (no meta) absent: list reset, row zero; present: as above without tag / id
template <typename LW>
__global__ void index_selfcheck_kernel(const uint64_t* keys, const void* bitmaps, const uint32_t* lists, uint32_t slots, uint32_t sets_mask, unsigned long long* bad) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  uint32_t nbad = 0;
  for (uint32_t row = wave; row < slots + 3u; row += nwaves) {
    const bool header = row < slots && !is_key_word(row);
    if (header) continue;                                                  // (wave-uniform: the meta words are checked through their slots)
    const uint64_t k = row < slots + 2u ? keys[row] : 0ull;
    const bool tomb = row < slots && k == kTomb;
    const bool present = row < slots + 2u && k != 0ull && !tomb;
    const uint32_t meta = row < slots ? ((const uint32_t*)keys)[meta_dword(row)] : 0u;
    const uint32_t tag = meta_tag(meta), sid = meta & kSidMask;
    const bool fat = row < slots && meta_fat(meta);
    const LW v = ((const LW*)bitmaps)[(size_t)row * 64u + lane];
    uint32_t members = (uint32_t)__builtin_popcountll((unsigned long long)v);
    for (uint32_t d = 32; d; d >>= 1) members += (uint32_t)__shfl_xor((int)members, (int)d);
    const uint32_t* L = lists + (size_t)row * kListDwords;
    const uint32_t count = L[3];
    // lane j < 24 looks at id j and its predecessor
    const uint32_t j = lane < kListCap ? lane : 0u;
    const uint32_t id = ((const uint16_t*)L)[list_pos(j)], prev = j ? ((const uint16_t*)L)[list_pos(j - 1u)] : 0u;
    bool ok = true;
    if (lane < kListCap) {
      if (tomb) ok = lane == 0u || id == kListNone;                        // positions 1.. (position 0: whatever the previous occupant left)
      else if (count <= kListCap) {
        if (lane < count) ok = id != kListNone && (id >> 6) < 8u * (uint32_t)sizeof(LW) && (lane == 0u || prev < id);   // valid, strictly ascending
        else ok = id == kListNone;
      }
    } else if (lane < kListCap + 3u) {
      ok = L[7u + 4u * (lane - kListCap)] == 0xFFFFFFFFu;                  // the spare dwords 7, 11, 15
    }
    const unsigned long long notok = __ballot(!ok);
    uint32_t why = notok ? 1u : 0u;                                        // bit 0: a list entry (ballot in the record), then per state
    if (tomb) why |= (count > 1u ? 2u : 0u) | (members != 0u ? 4u : 0u);                 // tombstone: a plain line at most, all-zero row
    else if (!present) why |= (count != 0u ? 2u : 0u) | (members != 0u ? 4u : 0u);       // empty / absent reserved row: reset list, all-zero row
    else if (count <= kListCap) why |= (count == 0u ? 8u : 0u) | (members != 0u ? 16u : 0u);   // listed: non-empty, all-zero row
    else why |= members <= kListCap ? 32u : 0u;                            // dense: more than kListCap members in the row
    if (row < slots) {
      if (!present) why |= meta != 0u ? 128u : 0u;                                       // no key: no tag, no id
      else {
        why |= tag == 0u ? 128u : 0u;                                                     // a key: a tag
        if (count == 1u) why |= sid != (L[0] & 0xFFFFu) ? 64u : 0u;                       // one pod: the id is that pod
        else if (count > kListCap) why |= sid != kSidNone ? 64u : 0u;                     // dense: no id
        else if (sid != kSidNone) {                                                       // a list: no id, or a line of the set table that equals it
          bool same = sid >= kSidSets && sid - kSidSets <= sets_mask;
          if (same && lane < kListDwords && (lane & 3u) != 3u) same = lists[((size_t)slots + 4u + (sid - kSidSets)) * kListDwords + lane] == L[lane];
          if (same && lane == 3u) same = lists[((size_t)slots + 4u + (sid - kSidSets)) * kListDwords + 3u] == count;
          why |= __ballot(!same) ? 64u : 0u;
        }
      }
    }
    if (why) {
      ++nbad;
      // the first eight offenders in full, for eppk_index_selfcheck's EPPK_SELFCHECK_VERBOSE: bad[2 + 24 k ..] = row, why, key, members, ballot, list[16]
      unsigned long long at = 0;
      if (lane == 0) at = atomicAdd(&bad[1], 1ull);
      at = (unsigned long long)__shfl((long long)at, 0);
      if (at < 8ull) {
        unsigned long long* rec = bad + 2u + 24u * at;
        if (lane == 0) { rec[0] = row; rec[1] = why | ((unsigned long long)tag << 32) | ((unsigned long long)fat << 40); rec[2] = k; rec[3] = members; rec[4] = notok; }
        if (lane < kListDwords) rec[5u + lane] = L[lane];
      }
    }
  }
  if (lane == 0 && nbad) atomicAdd(bad, (unsigned long long)nbad);
}


// The post-route index update of ONE small batch by the resident workgroup that has just answered it (pick_resident_kernel<..., LEARN>):
// what learn_picks (eppk.hip) launches as index_budget_kernel + index_insert_picks_kernel + index_lists_sort_kernel, here by the
// 1024 threads of the workgroup, in place.  Nothing else touches the index meanwhile (the library orders every other reader and
// writer behind the unit's `updated` word), so the protocol of index_insert_one holds as in a launch of its own.
template <typename LW>
__device__ __noinline__ void resident_learn_update(const ResidentArgs* a, const ResidentBuf& rb, uint32_t n) {
  __shared__ IxLaunch s_il;
  __shared__ unsigned long long s_tmp[4];
  const uint32_t max_blocks = a->max_blocks, slots = a->ix.slots;
  const uint32_t total = n * max_blocks;
  unsigned long long* ixc = a->ixc;
  // (1) the capacity verdict (index_budget_kernel): the first wavefront, lane = shard
  if (threadIdx.x < 64u) {
    const uint32_t l = threadIdx.x;
    unsigned long long lv = __hip_atomic_load(&ixc[l * 8u + kIxLive], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long wd = __hip_atomic_load(&ixc[l * 8u + kIxWords], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&ixc[l * 8u + kIxReserved], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int off = 32; off >= 1; off >>= 1) { lv += __shfl_xor((long long)lv, off); wd += __shfl_xor((long long)wd, off); }
    if (l == 0u) {
      const long long live = (long long)lv, words = (long long)wd;
      long long left = (long long)a->limit - live;
      const long long left_w = (long long)words_cap(slots) - words;
      if (left_w < left) left = left_w;
      s_il.left = left;
      s_il.safe = (live >= 0 && (unsigned long long)live + total < (unsigned long long)a->limit && (unsigned long long)words + total < (unsigned long long)words_cap(slots)) ? 1u : 0u;
    }
  }
  __syncthreads();                                         // (the booking words were zeroed with agent-scope atomic stores, acknowledged at the barrier)
  // (2) a thread per (request, block) pair, in rounds of the workgroup's size (index_insert_picks_kernel)
  SortWl sw;
  sw.wl = a->sort_wl; sw.cap = a->sort_cap; sw.which = 0u;
  const LW* act = (const LW*)a->act;
  for (uint32_t base = 0; base < total; base += blockDim.x) {           // (uniform: every thread calls index_insert_one in every round)
    const uint32_t t = base + threadIdx.x;
    const uint32_t r = t / max_blocks, i = t % max_blocks;
    bool active = t < total, known_only = false;
    int32_t pick = -1;
    uint64_t h = 0;
    if (active) {
      const uint32_t lw = a->learn[r];
      pick = (lw & 0x00FFFF00u) ? (int32_t)((lw >> 8) & 0xFFFFu) - 1 : __hip_atomic_load(&rb.out_pick[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      const uint8_t* row = rb.reqs + (size_t)r * a->stride;
      const uint32_t nb = ((const uint32_t*)row)[1];
      const bool bad = (pick >= 0 && (uint32_t)pick >= a->max_pods) || nb > max_blocks;     // (the host has validated the rows: never, as a rule)
      if (bad && i == 0u) atomicOr(a->status, (uint32_t)pick >= a->max_pods && pick >= 0 ? kStatusBadPick : kStatusBadRow);
      active = !bad && pick >= 0 && i < nb;
      if (active) h = ((const uint64_t*)(row + 8))[i];
      known_only = active && (lw >> 31) != 0u && i < (lw & 0xFFu);
    }
    index_insert_one<LW>(a->keys_w, a->bitmaps_w, a->lists_w, a->rstamps, slots, a->ix.shift, a->limit, a->epoch, ixc, &s_il, s_tmp, h, (uint32_t)pick, active, act, sw,
                         a->status, known_only);
  }
  // (3) the lists that changed go back to ascending order and get their set ids again (index_canon_kernel).  The work-list entries were
  // plain stores of this workgroup's lanes: acknowledged at the barrier, read below past the vector cache (agent-scope loads)
  __syncthreads();
  uint32_t* wl = a->sort_wl;
  const uint32_t n_listed_raw = __hip_atomic_load(&wl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const uint32_t n_listed = n_listed_raw < a->sort_cap ? n_listed_raw : a->sort_cap;
  SetTab st;
  st.lines = a->lists_w + ((size_t)slots + 4u) * kListDwords; st.mask = a->ix.sets_mask; st.ctl = a->set_ctl;
  for (uint32_t base = 0; base < n_listed; base += blockDim.x) {        // (uniform: canon_slot is called by every lane of a wavefront)
    const uint32_t i = base + threadIdx.x;
    const bool active = i < n_listed;
    const uint32_t slot = active ? __hip_atomic_load(&wl[4u + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    canon_slot(a->keys_w, a->lists_w, slots, st, slot, active, false);
  }
  __syncthreads();
  if (threadIdx.x == 0u) __hip_atomic_store(&wl[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // (the cursor, for the next batch)
}

}  // namespace eppk
