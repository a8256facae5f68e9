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
//   * argmax across lanes: 6-step xor-shuffle on (score desc, index asc).
//   * no MFMA: this is integer/bit/f64-add work (north_star); the bound is HBM for index rows.
//
// Replaces (reference, all spec-only or Go): Scorer.Score + weighted sum + Picker.Pick
//   docs/proposals/0845-scheduler-architecture-proposal/interfaces/interface.go:113-142,
//   called per request through EndpointPicker.Pick, pkg/lwepp/handlers/server.go:79-82.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

// Build-time tuning knobs (A/B-tested on MI355X; defaults are the measured best — DESIGN.md §7)
#ifndef EPPK_PIPE
#define EPPK_PIPE 0      // 2-stage software pipeline (probe request r+1 while the rows of r are in flight): measured neutral, costs 20 VGPRs
#endif
#ifndef EPPK_ROWS2
#define EPPK_ROWS2 0     // keep two 8-row batches (16 loads) in flight instead of one
#endif
#ifndef EPPK_PTERM_TAB
#define EPPK_PTERM_TAB 1 // prefix term of a matched pod from the exact host-built table instead of an f64 division
#endif
#ifndef EPPK_MIN_WAVES
#define EPPK_MIN_WAVES 1 // __launch_bounds__ minimum waves per SIMD for the fast kernel
#endif

namespace eppk {

constexpr uint32_t kNotFound = 0xFFFFFFFFu;
constexpr uint32_t kNoPod = 0xFFFFFFFFu;
constexpr uint64_t kHomeMul = 0x9E3779B97F4A7C15ull;
constexpr uint64_t kTomb = ~0ull;           // key of a slot whose pod set became empty (never matches, never reused)
constexpr uint32_t kStatSlots = 32768u;  // per-wavefront probe-statistics slots: stats[4 + 2*wave + {0,1}]

// ---- kernel argument blocks (plain structs, passed by value) --------------------------------

struct KSnap {
  const double*   base;    // [J*64] fused leading pod-only terms (fast path)
  const uint32_t* queue;   // [J*64]
  const double*   kv;      // [J*64]
  const void*     act_t;   // [A][64] LW  bit j of [a][l] : adapter a active on pod j*64+l
  const void*     wait_t;  // [A][64] LW
  const void*     free_t;  // [64] LW     loaded < max_lora
  const double*   topv;    // [129][64] per adapter row (128 = base model): the 64 best pods by
  const uint32_t* topi;    //           T_a[p] = base[p] (+ lw[tier(a,p)]), sorted (T desc, p asc); kNoPod-padded
  const void*     qmin_t;  // [64] LW  pods whose queue == qmin / qmax (masked fast path: are the request's
  const void*     qmax_t;  //          QUEUE normalisers the global ones?)
  uint32_t        lead_queue;  // the fused leading run contains a QUEUE scorer
  const double*   pterm;   // [(B+1)][pterm_ld] exact clamp01(c/n) * w_prefix for 1 <= n <= B, c <= n (null when B > 64)
  uint32_t pterm_ld;
  uint32_t n_pods;
  uint32_t J;              // ceil(n_pods/64)
  uint32_t qmin, qmax;     // over all pods (unmasked QUEUE scorer)
};

struct KIndex {
  const uint64_t* keys;    // [slots+2]; 0 = empty, ~0 = tombstone; keys[slots], keys[slots+1] = presence of hashes 0 / ~0.
                           // Invariant: a key that is present has a NON-EMPTY row.
  const void*     bitmaps; // [slots+3][64] LW: rows slots / slots+1 hold hashes 0 / ~0, row slots+2 is all-zero
  uint32_t slots;          // power of two (0 = no index)
  uint32_t shift;          // 64 - log2(slots)
};

struct KChain {            // the whole weighted chain (generic kernel)
  uint32_t n;
  uint32_t kind[8];
  double   w[8];
};

struct KTail {             // the request-dependent tail after fusion (fast kernel)
  double lw[4];            // clamp01(tier score) * w_lora, tier 0..3 = {0.0, 0.6, 0.8, 1.0}
  double wp;               // (double) w_prefix
};

// ---- small device helpers --------------------------------------------------------------------

__device__ __forceinline__ uint32_t home_slot(uint64_t h, uint32_t shift) {
  return (uint32_t)((h * kHomeMul) >> shift);
}

__device__ __forceinline__ double clamp01(double s) {
  if (!(s >= 0.0)) return 0.0;
  if (s > 1.0) return 1.0;
  return s;
}

// Look one hash up; returns its slot or kNotFound.  Linear probing; the table is never full.
__device__ __forceinline__ uint32_t probe(const KIndex& ix, uint64_t h, bool active) {
  if (!active || ix.slots == 0) return kNotFound;
  if (h == 0) return ix.keys[ix.slots] ? ix.slots : kNotFound;             // reserved hashes: presence words
  if (h == kTomb) return ix.keys[ix.slots + 1u] ? ix.slots + 1u : kNotFound;
  uint32_t s = home_slot(h, ix.shift);
  const uint32_t mask = ix.slots - 1;
  for (uint32_t n = 0; n < ix.slots; ++n) {
    const uint64_t k = ix.keys[s];
    if (k == h) return s;
    if (k == 0) return kNotFound;
    s = (s + 1) & mask;
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

// The prefix walk of one request (SEMANTICS.md §3 PREFIX) into bit-sliced counters.
// Returns the number of non-empty index rows added (= leading non-empty look-ups).
template <typename LW, int NPL>
__device__ __forceinline__ uint32_t prefix_walk(const KIndex& ix, const uint64_t* hs, uint32_t nb, int lane,
                                                LW (&c)[NPL]) {
  const LW* bm = (const LW*)ix.bitmaps;
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
    for (uint32_t k0 = 0; k0 < m && !stop; k0 += 8) {      // 8 independent row loads in flight
      LW w[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const uint32_t k = k0 + (uint32_t)u;
        const uint32_t s = __builtin_amdgcn_readlane(slot, (k < m) ? k : 0);
        w[u] = (k < m) ? bm[(size_t)s * 64u + (uint32_t)lane] : (LW)0;
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
// Full adder on bit vectors: x = a^b, sum = x^c, carry = maj(a,b,c) = bfi(x, c, a) — 3 VALU ops per 32 bits.
// hipcc does not form v_bfi_b32 from the C expression here (it emits and/and/or), hence the asm.
__device__ __forceinline__ uint32_t bfi32(uint32_t m, uint32_t x, uint32_t y) {  // (m & x) | (~m & y)
  uint32_t r;
  asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(m), "v"(x), "v"(y));
  return r;
}
template <typename LW>
__device__ __forceinline__ LW bfi(LW m, LW x, LW y) {
  if constexpr (sizeof(LW) == 8)
    return ((uint64_t)bfi32((uint32_t)(m >> 32), (uint32_t)(x >> 32), (uint32_t)(y >> 32)) << 32) | bfi32((uint32_t)m, (uint32_t)x, (uint32_t)y);
  else
    return (LW)bfi32((uint32_t)m, (uint32_t)x, (uint32_t)y);
}
template <typename LW>
__device__ __forceinline__ void full_add(LW a, LW b, LW c, LW& sum, LW& carry) {
  const LW x = a ^ b;
  sum = x ^ c;
  carry = bfi<LW>(x, c, a);
}
template <typename LW>
__device__ __forceinline__ void half_add(LW a, LW b, LW& sum, LW& carry) {
  sum = a ^ b;
  carry = a & b;
}

// Add eight 0/1 vectors into the bit-sliced counters: an 8->4 carry-save tree (4 full + 3 half adders)
// then one 4-bit ripple add into the NPL planes; ~4.5x fewer VALU ops than eight ripple adds.
template <typename LW, int NPL>
__device__ __forceinline__ void planes_add8(LW (&c)[NPL], const LW (&w)[8]) {
  static_assert(NPL >= 5, "need at least 5 planes");
  LW s1, c1, s2, c2, s3, c3, b0, c4, s5, c5, b1, c6, b2, b3;
  full_add<LW>(w[0], w[1], w[2], s1, c1);
  full_add<LW>(w[3], w[4], w[5], s2, c2);
  full_add<LW>(w[6], w[7], s1, s3, c3);
  half_add<LW>(s2, s3, b0, c4);          // ones
  full_add<LW>(c1, c2, c3, s5, c5);      // twos
  half_add<LW>(s5, c4, b1, c6);
  half_add<LW>(c5, c6, b2, b3);          // fours, eights
  LW carry, t;
  half_add<LW>(c[0], b0, t, carry); c[0] = t;
  full_add<LW>(c[1], b1, carry, t, carry); c[1] = t;
  full_add<LW>(c[2], b2, carry, t, carry); c[2] = t;
  full_add<LW>(c[3], b3, carry, t, carry); c[3] = t;
#pragma unroll
  for (int k = 4; k < NPL; ++k) { half_add<LW>(c[k], carry, t, carry); c[k] = t; }
}

template <int NPL>
__device__ __forceinline__ uint32_t planes_get(const uint32_t (&c32)[NPL], uint32_t jj) {
  uint32_t cnt = 0;
#pragma unroll
  for (int k = 0; k < NPL; ++k) cnt |= ((c32[k] >> jj) & 1u) << k;
  return cnt;
}

// Transpose one natural-layout candidate mask row ([J] u64, bit p%64 of word p/64) into a lane word.
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
// No LDS and few registers: occupancy and memory-level parallelism are what this kernel lives on.
// The request loop is a 2-stage software pipeline: while the rows of request r are in flight the keys
// of request r+1 are probed, and the row of request r+2 is prefetched.

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

// Stage 1 of a request: probe the keys of one 64-block chunk in parallel.  Returns the number m of leading
// hits; slot_eff[lane k] = row of hit k for k < m, the all-zero row otherwise.
__device__ __forceinline__ uint32_t probe_chunk(const KIndex& ix, uint64_t h, uint32_t nchunk, int lane, uint32_t& slot_eff) {
  const uint32_t slot = probe(ix, h, (uint32_t)lane < nchunk);
  const unsigned long long found = __ballot(slot != kNotFound);
  const uint32_t m = (~found == 0ull) ? 64u : (uint32_t)__builtin_ctzll(~found);
  slot_eff = ((uint32_t)lane < m) ? slot : ix.slots + 2u;
  return m;
}

template <typename LW>
__device__ __forceinline__ void load_rows8(const LW* bm, uint32_t slot_eff, uint32_t k0, int lane, LW (&w)[8]) {
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const uint32_t s = __builtin_amdgcn_readlane(slot_eff, k0 + (uint32_t)u);
    w[u] = bm[(size_t)s * 64u + (uint32_t)lane];
  }
}

// Exact evaluation of one request over its CANDIDATES ONLY, every scorer in chain order (the generic kernel's
// arithmetic), each lane walking the set bits of its candidate word.  Used by the masked fast kernel when the
// request's QUEUE normalisers differ from the snapshot-wide ones (so base[] / the top tables do not apply);
// cost is proportional to the candidates per lane — small subsets (the common reason for that case) are cheap.
template <typename LW, int NPL>
__device__ __forceinline__ void masked_exact(const KSnap& sn, const KChain& ch, LW cand, const LW (&c)[NPL], LW thi, LW tlo,
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
  LW rem = cand;
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

template <typename LW, int NPL, bool HAS_L, bool HAS_P, bool P_FIRST, bool MASKED>
__global__ __launch_bounds__(256, EPPK_MIN_WAVES) void pick_fast_kernel(KSnap sn, KIndex ix, KTail tl, const uint8_t* __restrict__ reqs,
                                                        uint32_t stride, uint32_t n_reqs, uint32_t pwn,
                                                        const uint64_t* __restrict__ cand_mask, KChain ch,
                                                        int32_t* __restrict__ out_pick, double* __restrict__ out_score,
                                                        unsigned long long* __restrict__ stats) {
  (void)pwn;
  const int lane = (int)(threadIdx.x & 63u);
  const uint32_t wpb = blockDim.x >> 6;
  const uint32_t gwave = blockIdx.x * wpb + (threadIdx.x >> 6);
  const uint32_t nwaves = gridDim.x * wpb;
  const LW* bm = (const LW*)ix.bitmaps;

  const LW freew = HAS_L ? ((const LW*)sn.free_t)[lane] : (LW)0;
  const LW valid = valid_word<LW>(sn.n_pods, lane);
  const LW qminw = (MASKED && sn.lead_queue) ? ((const LW*)sn.qmin_t)[lane] : (LW)0;
  const LW qmaxw = (MASKED && sn.lead_queue) ? ((const LW*)sn.qmax_t)[lane] : (LW)0;
  unsigned long long w_hits = 0, w_lookups = 0;
  const uint32_t hwords = (stride - 8u) / 8u < 64u ? (stride - 8u) / 8u : 64u;

  // ---- pipeline prologue: request r is probed, request r+nwaves is prefetched
  uint32_t r = gwave;
  if (r >= n_reqs) return;
  int32_t adapter;
  uint32_t nb, m0 = 0, slot0 = 0;
  uint64_t cur_h = 0;  // non-pipelined build: hashes of the current request
  {
    const uint64_t* row64 = (const uint64_t*)(reqs + (size_t)r * stride);
    const uint64_t hdr = row64[0];
    const uint64_t h = (HAS_P && (uint32_t)lane < hwords) ? row64[1 + lane] : 0ull;
    adapter = __builtin_amdgcn_readfirstlane((int32_t)(uint32_t)hdr);
    nb = (uint32_t)__builtin_amdgcn_readfirstlane((int32_t)(uint32_t)(hdr >> 32));
    if (HAS_P && EPPK_PIPE) m0 = probe_chunk(ix, h, nb < 64u ? nb : 64u, lane, slot0);
    cur_h = h;
  }
  uint64_t nx_hdr = 0, nx_h = 0;
  if (r + nwaves < n_reqs) {
    const uint64_t* row64 = (const uint64_t*)(reqs + (size_t)(r + nwaves) * stride);
    nx_hdr = row64[0];
    nx_h = (HAS_P && (uint32_t)lane < hwords) ? row64[1 + lane] : 0ull;
  }

  for (; r < n_reqs; r += nwaves) {
    // ---- A. issue everything the current request needs that is already addressable
    LW w[8];
#if EPPK_ROWS2
    LW w2[8];
#endif
    if (HAS_P && !EPPK_PIPE) m0 = probe_chunk(ix, cur_h, nb < 64u ? nb : 64u, lane, slot0);
    if (HAS_P && m0 > 0) load_rows8<LW>(bm, slot0, 0, lane, w);
#if EPPK_ROWS2
    if (HAS_P && m0 > 8) load_rows8<LW>(bm, slot0, 8, lane, w2);
#endif
    const uint32_t arow = (HAS_L && adapter >= 0) ? (uint32_t)adapter : 128u;
    const double top_t = sn.topv[(size_t)arow * 64u + (uint32_t)lane];
    const uint32_t top_p = sn.topi[(size_t)arow * 64u + (uint32_t)lane];
    LW thi = 0, tlo = 0;  // LoRA tier planes: tier = 2*hi + lo -> {0: 0.0, 1: 0.6 waiting, 2: 0.8 free slot, 3: 1.0 active}
    if (HAS_L) {
      LW a = 0, wt = 0;
      if (adapter >= 0) {
        a = ((const LW*)sn.act_t)[(size_t)adapter * 64u + (uint32_t)lane];
        wt = ((const LW*)sn.wait_t)[(size_t)adapter * 64u + (uint32_t)lane];
      }
      thi = a | freew;
      tlo = a | ((LW)~freew & wt);
    }
    LW cand = valid;   // Filter: the request's candidate subset (request.go:104-133 as a bitmask), lane-transposed
    if (MASKED) cand &= transpose_mask<LW>(cand_mask + (size_t)r * sn.J, sn.J, lane);

    // ---- B. stage 1 of the NEXT request (its row was prefetched one iteration ago), overlapping the loads above
    const uint32_t rn = r + nwaves;
    int32_t adapter_n = -1;
    uint32_t nb_n = 0, m_n = 0, slot_n = 0;
    if (rn < n_reqs) {
      adapter_n = __builtin_amdgcn_readfirstlane((int32_t)(uint32_t)nx_hdr);
      nb_n = (uint32_t)__builtin_amdgcn_readfirstlane((int32_t)(uint32_t)(nx_hdr >> 32));
      if (HAS_P && EPPK_PIPE) m_n = probe_chunk(ix, nx_h, nb_n < 64u ? nb_n : 64u, lane, slot_n);
      cur_h = nx_h;
      if (rn + nwaves < n_reqs) {  // prefetch the row after next
        const uint64_t* row64 = (const uint64_t*)(reqs + (size_t)(rn + nwaves) * stride);
        nx_hdr = row64[0];
        nx_h = (HAS_P && (uint32_t)lane < hwords) ? row64[1 + lane] : 0ull;
      }
    }

    // ---- C. stage 2 of the current request: count rows, evaluate, pick
    LW c[NPL];
#pragma unroll
    for (int k = 0; k < NPL; ++k) c[k] = 0;
    LW nz = 0;  // M: pods with matched > 0
    if (HAS_P) {
      uint32_t hits = m0;
      if (m0 > 0) {
        planes_add8<LW, NPL>(c, w);
        nz |= (LW)(w[0] | w[1] | w[2] | w[3] | w[4] | w[5] | w[6] | w[7]);
#if EPPK_ROWS2
        if (m0 > 8) {
          planes_add8<LW, NPL>(c, w2);
          nz |= (LW)(w2[0] | w2[1] | w2[2] | w2[3] | w2[4] | w2[5] | w2[6] | w2[7]);
        }
        for (uint32_t k0 = 16; k0 < m0; k0 += 8) {
#else
        for (uint32_t k0 = 8; k0 < m0; k0 += 8) {
#endif
          load_rows8<LW>(bm, slot0, k0, lane, w);
          planes_add8<LW, NPL>(c, w);
          nz |= (LW)(w[0] | w[1] | w[2] | w[3] | w[4] | w[5] | w[6] | w[7]);
        }
      }
      // chunks beyond the first 64 blocks (only when every earlier key hit); not pipelined
      uint32_t mlast = m0;
      for (uint32_t b0 = 64; b0 < nb && mlast == 64u; b0 += 64) {
        const uint64_t* hs = (const uint64_t*)(reqs + (size_t)r * stride + 8);
        const uint32_t nchunk = (nb - b0) < 64u ? (nb - b0) : 64u;
        const uint64_t h = ((uint32_t)lane < nchunk) ? hs[b0 + (uint32_t)lane] : 0ull;
        uint32_t slotc;
        mlast = probe_chunk(ix, h, nchunk, lane, slotc);
        for (uint32_t k0 = 0; k0 < mlast; k0 += 8) {
          load_rows8<LW>(bm, slotc, k0, lane, w);
          planes_add8<LW, NPL>(c, w);
          nz |= (LW)(w[0] | w[1] | w[2] | w[3] | w[4] | w[5] | w[6] | w[7]);
        }
        hits += mlast;
      }
      if (stats) { w_hits += hits; w_lookups += (hits + 1u < nb) ? hits + 1u : nb; }
      nz &= valid;
    }
    // Masked requests: base[] and the top tables embed the snapshot-wide QUEUE normalisers; they apply iff the
    // candidates contain a pod at the global minimum and one at the global maximum queue depth.
    bool exact = false;
    if (MASKED && sn.lead_queue) exact = !(__any((cand & qminw) != 0) && __any((cand & qmaxw) != 0));

    double best = -__builtin_inf();
    uint32_t bidx = kNoPod;
    double cand_t = -__builtin_inf();
    uint32_t cand_p = kNoPod;
    if (MASKED && exact) {
      if (__any(cand != 0)) {
        masked_exact<LW, NPL>(sn, ch, cand, c, thi, tlo, nb, lane, best, bidx);
        wave_argmax(best, bidx);
      }
    } else {
      const LW mset = MASKED ? (LW)(nz & cand) : nz;         // candidates with a prefix match: evaluated in full
      const LW okset = (LW)(cand & (LW)~nz);                 // candidates whose total is exactly T_a[p]
      const bool any_m = HAS_P && __any(mset != 0);

      // best candidate outside M: first table entry in okset
      const bool has = top_p != kNoPod;
      bool ok = has;
      if (MASKED || (HAS_P && __any(nz != 0))) {
        const uint32_t ql = has ? (top_p & 63u) : 0u, qj = has ? (top_p >> 6) : 0u;
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
      const unsigned long long okm = __ballot(ok);
      bool scan_rest = false;
      if (okm) {
        const int f = __builtin_ctzll(okm);
        cand_t = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(top_t), f),
                                  __builtin_amdgcn_readlane(__double2loint(top_t), f));
        cand_p = (uint32_t)__builtin_amdgcn_readlane((int)top_p, f);
      } else {
        scan_rest = sn.n_pods > 64u && __any(okset != 0);  // table exhausted although eligible pods remain
      }

      if (any_m) {
        LW rem = mset;
        while (__any(rem != 0)) {          // each lane walks its own pods of M in ascending order
          if (rem != 0) {
            const uint32_t j = (sizeof(LW) == 8) ? (uint32_t)__builtin_ctzll((unsigned long long)rem) : (uint32_t)__builtin_ctz((uint32_t)rem);
            rem = (LW)(rem & (LW)(rem - 1));
            const uint32_t p = j * 64u + (uint32_t)lane;
            uint32_t cnt = 0;
#pragma unroll
            for (int k = 0; k < NPL; ++k) cnt |= (uint32_t)((c[k] >> j) & 1) << k;
            const uint32_t tier = (uint32_t)(((thi >> j) & 1) << 1) | (uint32_t)((tlo >> j) & 1);
            // cnt > 0 implies nb > 0.  pterm = clamp01(cnt / nb) * w_prefix: from the host-built exact table when
            // there is one (max_blocks <= 64), else one binary64 division here.
            const double pterm = (EPPK_PTERM_TAB && sn.pterm) ? sn.pterm[(size_t)nb * sn.pterm_ld + cnt] : clamp01((double)cnt / (double)nb) * tl.wp;
            const double t = eval_total<HAS_L, HAS_P, P_FIRST>(sn.base[p], HAS_L ? tier_term(tl, tier) : 0.0, pterm);
            if (t > best) { best = t; bidx = p; }
          }
        }
      }
      if (scan_rest) {                      // rare: T_a over every eligible pod outside M (total == T_a there)
        double rbest = -__builtin_inf();
        uint32_t ridx = kNoPod;
        for (uint32_t j = 0; j < sn.J; ++j) {
          const uint32_t p = j * 64u + (uint32_t)lane;
          const uint32_t tier = (uint32_t)(((thi >> j) & 1) << 1) | (uint32_t)((tlo >> j) & 1);
          double t = sn.base[p];
          if (HAS_L) t = t + tier_term(tl, tier);
          const bool okp = (okset >> j) & 1;
          if (okp && t > rbest) { rbest = t; ridx = p; }
        }
        if (rbest > best || (rbest == best && ridx < bidx)) { best = rbest; bidx = ridx; }
      }
      if (any_m || scan_rest) wave_argmax(best, bidx);
    }
    if (cand_t > best || (cand_t == best && cand_p < bidx)) { best = cand_t; bidx = cand_p; }
    if (lane == 0) {
      const bool none = bidx == kNoPod;
      out_pick[r] = none ? -1 : (int32_t)bidx;
      if (out_score) out_score[r] = none ? 0.0 : best;
    }

    // ---- rotate the pipeline
    adapter = adapter_n; nb = nb_n; m0 = m_n; slot0 = slot_n;
  }
  // probe statistics: one private slot per wavefront (plain read-modify-write; same-address atomics
  // from ~10^4 waves serialise at ~12 ns each and would add >100 us of tail to the launch)
  if (HAS_P && stats && lane == 0 && (w_hits | w_lookups) && gwave < kStatSlots) {
    stats[4 + 2 * gwave] += w_hits;
    stats[5 + 2 * gwave] += w_lookups;
  }
}

// ---- GENERIC pick kernel -----------------------------------------------------------------------
// Any chain order (duplicates allowed), optional candidate mask; every scorer evaluated per pair in
// chain order.  Slower; it is both the fallback for non-canonical chains / masked batches and an
// independent on-device statement of SEMANTICS.md.
// LDS: queue[J*64] u32 | kv[J*64] f64 | per wave pw[pwn] f64 (raw ratios cnt/n).
template <typename LW, int NPL, bool MASKED>
__global__ __launch_bounds__(512) void pick_generic_kernel(KSnap sn, KIndex ix, KChain ch, const uint8_t* __restrict__ reqs,
                                                           uint32_t stride, uint32_t n_reqs, uint32_t pwn,
                                                           const uint64_t* __restrict__ cand_mask,
                                                           int32_t* __restrict__ out_pick, double* __restrict__ out_score,
                                                           unsigned long long* __restrict__ stats) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double* s_kv = (double*)smem;
  uint32_t* s_q = (uint32_t*)(s_kv + (size_t)sn.J * 64u);
  const int lane = (int)(threadIdx.x & 63u);
  const uint32_t wib = threadIdx.x >> 6;
  const uint32_t wpb = blockDim.x >> 6;
  double* s_pw = (double*)(s_q + (size_t)sn.J * 64u) + (size_t)wib * pwn;

  for (uint32_t i = threadIdx.x; i < sn.J * 64u; i += blockDim.x) { s_kv[i] = sn.kv[i]; s_q[i] = sn.queue[i]; }
  __syncthreads();

  bool has_q = false, has_l = false, has_p = false;
  for (uint32_t k = 0; k < ch.n; ++k) {
    has_q |= ch.kind[k] == 1u; has_l |= ch.kind[k] == 3u; has_p |= ch.kind[k] == 4u;
  }
  const LW freew = has_l ? ((const LW*)sn.free_t)[lane] : (LW)0;
  const LW valid = valid_word<LW>(sn.n_pods, lane);
  unsigned long long w_hits = 0, w_lookups = 0;

  const uint32_t gwave = blockIdx.x * wpb + wib;
  const uint32_t nwaves = gridDim.x * wpb;
  for (uint32_t r = gwave; r < n_reqs; r += nwaves) {
    const uint8_t* row = reqs + (size_t)r * stride;
    const int32_t adapter = __builtin_amdgcn_readfirstlane(((const int32_t*)row)[0]);
    const uint32_t nb = (uint32_t)__builtin_amdgcn_readfirstlane(((const int32_t*)row)[1]);

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
        const uint32_t hits = prefix_walk<LW, NPL>(ix, (const uint64_t*)(row + 8), nb, lane, cc);
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
      LW a = 0, w = 0;
      if (adapter >= 0) {
        a = ((const LW*)sn.act_t)[(size_t)adapter * 64u + (uint32_t)lane];
        w = ((const LW*)sn.wait_t)[(size_t)adapter * 64u + (uint32_t)lane];
      }
      thi = a | freew;
      tlo = a | ((LW)~freew & w);
    }

    // QUEUE normalisers over the request's candidates
    uint32_t qmin = sn.qmin, qmax = sn.qmax;
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

    double best = -__builtin_inf();
    uint32_t bidx = kNoPod;
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
      if (ok && t > best) { best = t; bidx = p; }
    }
    wave_argmax(best, bidx);
    if (lane == 0) {
      const bool none = bidx == kNoPod;
      out_pick[r] = none ? -1 : (int32_t)bidx;
      if (out_score) out_score[r] = none ? 0.0 : best;
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

// ---- prefix index maintenance (0602-…/README.md:101-108) -----------------------------------------
// stats[2] = occupied keys, stats[3] = dropped inserts (table at its load limit)

template <typename LW>
__device__ __forceinline__ void bitmap_set(void* bitmaps, uint32_t slot, uint32_t pod) {
  const uint32_t lane = pod & 63u, j = pod >> 6;
  if constexpr (sizeof(LW) == 8) {
    atomicOr((unsigned long long*)bitmaps + (size_t)slot * 64u + lane, 1ull << j);
  } else if constexpr (sizeof(LW) == 4) {
    atomicOr((unsigned int*)bitmaps + (size_t)slot * 64u + lane, 1u << j);
  } else {
    const size_t e = (size_t)slot * 64u + lane;           // u16 element index
    atomicOr((unsigned int*)bitmaps + (e >> 1), (1u << j) << (16u * (uint32_t)(e & 1u)));
  }
}

template <typename LW>
__device__ __forceinline__ void index_insert_one(uint64_t* keys, void* bitmaps, uint32_t slots, uint32_t shift,
                                                 uint32_t limit, unsigned long long* stats, uint64_t h, uint32_t pod) {
  uint32_t slot = kNotFound;
  if (h == 0 || h == kTomb) {
    slot = h == 0 ? slots : slots + 1u;
    __hip_atomic_store((unsigned long long*)&keys[slot], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    uint32_t s = home_slot(h, shift);
    const uint32_t mask = slots - 1;
    for (uint32_t n = 0; n < slots; ++n) {
      unsigned long long k = __hip_atomic_load((unsigned long long*)&keys[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (k == 0ull) {
        if (__hip_atomic_load(&stats[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned long long)limit) break;
        k = atomicCAS((unsigned long long*)&keys[s], 0ull, (unsigned long long)h);
        if (k == 0ull) { atomicAdd(&stats[2], 1ull); slot = s; break; }
      }
      if (k == (unsigned long long)h) { slot = s; break; }
      s = (s + 1) & mask;
    }
  }
  if (slot == kNotFound) { atomicAdd(&stats[3], 1ull); return; }
  bitmap_set<LW>(bitmaps, slot, pod);
}

template <typename LW>
__global__ void index_insert_kernel(uint64_t* keys, void* bitmaps, uint32_t slots, uint32_t shift, uint32_t limit,
                                    unsigned long long* stats, const uint64_t* hashes, const uint32_t* pods, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) index_insert_one<LW>(keys, bitmaps, slots, shift, limit, stats, hashes[i], pods[i]);
}

// thread (r, i): append picks[r] to hash i of request r
template <typename LW>
__global__ void index_insert_picks_kernel(uint64_t* keys, void* bitmaps, uint32_t slots, uint32_t shift, uint32_t limit,
                                          unsigned long long* stats, const uint8_t* reqs, uint32_t stride,
                                          uint32_t max_blocks, const int32_t* picks, uint32_t n_reqs) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t r = (uint32_t)(t / max_blocks), i = (uint32_t)(t % max_blocks);
  if (r >= n_reqs) return;
  const int32_t pick = picks[r];
  const uint8_t* row = reqs + (size_t)r * stride;
  const uint32_t nb = ((const uint32_t*)row)[1];
  if (pick < 0 || i >= nb) return;
  index_insert_one<LW>(keys, bitmaps, slots, shift, limit, stats, ((const uint64_t*)(row + 8))[i], (uint32_t)pick);
}

// Clear pod's bit in every row; a row that becomes empty gets its key tombstoned so that the hot path never
// meets a present key with an empty pod set.  One wavefront per row (rows = slots + 2: the reserved rows too).
template <typename LW>
__global__ void index_remove_pod_kernel(uint64_t* keys, void* bitmaps, uint32_t slots, uint32_t pod) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (row >= slots + 2u) return;
  LW* w = (LW*)bitmaps + (size_t)row * 64u + lane;
  LW v = *w;
  if (lane == (pod & 63u)) {
    const LW nv = (LW)(v & (LW)~((LW)1 << (pod >> 6)));
    if (nv != v) *w = nv;
    v = nv;
  }
  const bool empty = __ballot(v != 0) == 0ull;
  if (empty && lane == 0 && keys[row] != 0ull) keys[row] = row < slots ? kTomb : 0ull;  // reserved rows: clear presence
}

}  // namespace eppk
