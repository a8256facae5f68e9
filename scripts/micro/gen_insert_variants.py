#!/usr/bin/env python3
"""Experiment tool: copies of csrc/eppk_kernels.hip.h with a modified index_insert_one, each in its own namespace, for
scripts/micro/insertbreak.hip (the product sources are not touched).  Output: scripts/micro/_gen/ (not committed).
  v1  a key this thread just claimed (newkey): no loads before the atomics -- the stamp is stored, atomicOr's return value says whether
      the bit is new, the list position comes straight from atomicAdd
  v2  v1 + the bucket is read with four 16-byte coherent loads (sc0 sc1) instead of eight 8-byte atomic loads
  f1  v2 + the live / words / dropped counters bumped once per WORKGROUP (LDS) instead of once per wavefront
  f2  f1 with 64 counter shards instead of 32          f3  the library's insert with only the per-workgroup counters
  f4  f1 with 256 counter shards (ix_budget sums four per lane) + the evict kernel's counters once per workgroup
  e1  f2 + eviction with a LANE per victim (all victims of a wave step in flight) instead of the wavefront walking them one by one
  e2  e1 + the stamp loaded beside the key (not behind the key test); the harness launches it with a wavefront per 64 slots (no grid cap)
  d4  f4 where a claimed key does not write its row (diagnostic for "rows only for overflowed sets": wrong table)
  d1  v2 without the live / words / dropped counters      d2  d1 without the stamp / row / list updates (claim only)
  d3  v2 counting lost claims (a CAS that found the word taken by another key) in the "evicted" counter
  (d1-d3 are diagnostics: they partition the time and give wrong tables)"""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "gateway-api-inference-extension_amd", "csrc", "eppk_kernels.hip.h")
OUT = os.path.join(ROOT, "scripts", "micro", "_gen")

TAIL_OLD = """    if (st < epoch) atomicMax(&stamps[slot], epoch);
    if (!have && bitmap_set<LW>(bitmaps, slot, pod) && lists) list_append(lists, slot, pod);
  }
}"""
TAIL_NEW = """    if (st < epoch) atomicMax(&stamps[slot], epoch);
    if (!have && bitmap_set<LW>(bitmaps, slot, pod) && lists) list_append(lists, slot, pod);
  }
  }
}"""
HEAD_OLD = """  if (active && slot != kNotFound) {
    // the three things an insert may have to update, read together (one round trip): the key's stamp, the pod's bit, the list count
    const uint32_t lane = pod & 63u, j = pod >> 6;"""
HEAD_NEW = """  if (active && slot != kNotFound) {
    const uint32_t lane = pod & 63u, j = pod >> 6;
    if (newkey) {
      // this thread claimed the word a moment ago: whatever the stamp, the row and the list hold is nobody's yet -- no look before the
      // atomics.  (Every insert of a launch carries the same epoch: a racing atomicMax writes the same value.)
      __hip_atomic_store(&stamps[slot], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      bool fresh;
      if constexpr (sizeof(LW) == 8) fresh = !((atomicOr((unsigned long long*)bitmaps + (size_t)slot * 64u + lane, 1ull << j) >> j) & 1ull);
      else fresh = bitmap_set<LW>(bitmaps, slot, pod);
      if (fresh && lists) {
        uint32_t* L = lists + (size_t)slot * kListDwords;
        const uint32_t q = atomicAdd(&L[3], 1u);
        if (q < kListCap) ((uint16_t*)L)[list_pos(q)] = (uint16_t)pod;
      }
    } else {
    // the three things an insert may have to update, read together (one round trip): the key's stamp, the pod's bit, the list count"""
LOAD_OLD = """          unsigned long long w[kBucket];
#pragma unroll
          for (uint32_t i = 0; i < kBucket; ++i) w[i] = __hip_atomic_load(&kb[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // 8 loads in flight"""
LOAD_NEW = """          unsigned long long w[kBucket];
          {
            typedef unsigned long long u64x2_t __attribute__((ext_vector_type(2)));
            u64x2_t q0, q1, q2, q3;
            asm volatile("global_load_dwordx4 %0, %4, off sc0 sc1\\n\\tglobal_load_dwordx4 %1, %4, off offset:16 sc0 sc1\\n\\t"
                         "global_load_dwordx4 %2, %4, off offset:32 sc0 sc1\\n\\tglobal_load_dwordx4 %3, %4, off offset:48 sc0 sc1\\n\\ts_waitcnt vmcnt(0)"
                         : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3) : "v"(kb) : "memory");
            w[0] = q0.x; w[1] = q0.y; w[2] = q1.x; w[3] = q1.y; w[4] = q2.x; w[5] = q2.y; w[6] = q3.x; w[7] = q3.y;
          }"""


# diagnostics (WRONG results on purpose: they only partition the time)
CNT_OLD = """  if ((threadIdx.x & 63u) == 0u && (nk | nw | dropped)) {"""
CNT_NEW = """  if (false && (threadIdx.x & 63u) == 0u && (nk | nw | dropped)) {"""
UPD_OLD = """  if (active && slot != kNotFound) {
    const uint32_t lane = pod & 63u, j = pod >> 6;
    if (newkey) {"""
UPD_NEW = """  if (false && active && slot != kNotFound) {
    const uint32_t lane = pod & 63u, j = pod >> 6;
    if (newkey) {"""
RETRY_OLD = """        // else: somebody else took the word for another key -> search again"""
RETRY_NEW = """        else atomicAdd(&ixc[((blockIdx.x) & (kIxShards - 1u)) * 8u + kIxEvicted], 1ull);   // (diagnostic: lost claims)"""


# the fix under test: one counter update per WORKGROUP (LDS), not per wavefront; optionally more shards
AGG_OLD = """  if ((threadIdx.x & 63u) == 0u && (nk | nw | dropped)) {
    // exact mode counts in shard 0 (what its capacity test reads); otherwise the wavefront's own shard
    const uint32_t shard = bud.safe ? ((blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & (kIxShards - 1u)) : 0u;
    if (nk) atomicAdd(&ixc[shard * 8u + kIxLive], (unsigned long long)__builtin_popcountll(nk));
    if (nw) atomicAdd(&ixc[shard * 8u + kIxWords], (unsigned long long)__builtin_popcountll(nw));
    if (dropped) atomicAdd(&ixc[shard * 8u + kIxDropped], (unsigned long long)__builtin_popcountll(dropped));
  }"""
AGG_NEW = """  __shared__ unsigned int s_cnt[3];
  if (bud.safe) {          // (uniform over the workgroup) the common case: nobody reads the counters while the kernel runs
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
    // exact mode counts in shard 0 (what its capacity test reads)
    if (nk) atomicAdd(&ixc[kIxLive], (unsigned long long)__builtin_popcountll(nk));
    if (nw) atomicAdd(&ixc[kIxWords], (unsigned long long)__builtin_popcountll(nw));
    if (dropped) atomicAdd(&ixc[kIxDropped], (unsigned long long)__builtin_popcountll(dropped));
  }"""
SH_OLD = "constexpr uint32_t kIxShards = 32u;"
SH_NEW = "constexpr uint32_t kIxShards = 64u;"


EVICT_OLD = """  if (lane == 0 && gone) {
    const uint32_t shard = (wave & (kIxShards - 1u)) * 8u;
    atomicAdd(&ixc[shard + kIxLive], (unsigned long long)(0ull - (unsigned long long)gone));
    atomicAdd(&ixc[shard + kIxEvicted], (unsigned long long)gone);   // evicted by this launch (the synchronous entry point zeroes it first)
  }
}"""
EVICT_NEW = """  __shared__ unsigned int s_gone;
  if (threadIdx.x == 0u) s_gone = 0u;
  __syncthreads();
  if (lane == 0 && gone) atomicAdd(&s_gone, gone);
  __syncthreads();
  if (threadIdx.x == 0u && s_gone) {
    const uint32_t shard = (blockIdx.x & (kIxShards - 1u)) * 8u;
    atomicAdd(&ixc[shard + kIxLive], (unsigned long long)(0ull - (unsigned long long)s_gone));
    atomicAdd(&ixc[shard + kIxEvicted], (unsigned long long)s_gone);
  }
}"""
# 256 shards: ix_budget sums four shards per lane
BUD_OLD = """    unsigned long long lv = l < kIxShards ? __hip_atomic_load(&ixc[l * 8u + kIxLive], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
    unsigned long long wd = l < kIxShards ? __hip_atomic_load(&ixc[l * 8u + kIxWords], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
    const unsigned long long lv0 = __shfl((long long)lv, 0), wd0 = __shfl((long long)wd, 0);"""
BUD_NEW = """    unsigned long long lv = l < kIxShards ? __hip_atomic_load(&ixc[l * 8u + kIxLive], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
    unsigned long long wd = l < kIxShards ? __hip_atomic_load(&ixc[l * 8u + kIxWords], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
    const unsigned long long lv0 = __shfl((long long)lv, 0), wd0 = __shfl((long long)wd, 0);
    for (uint32_t sh = l + 64u; sh < kIxShards; sh += 64u) {
      lv += __hip_atomic_load(&ixc[sh * 8u + kIxLive], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      wd += __hip_atomic_load(&ixc[sh * 8u + kIxWords], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }"""
SH256_NEW = "constexpr uint32_t kIxShards = 256u;"
# diagnostic: a claimed key does not write its row (what "rows only for overflowed sets" would save); wrong table
ROW_OLD = """      if constexpr (sizeof(LW) == 8) fresh = !((atomicOr((unsigned long long*)bitmaps + (size_t)slot * 64u + lane, 1ull << j) >> j) & 1ull);
      else fresh = bitmap_set<LW>(bitmaps, slot, pod);"""
ROW_NEW = """      fresh = true; (void)lane; (void)j;"""


# eviction with every victim of a wave step in flight at once: a LANE per victim (its list by four 16-byte loads, a store per listed
# pod, the list reset by four 16-byte stores) instead of the wavefront walking its victims one after the other
EV_OLD = """    unsigned long long vm = __ballot(victim);
    gone += (uint32_t)__builtin_popcountll(vm);
    while (vm) {
      const uint32_t v = base + (uint32_t)__builtin_ctzll(vm);
      vm &= vm - 1ull;
      bool whole = true;
      if (lists) {"""
EV_NEW = """    unsigned long long vm = __ballot(victim);
    gone += (uint32_t)__builtin_popcountll(vm);
    bool whole_l = victim;
    if (victim && lists) {
      u32x4_t* Lp = (u32x4_t*)(lists + (size_t)row * kListDwords);
      u32x4_t c0 = Lp[0], c1 = Lp[1], c2 = Lp[2], c3 = Lp[3];
      if (c0.w <= kListCap) {
        whole_l = false;
        const uint32_t w6[12] = {c0.x, c0.y, c0.z, c1.x, c1.y, c1.z, c2.x, c2.y, c2.z, c3.x, c3.y, c3.z};   // ids: positions 0..5 of every chunk
#pragma unroll
        for (int q = 0; q < 12; ++q) {
          const uint32_t lo = w6[q] & 0xFFFFu, hi = w6[q] >> 16;
          if (lo != kListNone && (lo >> 6) < 8u * (uint32_t)sizeof(LW)) ((LW*)bitmaps)[(size_t)row * 64u + (lo & 63u)] = 0;
          if (hi != kListNone && (hi >> 6) < 8u * (uint32_t)sizeof(LW)) ((LW*)bitmaps)[(size_t)row * 64u + (hi & 63u)] = 0;
        }
      }
      const u32x4_t e0 = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u}, e1 = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
      Lp[0] = e0; Lp[1] = e1; Lp[2] = e1; Lp[3] = e1;
    }
    vm = __ballot(whole_l);          // overflowed lists (or no lists at all): the whole row, by the wavefront
    while (vm) {
      const uint32_t v = base + (uint32_t)__builtin_ctzll(vm);
      vm &= vm - 1ull;
      bool whole = true;
      if (false) {"""


ST_OLD = """      const uint64_t k = keys[row];
      victim = k != 0ull && !(row < slots && k == kTomb) && stamps[row] < min_epoch;"""
ST_NEW = """      const uint64_t k = keys[row];
      const uint32_t st = stamps[row];                 // (not behind the key test: the two loads travel together)
      victim = k != 0ull && !(row < slots && k == kTomb) && st < min_epoch;"""


def main():
    os.makedirs(OUT, exist_ok=True)
    src = open(SRC).read()
    for name, edits in (("v1", [(HEAD_OLD, HEAD_NEW), (TAIL_OLD, TAIL_NEW)]), ("v2", [(HEAD_OLD, HEAD_NEW), (TAIL_OLD, TAIL_NEW), (LOAD_OLD, LOAD_NEW)]),
                        ("d1", [(HEAD_OLD, HEAD_NEW), (TAIL_OLD, TAIL_NEW), (LOAD_OLD, LOAD_NEW), (CNT_OLD, CNT_NEW)]),
                        ("d2", [(HEAD_OLD, HEAD_NEW), (TAIL_OLD, TAIL_NEW), (LOAD_OLD, LOAD_NEW), (CNT_OLD, CNT_NEW), (UPD_OLD, UPD_NEW)]),
                        ("f1", [(HEAD_OLD, HEAD_NEW), (TAIL_OLD, TAIL_NEW), (LOAD_OLD, LOAD_NEW), (AGG_OLD, AGG_NEW)]),
                        ("f2", [(HEAD_OLD, HEAD_NEW), (TAIL_OLD, TAIL_NEW), (LOAD_OLD, LOAD_NEW), (AGG_OLD, AGG_NEW), (SH_OLD, SH_NEW)]),
                        ("f3", [(AGG_OLD, AGG_NEW)]),
                        ("f4", [(HEAD_OLD, HEAD_NEW), (TAIL_OLD, TAIL_NEW), (LOAD_OLD, LOAD_NEW), (AGG_OLD, AGG_NEW), (SH_OLD, SH256_NEW), (BUD_OLD, BUD_NEW), (EVICT_OLD, EVICT_NEW)]),
                        ("e1", [(HEAD_OLD, HEAD_NEW), (TAIL_OLD, TAIL_NEW), (LOAD_OLD, LOAD_NEW), (AGG_OLD, AGG_NEW), (SH_OLD, SH_NEW), (EV_OLD, EV_NEW)]),
                        ("e2", [(HEAD_OLD, HEAD_NEW), (TAIL_OLD, TAIL_NEW), (LOAD_OLD, LOAD_NEW), (AGG_OLD, AGG_NEW), (SH_OLD, SH_NEW), (EV_OLD, EV_NEW), (ST_OLD, ST_NEW)]),
                        ("d4", [(HEAD_OLD, HEAD_NEW), (TAIL_OLD, TAIL_NEW), (LOAD_OLD, LOAD_NEW), (AGG_OLD, AGG_NEW), (SH_OLD, SH256_NEW), (BUD_OLD, BUD_NEW), (EVICT_OLD, EVICT_NEW), (ROW_OLD, ROW_NEW)]),
                        ("d3", [(HEAD_OLD, HEAD_NEW), (TAIL_OLD, TAIL_NEW), (LOAD_OLD, LOAD_NEW), (RETRY_OLD, RETRY_NEW)])):
        s = src
        for a, b in edits:
            assert s.count(a) == 1, (name, a[:60])
            s = s.replace(a, b)
        s = s.replace("#pragma once", "").replace("namespace eppk {", "namespace eppk_%s {" % name, 1).replace("}  // namespace eppk", "}  // namespace eppk_%s" % name)
        s = s.replace('#include "../../include/eppk.h"', '#include "../../../include/eppk.h"')
        open(os.path.join(OUT, "eppk_kernels_%s.hip.h" % name), "w").write(s)
        print("wrote", name)


if __name__ == "__main__":
    main()
