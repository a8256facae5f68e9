// Micro-benchmark (measurement tool, not product code): what a NEW key and an eviction VICTIM cost under different index protocols.
//   new key   A0  CAS on its bucket line + 16-byte store to its list line + 4-byte store to its stamp line        (round 3 as landed)
//             A1  CAS + the whole 64-byte list line (four 16-byte stores) + s_waitcnt + stamp store                (claimer writes the line: victims keep theirs)
//             B0  CAS + 16-byte list store + 1-byte store into the bucket's header word                           (stamps in the header)
//             B1  CAS + 64-byte list line + s_waitcnt + 1-byte store into the bucket's header word
//   ageing    E0  scan keys + stamps; a victim: 64-byte list store + 8-byte key store                             (as landed)
//             E1  scan keys + stamps; the wavefront rewrites its 64 key words and 64 stamps when any lane holds a victim (no list access)
//             E2  scan keys only (stamps in the header word); rewrite of the key words
// Tables as in the closed loop of bench.py: 16 Mi slots (keys 128 MB, stamps 64 MB, lists 1 GiB), 2 Mi new keys / victims per launch.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint64_t mix(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
}

template <int MODE>
__global__ __launch_bounds__(256) void newkey(unsigned long long* keys, uint32_t* lists, uint32_t* stamps, uint32_t slots, uint32_t n, uint64_t seed) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  uint32_t slot = (uint32_t)(mix(seed + k) % slots) | 1u;             // (never a header word)
  if (MODE >= 4) slot = (slot & ~7u) | 1u;                             // buckets fill front to back: a new key sits in position 1, as a rule
  const unsigned long long h = mix(seed * 3u + k) | 2ull;
  atomicCAS(&keys[slot], keys[slot] & 0ull, h);                            // (the bucket was read first, like the search; always succeeds on 0, else a lost CAS: same cost)
  uint32_t* L = lists + (size_t)(MODE == 5 ? slot ^ ((slot >> 3) & 7u) : slot) * 16u;
  const u32x4 first = {0xFFFF0000u | (k & 4095u), 0xFFFFFFFFu, 0xFFFFFFFFu, 1u}, rest = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(L), "v"(first) : "memory");
  if (MODE == 1 || MODE == 3) {
    asm volatile("global_store_dwordx4 %0, %1, off offset:16 sc1\n\tglobal_store_dwordx4 %0, %1, off offset:32 sc1\n\tglobal_store_dwordx4 %0, %1, off offset:48 sc1\n\ts_nop 1\n\ts_waitcnt vmcnt(0)" ::"v"(L), "v"(rest) : "memory");
  }
  if (MODE == 0 || MODE == 1 || MODE >= 4) __hip_atomic_store(&stamps[slot], (uint32_t)seed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else __hip_atomic_store((uint8_t*)&keys[slot & ~7u] + (slot & 7u), (uint8_t)seed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int MODE>
__global__ __launch_bounds__(256) void evict(unsigned long long* keys, uint32_t* lists, uint32_t* stamps, uint32_t slots, uint64_t seed, unsigned long long* out) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  uint32_t gone = 0;
  for (uint32_t base = wave * 64u; base < slots; base += nwaves * 64u) {
    const uint32_t row = base + lane;
    unsigned long long k = keys[row];
    uint32_t st = MODE == 2 ? 0u : stamps[row];
    const bool header = (row & 7u) == 0u;
    const bool victim = !header && (mix(seed + row) & 7ull) == 0ull && (k | st | 1ull) != 0ull;
    const unsigned long long vm = __ballot(victim);
    gone += (uint32_t)__builtin_popcountll(vm);
    if (MODE == 0) {
      if (victim) {
        u32x4* Lp = (u32x4*)(lists + (size_t)row * 16u);
        const u32x4 e0 = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u}, e1 = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
        Lp[0] = e0; Lp[1] = e1; Lp[2] = e1; Lp[3] = e1;
        keys[row] = ~0ull;
      }
    } else if (vm) {
      if (victim) { k = ~0ull; st = 0u; }
      if (MODE == 2) {           // the header lane clears the stamp bytes of its bucket's victims
        const uint32_t vb = (uint32_t)(vm >> (lane & ~7u)) & 0xFFu;
        if (header) for (uint32_t i = 1; i < 8; ++i) if ((vb >> i) & 1u) k &= ~(0xFFull << (8u * i));
      }
      keys[row] = k;
      if (MODE == 1) stamps[row] = st;
    }
  }
  if (lane == 0 && gone) atomicAdd(&out[(wave & 1023u) * 8u], (unsigned long long)gone);   // (1024 counter lines: same-address atomics queue ~12 ns each)
}


// E0 with U chunks of 64 slots in flight per wavefront (the scan is latency-bound otherwise); VICT = false: the scan alone
template <int U, bool VICT, int POS = 0>
__global__ __launch_bounds__(256) void evict_u(unsigned long long* keys, uint32_t* lists, uint32_t* stamps, uint32_t slots, uint64_t seed, unsigned long long* out) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  uint32_t gone = 0;
  for (uint32_t base = wave * 64u * U; base < slots; base += nwaves * 64u * U) {
    unsigned long long k[U]; uint32_t st[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { k[u] = keys[base + u * 64u + lane]; st[u] = stamps[base + u * 64u + lane]; }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t row = base + u * 64u + lane;
      const bool header = (row & 7u) == 0u;
      bool victim = !header && (mix(seed + row) & 7ull) == 0ull && (k[u] | st[u] | 1ull) != 0ull;
      if (POS) victim = (row & 7u) == 1u && (mix(seed + row) & 7ull) != 0ull && (k[u] | st[u] | 1ull) != 0ull;   // (7/8 of the position-1 slots: as many victims)
      gone += (uint32_t)__builtin_popcountll(__ballot(victim));
      if (VICT && victim) {
        u32x4* Lp = (u32x4*)(lists + (size_t)(POS == 2 ? row ^ ((row >> 3) & 7u) : row) * 16u);
        const u32x4 e0 = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u}, e1 = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
        Lp[0] = e0; Lp[1] = e1; Lp[2] = e1; Lp[3] = e1;
        keys[row] = ~0ull;
      }
    }
  }
  if (lane == 0 && gone) atomicAdd(&out[(wave & 1023u) * 8u], (unsigned long long)gone);   // (1024 counter lines: same-address atomics queue ~12 ns each)
}


// One-line layouts (what an index whose small sets live INSIDE the bucket line would cost): `pairs` = 16 bytes per slot {key, meta}, a
// bucket of 8 slots = one 128-byte line.  A new key: CAS on the key word + an 8-byte store of {stamp, first pod, count} beside it.
//   LOCAL = false: agent-scope CAS, sc1 store (visible to every XCD inside the kernel, as the landed protocol needs)
//   LOCAL = true : workgroup-scope CAS, plain store: both execute in THIS XCD's L2 (only legal if the pairs were routed to the XCD that
//                  owns the bucket; timing only)
template <bool LOCAL>
__global__ __launch_bounds__(256) void newkey_line(unsigned long long* pairs, uint32_t slots, uint32_t n, uint64_t seed) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const uint32_t slot = (uint32_t)(mix(seed + k) % slots) | 1u;
  const unsigned long long h = mix(seed * 3u + k) | 2ull;
  unsigned long long* e = pairs + (size_t)slot * 2u;
  unsigned long long expect = e[0] & 0ull;
  const unsigned long long meta = (seed << 32) | (1ull << 16) | (k & 4095u);
  if (LOCAL) {
    __hip_atomic_compare_exchange_strong(e, &expect, h, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    e[1] = meta;
  } else {
    __hip_atomic_compare_exchange_strong(e, &expect, h, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(e + 1, meta, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// The landed protocol with every operation XCD-local (workgroup-scope CAS, plain stores): what device scope costs
__global__ __launch_bounds__(256) void newkey_local(unsigned long long* keys, uint32_t* lists, uint32_t* stamps, uint32_t slots, uint32_t n, uint64_t seed) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const uint32_t slot = (uint32_t)(mix(seed + k) % slots) | 1u;
  const unsigned long long h = mix(seed * 3u + k) | 2ull;
  unsigned long long expect = keys[slot] & 0ull;
  __hip_atomic_compare_exchange_strong(&keys[slot], &expect, h, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  const u32x4 first = {0xFFFF0000u | (k & 4095u), 0xFFFFFFFFu, 0xFFFFFFFFu, 1u};
  *(u32x4*)(lists + (size_t)slot * 16u) = first;
  stamps[slot] = (uint32_t)seed;
}

// ageing over the one-line layout: the scan reads 16 bytes per slot, a victim is rewritten in place (key = tombstone, meta = 0)
__global__ __launch_bounds__(256) void evict_line(unsigned long long* pairs, uint32_t slots, uint64_t seed, unsigned long long* out) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
  uint32_t gone = 0;
  for (uint32_t base = wave * 256u; base < slots; base += nwaves * 256u) {
    u64x2 e[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) e[u] = *(const u64x2*)(pairs + (size_t)(base + u * 64u + lane) * 2u);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint32_t row = base + u * 64u + lane;
      const bool victim = (row & 7u) != 0u && (mix(seed + row) & 7ull) == 0ull && (e[u].x | e[u].y | 1ull) != 0ull;
      gone += (uint32_t)__builtin_popcountll(__ballot(victim));
      if (victim) { const u64x2 t = {~0ull, 0ull}; *(u64x2*)(pairs + (size_t)row * 2u) = t; }
    }
  }
  if (lane == 0 && gone) atomicAdd(&out[(wave & 1023u) * 8u], (unsigned long long)gone);
}

template <typename F>
int timeit(const char* what, double per, F launch) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  launch(1);
  CK(hipEventRecord(a));
  const int reps = 4;
  for (int i = 0; i < reps; ++i) launch(i + 2);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  printf("%-72s %8.1f us per launch = %6.1f us per Mi\n", what, ms * 1e3 / reps, ms * 1e3 / reps / per);
  return 0;
}

int main() {
  const uint32_t slots = 16u << 20, n = 2u << 20;
  unsigned long long *keys, *out; uint32_t *lists, *stamps;
  CK(hipMalloc(&keys, (size_t)slots * 8)); CK(hipMemset(keys, 0, (size_t)slots * 8));
  CK(hipMalloc(&stamps, (size_t)slots * 4)); CK(hipMemset(stamps, 0, (size_t)slots * 4));
  CK(hipMalloc(&lists, (size_t)slots * 64)); CK(hipMemset(lists, 0xFF, (size_t)slots * 64));
  CK(hipMalloc(&out, 65536)); CK(hipMemset(out, 0, 65536));
  const int grid = (int)(n / 256u);
  timeit("A0 new key: CAS + 16 B list store + stamp store (landed)", 2.0, [&](int i) { hipLaunchKernelGGL(newkey<0>, dim3(grid), dim3(256), 0, 0, keys, lists, stamps, slots, n, (uint64_t)i << 32); });
  timeit("A1 new key: CAS + 64 B list line + wait + stamp store", 2.0, [&](int i) { hipLaunchKernelGGL(newkey<1>, dim3(grid), dim3(256), 0, 0, keys, lists, stamps, slots, n, (uint64_t)(i + 10) << 32); });
  timeit("B0 new key: CAS + 16 B list store + byte into the bucket header", 2.0, [&](int i) { hipLaunchKernelGGL(newkey<2>, dim3(grid), dim3(256), 0, 0, keys, lists, stamps, slots, n, (uint64_t)(i + 20) << 32); });
  timeit("B1 new key: CAS + 64 B list line + wait + byte into the bucket header", 2.0, [&](int i) { hipLaunchKernelGGL(newkey<3>, dim3(grid), dim3(256), 0, 0, keys, lists, stamps, slots, n, (uint64_t)(i + 30) << 32); });
  const int eg = 2048;
  timeit("E0 ageing: scan keys + stamps, victim = list line + key word (landed; per Mi victims)", 2.0, [&](int i) { hipLaunchKernelGGL(evict<0>, dim3(eg), dim3(256), 0, 0, keys, lists, stamps, slots, (uint64_t)(i + 40) << 32, out); });
  timeit("E1 ageing: scan keys + stamps, wavefront rewrites keys + stamps (per Mi victims)", 2.0, [&](int i) { hipLaunchKernelGGL(evict<1>, dim3(eg), dim3(256), 0, 0, keys, lists, stamps, slots, (uint64_t)(i + 50) << 32, out); });
  timeit("E2 ageing: scan keys, wavefront rewrites keys (stamps in the header; per Mi victims)", 2.0, [&](int i) { hipLaunchKernelGGL(evict<2>, dim3(eg), dim3(256), 0, 0, keys, lists, stamps, slots, (uint64_t)(i + 60) << 32, out); });

  timeit("S1 scan alone, 1 chunk in flight", 2.0, [&](int i) { hipLaunchKernelGGL((evict_u<1, false>), dim3(eg), dim3(256), 0, 0, keys, lists, stamps, slots, (uint64_t)(i + 70) << 32, out); });
  timeit("S4 scan alone, 4 chunks in flight", 2.0, [&](int i) { hipLaunchKernelGGL((evict_u<4, false>), dim3(eg), dim3(256), 0, 0, keys, lists, stamps, slots, (uint64_t)(i + 70) << 32, out); });
  timeit("S8 scan alone, 8 chunks in flight", 2.0, [&](int i) { hipLaunchKernelGGL((evict_u<8, false>), dim3(eg), dim3(256), 0, 0, keys, lists, stamps, slots, (uint64_t)(i + 70) << 32, out); });
  timeit("E0u2 landed protocol, 2 chunks in flight", 2.0, [&](int i) { hipLaunchKernelGGL((evict_u<2, true>), dim3(eg), dim3(256), 0, 0, keys, lists, stamps, slots, (uint64_t)(i + 80) << 32, out); });
  timeit("E0u4 landed protocol, 4 chunks in flight", 2.0, [&](int i) { hipLaunchKernelGGL((evict_u<4, true>), dim3(eg), dim3(256), 0, 0, keys, lists, stamps, slots, (uint64_t)(i + 90) << 32, out); });
  timeit("E0u8 landed protocol, 8 chunks in flight", 2.0, [&](int i) { hipLaunchKernelGGL((evict_u<8, true>), dim3(eg), dim3(256), 0, 0, keys, lists, stamps, slots, (uint64_t)(i + 100) << 32, out); });
  timeit("E0u4 landed protocol, 4 chunks in flight, 1024 workgroups", 2.0, [&](int i) { hipLaunchKernelGGL((evict_u<4, true>), dim3(1024), dim3(256), 0, 0, keys, lists, stamps, slots, (uint64_t)(i + 110) << 32, out); });
  timeit("E0u4 landed protocol, 4 chunks in flight, 4096 workgroups", 2.0, [&](int i) { hipLaunchKernelGGL((evict_u<4, true>), dim3(4096), dim3(256), 0, 0, keys, lists, stamps, slots, (uint64_t)(i + 120) << 32, out); });

  timeit("A0p new key as landed, every key in position 1 of its bucket", 2.0, [&](int i) { hipLaunchKernelGGL(newkey<4>, dim3(grid), dim3(256), 0, 0, keys, lists, stamps, slots, n, (uint64_t)(i + 130) << 32); });
  timeit("A0x the same, list row = slot ^ (bucket & 7)", 2.0, [&](int i) { hipLaunchKernelGGL(newkey<5>, dim3(grid), dim3(256), 0, 0, keys, lists, stamps, slots, n, (uint64_t)(i + 140) << 32); });
  timeit("E0p landed protocol, victims in position 1 of their buckets", 2.0, [&](int i) { hipLaunchKernelGGL((evict_u<1, true, 1>), dim3(eg), dim3(256), 0, 0, keys, lists, stamps, slots, (uint64_t)(i + 150) << 32, out); });
  timeit("E0x the same, list row = slot ^ (bucket & 7)", 2.0, [&](int i) { hipLaunchKernelGGL((evict_u<1, true, 2>), dim3(eg), dim3(256), 0, 0, keys, lists, stamps, slots, (uint64_t)(i + 160) << 32, out); });
  {
    unsigned long long* pairs; CK(hipMalloc(&pairs, (size_t)slots * 16)); CK(hipMemset(pairs, 0, (size_t)slots * 16));
    timeit("C0 one-line layout: agent-scope CAS + 8 B meta store in the same 128 B line", 2.0, [&](int i) { hipLaunchKernelGGL(newkey_line<false>, dim3(grid), dim3(256), 0, 0, pairs, slots, n, (uint64_t)(i + 170) << 32); });
    timeit("C1 one-line layout, XCD-local CAS + plain store (timing only)", 2.0, [&](int i) { hipLaunchKernelGGL(newkey_line<true>, dim3(grid), dim3(256), 0, 0, pairs, slots, n, (uint64_t)(i + 180) << 32); });
    timeit("A0l landed layout, XCD-local CAS + plain stores (timing only)", 2.0, [&](int i) { hipLaunchKernelGGL(newkey_local, dim3(grid), dim3(256), 0, 0, keys, lists, stamps, slots, n, (uint64_t)(i + 190) << 32); });
    timeit("E3 ageing over the one-line layout: 16 B per slot scanned, victims rewritten in place (per Mi victims)", 2.0, [&](int i) { hipLaunchKernelGGL(evict_line, dim3(eg), dim3(256), 0, 0, pairs, slots, (uint64_t)(i + 200) << 32, out); });
    CK(hipFree(pairs));
  }
  static unsigned long long hh[8192]; CK(hipMemcpy(hh, out, 65536, hipMemcpyDeviceToHost));
  unsigned long long h = 0; for (int i = 0; i < 8192; ++i) h += hh[i];
  printf("victims counted: %llu (7 launches x ~1.75 Mi x 5 incl. warm-up)\n", h);
  return 0;
}
