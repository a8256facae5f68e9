// Micro-benchmark (measurement tool, not product code), round 4: what a NEW key and an eviction VICTIM cost when the index does NOT sit in
// the 256 MB Infinity Cache -- claimcost.hip (round 3) re-ran its kernels back to back over tables of 192 MB (keys + stamps), which the
// cache holds; in the closed loop of bench.py the pick and update kernels of a step push them out again before the next eviction.
// Here 1 GiB of unrelated memory is rewritten in front of every timed launch (`flush`), and the variants price the protocol changes
// considered for round 4:
//   new key   N0  CAS + 16 B list store + 4 B stamp store                                  (round 3 as landed)
//             N1  CAS + 16 B list store + 1 byte into the bucket header                    (stamps as header bytes)
//             N2  CAS + 64 B list line (four 16 B stores) + header byte, no wait           (claimer writes the whole line: victims keep theirs)
//             N3  CAS + 64 B list line + s_waitcnt + header byte                          (... ordered: list visible before the tag)
//             N4  CAS + 16 B list store + s_waitcnt + header byte
//   ageing    E0  scan keys + stamps; victim = 64 B list line + 8 B key word                (as landed)
//             E1  scan keys (tags in the headers); victim = 64 B list line + key word
//             E2  scan keys; victim = key word only                                         (list line left to the next claimer)
//             E3  scan keys; victim = key word + its header byte
// 16 Mi slots (keys 128 MB, stamps 64 MB, lists 1 GiB), 1 Mi new keys / 2 Mi victims per launch (the closed loop's numbers).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint64_t mix(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
}

__global__ void flush_kernel(uint64_t* p, size_t n, uint64_t v) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v + i;
}

template <int MODE>
__global__ __launch_bounds__(256) void newkey(unsigned long long* keys, uint32_t* lists, uint32_t* stamps, uint32_t slots, uint32_t n, uint64_t seed) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const uint32_t slot = ((uint32_t)(mix(seed + k) % slots) & ~7u) | 1u;       // buckets fill front to back: a new key sits in position 1, as a rule
  const unsigned long long h = mix(seed * 3u + k) | 2ull;
  // the search: the whole bucket line by four 16-byte loads (as index_insert_one does), then the claim
  const u32x4* B4 = (const u32x4*)(keys + (slot & ~7u));
  const u32x4 b0 = B4[0], b1 = B4[1], b2 = B4[2], b3 = B4[3];
  const unsigned long long seen = ((unsigned long long)b0.w << 32) | b0.z;
  const unsigned long long junk = (b1.x ^ b2.y ^ b3.z) & 0ull;
  if (MODE < 5) atomicCAS(&keys[slot], seen + junk, h);
  if (MODE >= 5) {
    // round-5 candidates: single-pod sets INLINE in the bucket (docs/next/bucket_with_inline_sets.md): no list line at all
    //   N5  CAS + ONE 4-byte meta store {tag, flags, pod} into the bucket line     N6  CAS only (floor)     N7  bucket load only
    //   N8  as N5 with the 5-key layout's addresses (keys at bytes 24..63, meta at 4..23)
    if (MODE == 7) { if (seen + junk == 0x1234567ull) keys[slot] = h; return; }
    if (MODE == 8) {
      unsigned long long* kw = keys + (slot & ~7u) + 3u;            // key[1] of the 5-key layout: byte 24 of the bucket
      atomicCAS(kw, ((unsigned long long)b1.w << 32 | b1.z) + junk, h);
      __hip_atomic_store((uint32_t*)(keys + (slot & ~7u)) + 1, (uint32_t)((seed & 0xFFu) << 24) | (k & 4095u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
    atomicCAS(&keys[slot], seen + junk, h);
    if (MODE == 5) __hip_atomic_store((uint32_t*)(keys + (slot & ~7u)) + 1, (uint32_t)((seed & 0xFFu) << 24) | (k & 4095u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  uint32_t* L = lists + (size_t)slot * 16u;
  const u32x4 first = {0xFFFF0000u | (k & 4095u), 0xFFFFFFFFu, 0xFFFFFFFFu, 1u}, rest = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
  if (MODE == 2 || MODE == 3)
    asm volatile("global_store_dwordx4 %0, %1, off offset:16 sc1\n\tglobal_store_dwordx4 %0, %1, off offset:32 sc1\n\tglobal_store_dwordx4 %0, %1, off offset:48 sc1\n\ts_nop 1" ::"v"(L), "v"(rest) : "memory");
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(L), "v"(first) : "memory");
  if (MODE == 3 || MODE == 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (MODE == 0) __hip_atomic_store(&stamps[slot], (uint32_t)seed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else __hip_atomic_store((uint8_t*)&keys[slot & ~7u] + (slot & 7u), (uint8_t)seed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int MODE>
__global__ __launch_bounds__(256) void evict(unsigned long long* keys, uint32_t* lists, const uint32_t* stamps, uint32_t slots, uint64_t seed, unsigned long long* out) {
  constexpr int U = 4;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  uint32_t gone = 0;
  for (uint32_t base = wave * 64u * U; base < slots; base += nwaves * 64u * U) {
    unsigned long long k[U]; uint32_t st[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { k[u] = keys[base + u * 64u + lane]; st[u] = MODE == 0 ? stamps[base + u * 64u + lane] : 0u; }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t row = base + u * 64u + lane;
      const bool victim = (row & 7u) == 1u && (mix(seed + row) & 7ull) != 0ull && (k[u] | st[u] | 1ull) != 0ull;   // 7/8 of the position-1 slots: 1.75 Mi... x 8/7 below
      gone += (uint32_t)__builtin_popcountll(__ballot(victim));
      if (victim) {
        if (MODE <= 1) {
          u32x4* Lp = (u32x4*)(lists + (size_t)row * 16u);
          const u32x4 e0 = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u}, e1 = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
          Lp[0] = e0; Lp[1] = e1; Lp[2] = e1; Lp[3] = e1;
        }
        keys[row] = ~0ull;
        if (MODE == 3) *((uint8_t*)&keys[row & ~7u] + (row & 7u)) = 0;
      }
    }
  }
  if (lane == 0 && gone) atomicAdd(&out[(wave & 1023u) * 8u], (unsigned long long)gone);
}

int main() {
  const uint32_t slots = 16u << 20, n = 1u << 20;
  unsigned long long *keys, *out; uint32_t *lists, *stamps; uint64_t* fl;
  CK(hipMalloc(&keys, (size_t)slots * 8)); CK(hipMemset(keys, 0, (size_t)slots * 8));
  CK(hipMalloc(&stamps, (size_t)slots * 4)); CK(hipMemset(stamps, 0, (size_t)slots * 4));
  CK(hipMalloc(&lists, (size_t)slots * 64)); CK(hipMemset(lists, 0xFF, (size_t)slots * 64));
  CK(hipMalloc(&out, 65536)); CK(hipMemset(out, 0, 65536));
  const size_t fl_n = (size_t)1 << 27;                        // 1 GiB
  CK(hipMalloc(&fl, fl_n * 8));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  auto timeit = [&](const char* what, double per_mi, auto launch) -> int {
    float tot = 0; const int reps = 5;
    for (int i = 0; i < reps + 1; ++i) {
      hipLaunchKernelGGL(flush_kernel, dim3(4096), dim3(256), 0, 0, fl, fl_n, (uint64_t)i);
      CK(hipEventRecord(a));
      launch(i + 1);
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      if (i) tot += ms;                                        // (the first launch warms the code up)
    }
    printf("%-100s %8.1f us per launch = %6.1f us per Mi\n", what, tot * 1e3 / reps, tot * 1e3 / reps / per_mi);
    return 0;
  };
  const int grid = (int)(n / 256u);
  timeit("N0 new key: bucket load + CAS + 16 B list store + 4 B stamp store (landed)", 1.0, [&](int i) { hipLaunchKernelGGL(newkey<0>, dim3(grid), dim3(256), 0, 0, keys, lists, stamps, slots, n, (uint64_t)(i + 0) << 32); });
  timeit("N1 new key: ... + 16 B list store + header byte", 1.0, [&](int i) { hipLaunchKernelGGL(newkey<1>, dim3(grid), dim3(256), 0, 0, keys, lists, stamps, slots, n, (uint64_t)(i + 10) << 32); });
  timeit("N2 new key: ... + 64 B list line + header byte, no wait", 1.0, [&](int i) { hipLaunchKernelGGL(newkey<2>, dim3(grid), dim3(256), 0, 0, keys, lists, stamps, slots, n, (uint64_t)(i + 20) << 32); });
  timeit("N3 new key: ... + 64 B list line + s_waitcnt + header byte", 1.0, [&](int i) { hipLaunchKernelGGL(newkey<3>, dim3(grid), dim3(256), 0, 0, keys, lists, stamps, slots, n, (uint64_t)(i + 30) << 32); });
  timeit("N4 new key: ... + 16 B list store + s_waitcnt + header byte", 1.0, [&](int i) { hipLaunchKernelGGL(newkey<4>, dim3(grid), dim3(256), 0, 0, keys, lists, stamps, slots, n, (uint64_t)(i + 40) << 32); });
  timeit("N5 new key, inline set: bucket load + CAS + ONE 4 B meta store into the bucket line", 1.0, [&](int i) { hipLaunchKernelGGL(newkey<5>, dim3(grid), dim3(256), 0, 0, keys, lists, stamps, slots, n, (uint64_t)(i + 100) << 32); });
  timeit("N6 floor: bucket load + CAS", 1.0, [&](int i) { hipLaunchKernelGGL(newkey<6>, dim3(grid), dim3(256), 0, 0, keys, lists, stamps, slots, n, (uint64_t)(i + 110) << 32); });
  timeit("N7 floor: bucket load only", 1.0, [&](int i) { hipLaunchKernelGGL(newkey<7>, dim3(grid), dim3(256), 0, 0, keys, lists, stamps, slots, n, (uint64_t)(i + 120) << 32); });
  timeit("N8 new key, inline set, 5-key layout addresses (key at byte 24, meta at byte 4)", 1.0, [&](int i) { hipLaunchKernelGGL(newkey<8>, dim3(grid), dim3(256), 0, 0, keys, lists, stamps, slots, n, (uint64_t)(i + 130) << 32); });
  timeit("N1 again (order check)", 1.0, [&](int i) { hipLaunchKernelGGL(newkey<1>, dim3(grid), dim3(256), 0, 0, keys, lists, stamps, slots, n, (uint64_t)(i + 140) << 32); });
  const int eg = 2048;
  const double vic = 2.0 * 7.0 / 8.0;                          // 2 Mi position-1 slots x 7/8 = 1.75 Mi victims per launch
  timeit("E0 ageing: scan keys + stamps; victim = 64 B list line + key word (landed; per Mi victims)", vic, [&](int i) { hipLaunchKernelGGL(evict<0>, dim3(eg), dim3(256), 0, 0, keys, lists, (const uint32_t*)stamps, slots, (uint64_t)(i + 50) << 32, out); });
  timeit("E1 ageing: scan keys; victim = 64 B list line + key word", vic, [&](int i) { hipLaunchKernelGGL(evict<1>, dim3(eg), dim3(256), 0, 0, keys, lists, (const uint32_t*)stamps, slots, (uint64_t)(i + 60) << 32, out); });
  timeit("E2 ageing: scan keys; victim = key word only", vic, [&](int i) { hipLaunchKernelGGL(evict<2>, dim3(eg), dim3(256), 0, 0, keys, lists, (const uint32_t*)stamps, slots, (uint64_t)(i + 70) << 32, out); });
  timeit("E3 ageing: scan keys; victim = key word + header byte", vic, [&](int i) { hipLaunchKernelGGL(evict<3>, dim3(eg), dim3(256), 0, 0, keys, lists, (const uint32_t*)stamps, slots, (uint64_t)(i + 80) << 32, out); });
  static unsigned long long hh[8192]; CK(hipMemcpy(hh, out, 65536, hipMemcpyDeviceToHost));
  unsigned long long h = 0; for (int i = 0; i < 8192; ++i) h += hh[i];
  printf("victims counted: %llu\n", h);
  return 0;
}
