// Micro-benchmark (measurement tool, not product code): does ANY load flavour or allocation type make the memory side fetch a random
// 64-byte line as a 64-byte request?  The cold-index pick kernel moves 1.9 x its algorithmic bytes because every 64-byte bucket / pod
// list arrives as a 128-byte HBM request (profiles/pmc_traffic_cold.json, profiles/r02_fetchcal.txt).  Random 64-byte lines (four
// lanes x 16 bytes), 4 independent loads in flight per lane, out of 1.5 GiB:
//   allocation:  hipMalloc | hipExtMallocWithFlags(hipDeviceMallocUncached) | (hipDeviceMallocFinegrained)
//   load:        global_load_dwordx4 with every combination of the gfx950 cache-policy bits sc0 / sc1 / nt
// One kernel NAME per (allocation, flavour), so that `rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum` (and _64B / _128B in a
// second pass) attributes the request sizes per variant (scripts/pmc_summary.py <dir> gather2 --by-kernel).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <random>

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int FLAVOUR>
__device__ __forceinline__ void load16(u32x4_t& v, const uint8_t* p) {
  if constexpr (FLAVOUR == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  if constexpr (FLAVOUR == 1) asm volatile("global_load_dwordx4 %0, %1, off sc0" : "=v"(v) : "v"(p) : "memory");
  if constexpr (FLAVOUR == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  if constexpr (FLAVOUR == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
  if constexpr (FLAVOUR == 4) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
  if constexpr (FLAVOUR == 5) asm volatile("global_load_dwordx4 %0, %1, off sc0 nt" : "=v"(v) : "v"(p) : "memory");
  if constexpr (FLAVOUR == 6) asm volatile("global_load_dwordx4 %0, %1, off sc1 nt" : "=v"(v) : "v"(p) : "memory");
  if constexpr (FLAVOUR == 7) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt" : "=v"(v) : "v"(p) : "memory");
}

template <int FLAVOUR, int ALLOC>      // (ALLOC only names the kernel)
__global__ __launch_bounds__(256) void gather2(const uint8_t* __restrict__ tab, const uint32_t* __restrict__ idx, uint64_t* out, uint32_t n_groups) {
  constexpr int DEPTH = 4;
  const uint32_t lane = threadIdx.x & 63u, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  uint32_t acc = 0;
  for (uint32_t g = wave * DEPTH; g + DEPTH <= n_groups; g += nwaves * DEPTH) {
    u32x4_t v[DEPTH];
    uint32_t line[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) line[d] = idx[(size_t)(g + d) * 16u + (lane >> 2)];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) load16<FLAVOUR>(v[d], tab + (size_t)line[d] * 64u + (lane & 3u) * 16u);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) acc ^= v[d].x ^ v[d].y ^ v[d].z ^ v[d].w;
  }
  if (acc == 0x12345u) out[0] = acc;
}

template <int FLAVOUR, int ALLOC>
int run(const char* aname, const uint8_t* tab, const uint32_t* idx, uint64_t* out, uint32_t n_groups) {
  static const char* fl[8] = {"(none)", "sc0", "sc1", "sc0 sc1", "nt", "sc0 nt", "sc1 nt", "sc0 sc1 nt"};
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int grid = 256 * 8;
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((gather2<FLAVOUR, ALLOC>), dim3(grid), dim3(256), 0, 0, tab, idx, out, n_groups);
  CK(hipEventRecord(a));
  const int reps = 5;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((gather2<FLAVOUR, ALLOC>), dim3(grid), dim3(256), 0, 0, tab, idx, out, n_groups);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  const double bytes = (double)(n_groups / 4 * 4) * 16.0 * 64.0 * reps;
  printf("gather2<%d,%d>  alloc %-12s load flags %-11s  %.2f TB/s of 64-byte lines  (%.1f us per %u line reads)\n", FLAVOUR, ALLOC, aname, fl[FLAVOUR],
         bytes / (ms * 1e-3) / 1e12, ms * 1e3 / reps, n_groups * 16u);
  return 0;
}

template <int ALLOC>
int run_all(const char* aname, const uint8_t* tab, const uint32_t* idx, uint64_t* out, uint32_t n_groups) {
  return run<0, ALLOC>(aname, tab, idx, out, n_groups) | run<1, ALLOC>(aname, tab, idx, out, n_groups) | run<2, ALLOC>(aname, tab, idx, out, n_groups) |
         run<3, ALLOC>(aname, tab, idx, out, n_groups) | run<4, ALLOC>(aname, tab, idx, out, n_groups) | run<5, ALLOC>(aname, tab, idx, out, n_groups) |
         run<6, ALLOC>(aname, tab, idx, out, n_groups) | run<7, ALLOC>(aname, tab, idx, out, n_groups);
}

int main() {
  const size_t n_lines = 24u << 20;               // 24 Mi lines x 64 B = 1.5 GiB
  const uint32_t n_groups = 1u << 18;             // x 16 lines = 4 Mi line reads = 256 MiB per launch
  uint32_t* idx; uint64_t* out;
  std::vector<uint32_t> h((size_t)n_groups * 16);
  std::mt19937 rng(1); for (auto& v : h) v = rng() % n_lines;
  CK(hipMalloc(&idx, h.size() * 4)); CK(hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&out, 64));
  {
    uint8_t* tab; CK(hipMalloc(&tab, n_lines * 64)); CK(hipMemset(tab, 1, n_lines * 64)); CK(hipDeviceSynchronize());
    if (run_all<0>("hipMalloc", tab, idx, out, n_groups)) return 1;
    CK(hipFree(tab));
  }
  {
    uint8_t* tab = nullptr;
    hipError_t e = hipExtMallocWithFlags((void**)&tab, n_lines * 64, hipDeviceMallocUncached);
    if (e != hipSuccess) { printf("hipDeviceMallocUncached: %s\n", hipGetErrorString(e)); (void)hipGetLastError(); }
    else { CK(hipMemset(tab, 1, n_lines * 64)); CK(hipDeviceSynchronize()); if (run_all<1>("uncached", tab, idx, out, n_groups)) return 1; CK(hipFree(tab)); }
  }
  {
    uint8_t* tab = nullptr;
    hipError_t e = hipExtMallocWithFlags((void**)&tab, n_lines * 64, hipDeviceMallocFinegrained);
    if (e != hipSuccess) { printf("hipDeviceMallocFinegrained: %s\n", hipGetErrorString(e)); (void)hipGetLastError(); }
    else { CK(hipMemset(tab, 1, n_lines * 64)); CK(hipDeviceSynchronize()); if (run_all<2>("fine-grained", tab, idx, out, n_groups)) return 1; CK(hipFree(tab)); }
  }
  return 0;
}
