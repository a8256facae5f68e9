// Micro-benchmark (measurement tool, not product code): the HBM ceiling for the pick kernel's COLD access pattern -- random
// 64-byte lines (key buckets, pod lists) out of a table far larger than L2 + Infinity Cache, four lanes x 16 bytes per line,
// DEPTH independent loads in flight per lane.  Prints achieved TB/s per depth / occupancy (the counterpart of rowgather.hip,
// which does the same for 512-byte rows).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <random>

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int DEPTH>
__global__ __launch_bounds__(256) void gather(const uint8_t* __restrict__ tab, const uint32_t* __restrict__ idx, uint64_t* out, uint32_t n_groups) {
  // one "group" = 16 lines per wavefront step (64 lanes / 4 lanes per line); idx holds the line numbers
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)tab, 0, (int)0x7FFFFFFF, 0x00020000);
  const uint32_t lane = threadIdx.x & 63u, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  uint32_t acc = 0;
  for (uint32_t g = wave * DEPTH; g + DEPTH <= n_groups; g += nwaves * DEPTH) {
    u32x4_t v[DEPTH];
    uint32_t line[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) line[d] = idx[(size_t)(g + d) * 16u + (lane >> 2)];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) v[d] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(line[d] * 64u + (lane & 3u) * 16u), 0, 0);
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) acc ^= v[d].x ^ v[d].y ^ v[d].z ^ v[d].w;
  }
  if (acc == 0x12345u) out[0] = acc;
}

template <int DEPTH>
int run(const uint8_t* tab, const uint32_t* idx, uint64_t* out, uint32_t n_groups, int blocks_per_cu) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int grid = 256 * blocks_per_cu;
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(gather<DEPTH>, dim3(grid), dim3(256), 0, 0, tab, idx, out, n_groups);
  CK(hipEventRecord(a));
  const int reps = 5;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(gather<DEPTH>, dim3(grid), dim3(256), 0, 0, tab, idx, out, n_groups);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  const double bytes = (double)(n_groups / DEPTH * DEPTH) * 16.0 * 64.0 * reps;
  printf("lines_in_flight_per_lane=%d waves_per_simd=%d  %.2f TB/s  (%.1f us per %u line reads)\n", DEPTH, blocks_per_cu, bytes / (ms * 1e-3) / 1e12,
         ms * 1e3 / reps, n_groups * 16u);
  return 0;
}

int main() {
  const size_t n_lines = 24u << 20;               // 24 Mi lines x 64 B = 1.5 GiB (one raw descriptor reaches 2 GiB)
  const uint32_t n_groups = 1u << 18;             // x 16 lines = 4 Mi line reads = 256 MiB per launch
  uint8_t* tab; uint32_t* idx; uint64_t* out;
  CK(hipMalloc(&tab, n_lines * 64)); CK(hipMemset(tab, 1, n_lines * 64));
  std::vector<uint32_t> h((size_t)n_groups * 16);
  std::mt19937 rng(1); for (auto& v : h) v = rng() % n_lines;
  CK(hipMalloc(&idx, h.size() * 4)); CK(hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&out, 64));
  for (int bpc : {4, 8}) { run<1>(tab, idx, out, n_groups, bpc); run<2>(tab, idx, out, n_groups, bpc); run<4>(tab, idx, out, n_groups, bpc); run<8>(tab, idx, out, n_groups, bpc); }
  return 0;
}
