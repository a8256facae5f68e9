// Micro-benchmark (measurement tool, not product code): the HBM ceiling for the prefix-index access pattern --
// random 512-byte rows out of a table far larger than L2 + MALL.  Every wavefront keeps ROWS independent row loads in
// flight; no other work.  Prints achieved TB/s for several in-flight depths.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <random>

template <int ROWS>
__global__ __launch_bounds__(256) void gather(const uint64_t* __restrict__ tab, const uint32_t* __restrict__ idx, uint64_t* out, uint32_t n_batches) {
  const int lane = threadIdx.x & 63;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  uint64_t acc = 0;
  for (uint32_t b = wave; b < n_batches; b += nwaves) {
    uint64_t w[ROWS];
    const uint32_t mine = idx[(size_t)b * 64 + lane];
#pragma unroll
    for (int u = 0; u < ROWS; ++u) {
      const uint32_t s = __builtin_amdgcn_readlane(mine, u);
      w[u] = tab[(size_t)s * 64 + lane];
    }
#pragma unroll
    for (int u = 0; u < ROWS; ++u) acc ^= w[u];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int ROWS>
int run(const uint64_t* tab, const uint32_t* idx, uint64_t* out, uint32_t n_batches, int blocks_per_cu) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int grid = 256 * blocks_per_cu;
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(gather<ROWS>, dim3(grid), dim3(256), 0, 0, tab, idx, out, n_batches);
  CK(hipEventRecord(a));
  const int reps = 10;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(gather<ROWS>, dim3(grid), dim3(256), 0, 0, tab, idx, out, n_batches);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  const double bytes = (double)n_batches * ROWS * 512.0 * reps;
  printf("rows_in_flight_per_wave=%d waves_per_simd=%d  %.2f TB/s  (%.1f us per %u row reads)\n", ROWS, blocks_per_cu, bytes / (ms * 1e-3) / 1e12,
         ms * 1e3 / reps, n_batches * ROWS);
  return 0;
}

int main() {
  const size_t n_rows = 2u << 20;                 // 2M rows x 512 B = 1 GiB
  const uint32_t n_batches = 65536;               // x ROWS rows each
  uint64_t* tab; uint32_t* idx; uint64_t* out;
  CK(hipMalloc(&tab, n_rows * 512)); CK(hipMemset(tab, 1, n_rows * 512));
  std::vector<uint32_t> h((size_t)n_batches * 64);
  std::mt19937 rng(1); for (auto& v : h) v = rng() % n_rows;
  CK(hipMalloc(&idx, h.size() * 4)); CK(hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&out, 256 * 8 * 256 * 8));
  for (int bpc : {2, 4, 8}) { run<8>(tab, idx, out, n_batches, bpc); run<16>(tab, idx, out, n_batches, bpc); run<32>(tab, idx, out, n_batches, bpc); }
  return 0;
}
