// Micro-benchmark (measurement tool): the shader clock a kernel really runs at.  A wavefront executes a chain of dependent
// VALU ops; cycles = clock64() delta (s_memtime), time = wall_clock64() (constant-rate counter) and HIP events.  Run as short
// kernels (the shape of the pick launches) and as one long kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void spin(uint32_t n, uint64_t* out) {
  const uint64_t t0 = clock64(), w0 = wall_clock64();
  uint32_t a = threadIdx.x, b = blockIdx.x;
  for (uint32_t i = 0; i < n; ++i) { a = a * 3u + b; b = b * 5u + a; a ^= b >> 3; b += a; }   // dependent chain
  const uint64_t t1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = w1 - w0; out[2] = a + b; }
}

int main() {
  uint64_t* d; uint64_t h[3];
  CK(hipMalloc(&d, 64));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int wrate = 0; CK(hipDeviceGetAttribute(&wrate, hipDeviceAttributeWallClockRate, 0));   // kHz
  int crate = 0; CK(hipDeviceGetAttribute(&crate, hipDeviceAttributeClockRate, 0));       // kHz (max shader clock)
  printf("wall clock rate %d kHz, max shader clock %d kHz\n", wrate, crate);
  auto run = [&](uint32_t n, int grid, int threads, const char* name) {
    (void)hipEventRecord(e0); hipLaunchKernelGGL(spin, dim3(grid), dim3(threads), 0, 0, n, d); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1); (void)hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    const double us_wall = (double)h[1] / (wrate * 1e-3);
    printf("%-28s n=%u grid=%d x %d: event %.1f us, s_memtime delta %llu, wall_clock delta %llu (%.1f us) -> s_memtime ticks at %.3f GHz; %.2f ns per iteration\n",
           name, n, grid, threads, ms * 1e3, (unsigned long long)h[0], (unsigned long long)h[1], us_wall, (double)h[0] / (us_wall * 1e3), us_wall * 1e3 / n);
  };
  run(4000, 1024, 256, "cold short (first launch)");
  for (int i = 0; i < 4; ++i) run(4000, 1024, 256, "short");
  run(4000000, 1024, 256, "long (sustained)");
  for (int i = 0; i < 3; ++i) run(4000, 1024, 256, "short after long");
  // one wave per SIMD: the dependent chain's latency per op (ns) -> cycles per op at the clock above
  run(1000000, 1024, 64, "1 wave per SIMD");
  run(1000000, 1024, 256, "4 waves per SIMD");
  run(1000000, 2048, 512, "16 waves per SIMD (2 blocks/CU)");
  return 0;
}
