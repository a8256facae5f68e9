// Micro-benchmark (measurement tool, not product code): what a post-route index update can cost at best.  A key nobody has seen
// claims LINES random 64-byte lines of a table far larger than L2 + Infinity Cache (bucket word, stamp, row word, list today:
// LINES = 4) with one 8-byte read-modify-write each (atomicOr, as index_insert_kernel does) -- one thread per key, the lines of a
// key independent of each other.  Prints the time per 1 Mi keys for LINES = 1, 2, 3, 4, for plain 8-byte stores instead of
// atomics, for reads alone, and for lines that pair up into 128-byte blocks (what a bucket with its stamps in its second half costs); the table is 6 GiB (three 2 GiB regions so that the stamp / row / list lines of a key sit far from
// its bucket, as in the library's layout).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
}

// MODE 0: atomicOr   1: plain store   2: load only   3: atomicOr, lines 2l and 2l+1 of a key are the two halves of ONE 128-byte block
template <int LINES, int MODE>
__global__ __launch_bounds__(256) void touch(uint64_t* __restrict__ tab, uint64_t n_lines, uint32_t n_keys, uint64_t seed, uint64_t* out) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_keys) return;
  uint64_t acc = 0;
#pragma unroll
  for (int l = 0; l < LINES; ++l) {
    uint64_t line = mix(seed + (uint64_t)k * 4u + (uint64_t)(MODE == 3 ? l >> 1 : l)) % n_lines;
    if (MODE == 3) line = (line & ~1ull) | (uint64_t)(l & 1);
    uint64_t* w = tab + line * 8u + (k & 7u);
    if (MODE == 0 || MODE == 3) atomicOr((unsigned long long*)w, 1ull << (k & 63u));
    else if (MODE == 1) *w = k;
    else acc ^= *w;
  }
  if (MODE == 2 && acc == 0x1234567ull) out[0] = acc;
}

template <int LINES, int MODE>
int run(uint64_t* tab, uint64_t n_lines, uint32_t n_keys, uint64_t* out, const char* what) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int grid = (int)((n_keys + 255u) / 256u);
  hipLaunchKernelGGL((touch<LINES, MODE>), dim3(grid), dim3(256), 0, 0, tab, n_lines, n_keys, 1ull, out);
  CK(hipEventRecord(a));
  const int reps = 4;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((touch<LINES, MODE>), dim3(grid), dim3(256), 0, 0, tab, n_lines, n_keys, (uint64_t)(i + 2) << 40, out);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  const double us = ms * 1e3 / reps, per_mi = us * (1048576.0 / n_keys);
  printf("%-10s lines_per_key=%d  %8.1f us per %u keys  = %6.1f us per Mi keys  (%.2f G lines/s, %.2f TB/s at 192 B per line)\n", what, LINES, us, n_keys,
         per_mi, (double)n_keys * LINES / (us * 1e-6) / 1e9, (double)n_keys * LINES * 192.0 / (us * 1e-6) / 1e12);
  return 0;
}

int main() {
  const uint64_t n_lines = 96ull << 20;           // 96 Mi lines x 64 B = 6 GiB
  const uint32_t n_keys = 2u << 20;               // 2 Mi keys per launch (a 64k x 32-block batch brings 1 Mi new keys and 1 Mi known ones)
  uint64_t *tab, *out;
  CK(hipMalloc(&tab, n_lines * 64)); CK(hipMemset(tab, 0, n_lines * 64));
  CK(hipMalloc(&out, 64));
  run<1, 0>(tab, n_lines, n_keys, out, "atomicOr"); run<2, 0>(tab, n_lines, n_keys, out, "atomicOr"); run<3, 0>(tab, n_lines, n_keys, out, "atomicOr"); run<4, 0>(tab, n_lines, n_keys, out, "atomicOr");
  run<1, 1>(tab, n_lines, n_keys, out, "store"); run<4, 1>(tab, n_lines, n_keys, out, "store");
  run<2, 3>(tab, n_lines, n_keys, out, "or-pair128"); run<4, 3>(tab, n_lines, n_keys, out, "or-pair128");
  run<1, 2>(tab, n_lines, n_keys, out, "load"); run<4, 2>(tab, n_lines, n_keys, out, "load");
  return 0;
}
