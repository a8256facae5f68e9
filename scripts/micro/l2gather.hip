// Micro-benchmark (measurement tool, not product code): what does a gather of 64-byte lines out of an L2-RESIDENT table cost per
// line, as a function of how the 64 lanes of a load instruction are spread over the lines?  The pick kernels probe 32 key
// buckets + up to 32 pod lists per request out of an index that lives in L2; the vector memory pipe of a CU (TA/TCP) looks every
// distinct line of an instruction up separately, so the lane -> line mapping sets the cost, not the bytes.
//   A  lane per line, 4 x 16 B     each lane reads all four 16-byte pieces of ITS OWN line      (64 lines per 4 instructions)
//   B  quad per line, 1 x 16 B     lanes 4q..4q+3 read the four pieces of ONE line              (16 lines per instruction)
//   C  pair per line, 2 x 16 B     lanes 2p, 2p+1 read one half line each, two instructions     (32 lines per 2 instructions)
//   D  lane per line, 1 x 8 B      one 8-byte word of 64 different lines                        (64 lines per instruction)
//   E  row  per line, 1 x 4 B      16 lanes read 64 contiguous bytes                            (4 lines per instruction)
//   F  octet per 128-byte line     lanes 8o..8o+7 read the eight pieces of ONE 128-byte-aligned line (8 double lines per instruction):
//                                  does a 128-byte bucket cost one line or two?  (round 6: pricing a bucket that carries its pod-set ids)
// Prints ns per line and lines per clock per CU (2.0 GHz assumed) at 4 wavefronts per SIMD.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int MODE>
__global__ __launch_bounds__(256) void gather(const uint8_t* __restrict__ tab, uint32_t line_mask, uint64_t* out, uint32_t iters) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)tab, 0, (int)((line_mask + 1u) * 64u), 0x00020000);
  const uint32_t lane = threadIdx.x & 63u, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  uint32_t acc = 0;
  for (uint32_t it = 0; it < iters; ++it) {
    const uint32_t seed = (wave * iters + it) * 64u;
    if constexpr (MODE == 0) {          // A
      const uint32_t off = (mix(seed + lane) & line_mask) * 64u;
      u32x4_t v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(off + 16u * i), 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
    } else if constexpr (MODE == 1) {   // B: 4 instructions x 16 lines
      u32x4_t v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)((mix(seed + i * 16u + (lane >> 2)) & line_mask) * 64u + (lane & 3u) * 16u), 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
    } else if constexpr (MODE == 2) {   // C: 2 x (2 instructions x 32 lines)
      u32x4_t v[4];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const uint32_t off = (mix(seed + h * 32u + (lane >> 1)) & line_mask) * 64u + (lane & 1u) * 32u;
        v[2 * h] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0);
        v[2 * h + 1] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(off + 16u), 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) acc ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
    } else if constexpr (MODE == 3) {   // D: one instruction, 64 lines
      const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)((mix(seed + lane) & line_mask) * 64u + (lane & 7u) * 8u), 0, 0);
      acc ^= v.x ^ v.y;
    } else if constexpr (MODE == 5) {   // F: 4 instructions x 8 double lines
      u32x4_t v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)((mix(seed + i * 8u + (lane >> 3)) & (line_mask >> 1)) * 128u + (lane & 7u) * 16u), 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
    } else {                            // E: 4 instructions x 4 lines
      uint32_t v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b32(rs, (int)((mix(seed + i * 4u + (lane >> 4)) & line_mask) * 64u + (lane & 15u) * 4u), 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc ^= v[i];
    }
  }
  if (acc == 0x12345u) out[0] = acc;
}

template <int MODE>
int run(const char* name, const uint8_t* tab, uint32_t line_mask, uint64_t* out, double lines_per_iter) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int grid = 256 * 4;             // 4 workgroups of 4 wavefronts per CU = 4 wavefronts per SIMD
  const uint32_t iters = 256;
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(gather<MODE>, dim3(grid), dim3(256), 0, 0, tab, line_mask, out, iters);
  CK(hipEventRecord(a));
  const int reps = 5;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(gather<MODE>, dim3(grid), dim3(256), 0, 0, tab, line_mask, out, iters);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  const double lines = (double)grid * 4.0 * iters * lines_per_iter;        // per launch
  const double us = ms * 1e3 / reps;
  printf("%-28s table %5u KiB  %8.1f us  %.3f ns/line  %.3f lines/clk/CU  %.2f TB/s of lines\n", name, (line_mask + 1u) / 16u, us, us * 1e3 / lines * 1.0,
         lines / 256.0 / (us * 1e-6 * 2.0e9), lines * 64.0 / (us * 1e-6) / 1e12);
  return 0;
}

int main() {
  uint8_t* tab; uint64_t* out;
  const size_t bytes = 64u << 20;
  CK(hipMalloc(&tab, bytes)); CK(hipMemset(tab, 1, bytes)); CK(hipMalloc(&out, 64));
  for (uint32_t kib : {512u, 16384u}) {          // L2-resident (per XCD) / Infinity-Cache-resident
    const uint32_t mask = kib * 16u - 1u;
    run<0>("A lane/line 4x16B", tab, mask, out, 64);
    run<1>("B quad/line 1x16B (x4)", tab, mask, out, 64);
    run<2>("C pair/line 2x16B (x2)", tab, mask, out, 64);
    run<3>("D lane/line 1x8B", tab, mask, out, 64);
    run<4>("E row/line 1x4B (x4)", tab, mask, out, 16);
    run<5>("F octet/128-B line (x4)", tab, mask, out, 32);     // (counted in 128-byte lines)
  }
  return 0;
}
