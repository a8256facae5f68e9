// Micro-benchmark (measurement tool, not product code): calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE on KNOWN byte counts
// for the access shapes the pick kernel uses (MI355X_MICROARCH.md "HBM": on gfx950 FETCH_SIZE reports 1/2 of a wide
// coalesced 16 B/lane streaming read; other widths are uncalibrated -- calibrate before trusting an absolute).
//
// Each kernel below reads (or writes) exactly BYTES bytes of a 2 GiB buffer (>> 32 MiB L2 + 256 MiB Infinity Cache), once,
// through a raw buffer descriptor like the pick kernel's loads:
//   stream4   4 B/lane coalesced (picks / table entries)
//   stream8   8 B/lane coalesced (request-row hashes, f64 table entries)
//   stream16  16 B/lane coalesced
//   line64    random 64-byte lines, 16 B/lane x 4 lanes per line (key buckets, pod lists)
//   row512    random 512-byte rows, 8 B/lane x 64 lanes (dense pod-set rows)
//   write4    4 B/lane coalesced stores (picks), write8 (scores)
// Run:  rocprofv3 --kernel-trace --pmc FETCH_SIZE  -- ./fetchcal      (then WRITE_SIZE, TCC_EA0_RDREQ_sum in passes of their own)
// and compare Counter_Value (FETCH_SIZE is in KiB... the CSV says which unit) per kernel with the "bytes=" this program prints.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <random>

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr size_t kBuf = 2ull << 30;         // 2 GiB
constexpr uint32_t kChunk = 1u << 30;       // one descriptor spans 1 GiB (32-bit offsets); kernels touch chunk 0 or 1 by blockIdx parity

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)bytes, 0x00020000);
}

// n_elems elements of W bytes per lane, coalesced, grid-stride
template <int W>
__global__ __launch_bounds__(256) void cal_stream(const uint8_t* buf, uint64_t n_elems, uint64_t* sink) {
  const __amdgpu_buffer_rsrc_t r0 = rsrc(buf, kChunk), r1 = rsrc(buf + kChunk, kChunk);
  uint64_t acc = 0;
  const uint64_t per_chunk = kChunk / W;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_elems; i += (uint64_t)gridDim.x * blockDim.x) {
    const bool hi = i >= per_chunk;
    const uint32_t off = (uint32_t)((hi ? i - per_chunk : i) * W);
    if constexpr (W == 4) acc += __builtin_amdgcn_raw_buffer_load_b32(hi ? r1 : r0, (int)off, 0, 0);
    else if constexpr (W == 8) { const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(hi ? r1 : r0, (int)off, 0, 0); acc += v.x ^ v.y; }
    else { const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(hi ? r1 : r0, (int)off, 0, 0); acc += v.x ^ v.y ^ v.z ^ v.w; }
  }
  if (acc == 0x123456789ull) sink[0] = acc;
}

// n_lines random 64-byte lines (index list precomputed, distinct), 4 lanes x 16 B per line
__global__ __launch_bounds__(256) void cal_line64(const uint8_t* buf, const uint32_t* idx, uint32_t n_lines, uint64_t* sink) {
  const __amdgpu_buffer_rsrc_t r0 = rsrc(buf, kChunk);
  uint64_t acc = 0;
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
  for (uint32_t i = t; i < n_lines * 4u; i += nt) {
    const uint32_t line = idx[i >> 2];
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r0, (int)(line * 64u + (i & 3u) * 16u), 0, 0);
    acc += v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x123456789ull) sink[0] = acc;
}

// n_rows random 512-byte rows, one wavefront per row, 8 B/lane
__global__ __launch_bounds__(256) void cal_row512(const uint8_t* buf, const uint32_t* idx, uint32_t n_rows, uint64_t* sink) {
  const __amdgpu_buffer_rsrc_t r0 = rsrc(buf, kChunk);
  uint64_t acc = 0;
  const uint32_t lane = threadIdx.x & 63u, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = (gridDim.x * blockDim.x) >> 6;
  for (uint32_t i = wave; i < n_rows; i += nw) {
    const uint32_t row = __builtin_amdgcn_readfirstlane(idx[i]);
    const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(r0, (int)(lane * 8u), (int)(row * 512u), 0);
    acc += v.x ^ v.y;
  }
  if (acc == 0x123456789ull) sink[0] = acc;
}

template <int W>
__global__ __launch_bounds__(256) void cal_write(uint8_t* buf, uint64_t n_elems) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_elems; i += (uint64_t)gridDim.x * blockDim.x) {
    if constexpr (W == 4) ((uint32_t*)buf)[i] = (uint32_t)i;
    else ((uint64_t*)buf)[i] = i;
  }
}

int main() {
  uint8_t* buf; uint64_t* sink; uint32_t* idx;
  CK(hipMalloc(&buf, kBuf)); CK(hipMemset(buf, 1, kBuf));
  CK(hipMalloc(&sink, 64));
  const uint32_t n_lines = 4u << 20;            // 4 Mi distinct 64-byte lines out of the 16 Mi of chunk 0 = 256 MiB
  const uint32_t n_rows = 1u << 20;             // 1 Mi distinct 512-byte rows out of the 2 Mi of chunk 0 = 512 MiB
  std::vector<uint32_t> perm(16u << 20);
  for (uint32_t i = 0; i < perm.size(); ++i) perm[i] = i;
  std::mt19937 rng(7);
  for (uint32_t i = 0; i < n_lines; ++i) std::swap(perm[i], perm[i + rng() % (perm.size() - i)]);
  std::vector<uint32_t> rows(2u << 20);
  for (uint32_t i = 0; i < rows.size(); ++i) rows[i] = i;
  for (uint32_t i = 0; i < n_rows; ++i) std::swap(rows[i], rows[i + rng() % (rows.size() - i)]);
  CK(hipMalloc(&idx, (size_t)n_lines * 4));
  const dim3 grid(256 * 8), block(256);
  const uint64_t stream_bytes = 1ull << 30;     // each streaming kernel reads 1 GiB
  auto flush = [&]() { return hipMemset(buf + kChunk, 2, kChunk); };   // push chunk 0 out of the caches between kernels
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  auto timed = [&](const char* name, double bytes, auto&& launch) {
    (void)flush(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(a); launch(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
    printf("%-10s bytes=%.0f  %.1f us  %.2f TB/s\n", name, bytes, ms * 1e3, bytes / (ms * 1e-3) / 1e12);
  };
  timed("stream4", (double)stream_bytes, [&] { hipLaunchKernelGGL(cal_stream<4>, grid, block, 0, 0, buf, stream_bytes / 4, sink); });
  timed("stream8", (double)stream_bytes, [&] { hipLaunchKernelGGL(cal_stream<8>, grid, block, 0, 0, buf, stream_bytes / 8, sink); });
  timed("stream16", (double)stream_bytes, [&] { hipLaunchKernelGGL(cal_stream<16>, grid, block, 0, 0, buf, stream_bytes / 16, sink); });
  CK(hipMemcpy(idx, perm.data(), (size_t)n_lines * 4, hipMemcpyHostToDevice));
  timed("line64", (double)n_lines * 64.0, [&] { hipLaunchKernelGGL(cal_line64, grid, block, 0, 0, buf, idx, n_lines, sink); });
  CK(hipMemcpy(idx, rows.data(), (size_t)n_rows * 4, hipMemcpyHostToDevice));
  timed("row512", (double)n_rows * 512.0, [&] { hipLaunchKernelGGL(cal_row512, grid, block, 0, 0, buf, idx, n_rows, sink); });
  timed("write4", (double)(256u << 20), [&] { hipLaunchKernelGGL(cal_write<4>, grid, block, 0, 0, buf, (256ull << 20) / 4); });
  timed("write8", (double)(256u << 20), [&] { hipLaunchKernelGGL(cal_write<8>, grid, block, 0, 0, buf, (256ull << 20) / 8); });
  CK(hipDeviceSynchronize());
  return 0;
}
