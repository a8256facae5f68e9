// Measurement tool (not product code): the closed loop's index maintenance alone -- 64k x (16 known + 16 new) pairs per step into a
// 16 Mi-slot index, epoch tick + eviction every second step (bench.py --closed-loop's cadence) -- with the library's own kernels:
// update (budget + insert + list sort) and eviction times per step, standalone.  (Round 4, first run: the eviction alternately by the
// library kernel and by a bare scan over the same index state took the same 137-145 us -- the time is the data, 2 Mi victims among
// 16 Mi slots beyond the Infinity Cache, not the kernel: profiles/r04_micro_evictloop_round3_protocol.txt.)
//   evictloop [steps] [learn]     learn = 1: the update is told which pairs the pick kernel has vouched for (blocks 0..15, picked pod listed)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define EPPK_MAIN_UNIT 1
#include "../../gateway-api-inference-extension_amd/csrc/eppk_kernels.hip.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static uint64_t mix(uint64_t z) { z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// EXPERIMENT (round 6, not in the library): the eviction scan with lane = BUCKET, reading only the 24 bytes of a bucket that hold its flags and
// meta dwords (a key is present iff its tag is non-zero: the key words need not be read to find the victims).  mode 1 = with the victims'
// stores, mode 2 = the scan alone.  Listed sets of more than one pod are left to the library's kernel (none in this workload's victims).
__global__ void evict_meta_scan(uint64_t* keys, uint32_t slots, uint32_t epoch, uint32_t min_epoch, unsigned long long* ixc, int store) {
  const uint32_t nb = slots / eppk::kBucket, cur_tag = eppk::tag_of_epoch(epoch);
  const long long keep = (long long)epoch - (long long)min_epoch;
  uint32_t gone = 0;
  for (uint32_t b = blockIdx.x * blockDim.x + threadIdx.x; b < nb; b += gridDim.x * blockDim.x) {
    const u32x4* P = (const u32x4*)(keys + (size_t)b * eppk::kBucket);
    const u32x4 m0 = P[0];
    const uint2 m1 = *(const uint2*)(P + 1);
    const uint32_t meta[5] = {m0.y, m0.z, m0.w, m1.x, m1.y};
#pragma unroll
    for (uint32_t i = 0; i < 5u; ++i) {
      const uint32_t tag = meta[i] >> 24;
      if (tag != 0u && (long long)eppk::tag_age(cur_tag, tag) > keep) {
        ++gone;
        if (store) {
          keys[(size_t)b * eppk::kBucket + eppk::kKeySub0 + i] = eppk::kTomb;
          ((uint32_t*)(keys + (size_t)b * eppk::kBucket))[1u + i] = 0u;
        }
      }
    }
  }
  for (int off = 32; off >= 1; off >>= 1) gone += (uint32_t)__shfl_xor((int)gone, off);
  if ((threadIdx.x & 63u) == 0u && gone && store) {
    const uint32_t shard = (blockIdx.x & 63u) * 8u;
    atomicAdd(&ixc[shard + eppk::kIxLive], (unsigned long long)(0ull - (unsigned long long)gone));
    atomicAdd(&ixc[shard + eppk::kIxEvicted], (unsigned long long)gone);
  }
}

int main(int argc, char** argv) {
  using LW = uint64_t;
  const int evict_mode = argc > 3 ? atoi(argv[3]) : 0;
  const uint32_t api_slots = 16u << 20, slots = 2u * api_slots /* physical words: five of a bucket's eight hold keys (protocol v5) */, R = 65536, B = 32, P = 4096, stride = 8 + 8 * B;
  const int steps = argc > 1 ? atoi(argv[1]) : 16;
  const bool use_learn = argc > 2 && atoi(argv[2]) != 0;
  uint32_t* d_learn; CK(hipMalloc((void**)&d_learn, 65536 * 4));
  { std::vector<uint32_t> lw(65536, 0x80000000u | 16u); CK(hipMemcpy(d_learn, lw.data(), 65536 * 4, hipMemcpyHostToDevice)); }
  uint32_t lg = 0; while ((1u << lg) < slots / eppk::kBucket) ++lg;
  const uint32_t shift = 32u - lg, limit = api_slots / 2u;
  const size_t rows_bytes = (((size_t)slots + 3u) * 64u * sizeof(LW) + 255u) & ~(size_t)255u, index_bytes = rows_bytes + ((size_t)slots + 2u) * 8u;
  void* bitmaps; uint32_t *stamps, *lists, *status; unsigned long long* ixc;
  CK(hipMalloc(&bitmaps, index_bytes)); CK(hipMemset(bitmaps, 0, index_bytes));
  uint64_t* keys = (uint64_t*)((uint8_t*)bitmaps + rows_bytes);
  CK(hipMalloc((void**)&stamps, 2u * 4u)); CK(hipMemset(stamps, 0, 2u * 4u));          // (the reserved rows' exact stamps; every other stamp is a header tag)
  const size_t nd = ((size_t)slots + 4u) * eppk::kListDwords;
  const uint32_t sets_cap = slots / 16u;                      // the set table behind the lists (eppk.hip: eppk_create)
  CK(hipMalloc((void**)&lists, (nd + (size_t)sets_cap * eppk::kListDwords) * 4u));
  hipLaunchKernelGGL(eppk::lists_fill_kernel, dim3(1024), dim3(256), 0, 0, lists, nd);
  CK(hipMemset(lists + nd, 0, (size_t)sets_cap * eppk::kListDwords * 4u));
  uint32_t* set_ctl; CK(hipMalloc((void**)&set_ctl, 8)); CK(hipMemset(set_ctl, 0, 8));
  const eppk::SetTab settab{lists + nd, sets_cap - 1u, set_ctl};
  CK(hipMalloc((void**)&ixc, 64 * 64)); CK(hipMemset(ixc, 0, 64 * 64));
  CK(hipMalloc((void**)&status, 8)); CK(hipMemset(status, 0, 8));
  eppk::IxLaunch* d_ixl; CK(hipMalloc((void**)&d_ixl, sizeof(eppk::IxLaunch)));
  uint32_t* d_wl; const uint32_t wl_cap = R * B; CK(hipMalloc((void**)&d_wl, (4u + (size_t)wl_cap) * 4u)); CK(hipMemset(d_wl, 0, 16));
  uint8_t* d_rows[4]; int32_t* d_picks[4];
  std::vector<uint8_t> rows((size_t)R * stride); std::vector<int32_t> picks(R);
  for (int b = 0; b < 4; ++b) { CK(hipMalloc((void**)&d_rows[b], (size_t)R * stride)); CK(hipMalloc((void**)&d_picks[b], R * 4)); }
  hipEvent_t e0, e1, e2, e3; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2)); CK(hipEventCreate(&e3));
  uint32_t epoch = 1, sort_uses = 0;
  for (int g = 0; g < steps; ++g) {
    const int b = g & 3;
    for (uint32_t r = 0; r < R; ++r) {           // a fresh batch: 16 hot blocks of one of 256 prefixes + 16 blocks nobody has seen
      uint8_t* row = rows.data() + (size_t)r * stride;
      const uint32_t grp = (uint32_t)(mix(r * 7919ull + 1 + g) % 256u);
      ((int32_t*)row)[0] = -1; ((uint32_t*)row)[1] = 32u;
      uint64_t* h = (uint64_t*)(row + 8);
      for (uint32_t i = 0; i < 16; ++i) h[i] = mix(0xABCD0000ull + grp * 16u + i) | 2ull;
      for (uint32_t i = 16; i < 32; ++i) h[i] = mix(((uint64_t)(g + 1) << 40) + (uint64_t)r * 32u + i) | 2ull;
      picks[r] = (int32_t)((grp * 8u + (r & 7u)) % P);
    }
    CK(hipMemcpy(d_rows[b], rows.data(), rows.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(d_picks[b], picks.data(), R * 4, hipMemcpyHostToDevice));
    const uint64_t total = (uint64_t)R * B;
    const eppk::SortWl sw{d_wl, wl_cap, sort_uses & 1u}; ++sort_uses;
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(eppk::index_budget_kernel, dim3(1), dim3(64), 0, 0, ixc, limit, slots, (unsigned long long)total, d_ixl);
    hipLaunchKernelGGL((eppk::index_insert_picks_kernel<LW>), dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, 0, keys, bitmaps, lists, stamps, slots, shift, limit,
                       epoch, ixc, d_rows[b], stride, B, d_picks[b], R, P, status, (const LW*)nullptr, sw, (const eppk::IxLaunch*)d_ixl,
                       (const uint32_t*)(use_learn && g >= 4 ? d_learn : nullptr));       // (the hot prefixes' pods are listed after a few steps)
    hipLaunchKernelGGL(eppk::index_canon_kernel, dim3(64), dim3(256), 0, 0, keys, lists, slots, settab, sw.wl, sw.cap, sw.which, 0u, (uint32_t*)nullptr);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("step %2d epoch %u: update %7.1f us", g, epoch, ms * 1e3);
    if (g & 1) {
      ++epoch;
      if (epoch > 2) {
        CK(hipEventRecord(e2));
        if (evict_mode == 2) {      // the two scans alone (no victims: min_epoch 0 keeps everything), one after the other
          hipLaunchKernelGGL((eppk::index_evict_kernel<LW>), dim3(4096), dim3(256), 0, 0, keys, bitmaps, lists, (const uint32_t*)stamps, slots, epoch, 0u, ixc);
          CK(hipEventRecord(e3)); CK(hipEventSynchronize(e3)); CK(hipEventElapsedTime(&ms, e2, e3)); printf("   [library scan alone %6.1f us]", ms * 1e3);
          CK(hipEventRecord(e2));
          hipLaunchKernelGGL(evict_meta_scan, dim3(4096), dim3(256), 0, 0, keys, slots, epoch, 0u, ixc, 0);
          CK(hipEventRecord(e3)); CK(hipEventSynchronize(e3)); CK(hipEventElapsedTime(&ms, e2, e3)); printf(" [meta scan alone %6.1f us]", ms * 1e3);
          CK(hipEventRecord(e2));
        }
        if (evict_mode == 1) hipLaunchKernelGGL(evict_meta_scan, dim3(4096), dim3(256), 0, 0, keys, slots, epoch, epoch - 1u, ixc, 1);
        else hipLaunchKernelGGL((eppk::index_evict_kernel<LW>), dim3(4096), dim3(256), 0, 0, keys, bitmaps, lists, (const uint32_t*)stamps, slots, epoch, epoch - 1u, ixc);
        CK(hipEventRecord(e3)); CK(hipEventSynchronize(e3));
        CK(hipEventElapsedTime(&ms, e2, e3));
        unsigned long long h[64 * 8]; CK(hipMemcpy(h, ixc, sizeof h, hipMemcpyDeviceToHost));
        long long live = 0; for (int s2 = 0; s2 < 64; ++s2) live += (long long)h[s2 * 8 + eppk::kIxLive];
        printf("   evict < %u %7.1f us   live after %lld", epoch - 1u, ms * 1e3, live);
      }
    }
    printf("\n");
  }
  return 0;
}
