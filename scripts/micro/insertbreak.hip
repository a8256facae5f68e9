// Measurement tool (not product code): where index_insert_picks_kernel's time goes.  The library's own kernel (eppk_kernels.hip.h,
// included as is) on an 8 Mi-slot index with the shapes of a C5 closed-loop step, split into its two populations:
//   new      65536 requests x 16 blocks nobody has seen      (1 Mi new keys: bucket CAS + list CAS + stamp store; the dense row is not touched)
//   known    65536 requests x 16 blocks of 256 hot prefixes, the (hash, pod) pair already present   (1 Mi look-ups on 4096 keys)
//   known+   the same with the index epoch advanced (the first touch of a key refreshes its stamp)
//   step     65536 requests x (16 known + 16 new): what the closed loop runs
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#define EPPK_MAIN_UNIT 1
#include "../../gateway-api-inference-extension_amd/csrc/eppk_kernels.hip.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

static uint64_t mix(uint64_t z) { z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }

// order-independent digest of the index: per live key its pod count, stamp, list count and list ids
__global__ void digest_kernel(const uint64_t* keys, const uint64_t* rows, const uint32_t* lists, uint32_t slots, unsigned long long* out) {
  unsigned long long acc = 0;
  for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < slots; s += gridDim.x * blockDim.x) {
    if (!eppk::is_key_word(s)) continue;
    const uint64_t k = keys[s];
    if (k == 0ull || k == eppk::kTomb) continue;
    unsigned long long pc = 0, ids = 0;
    const uint32_t* L = lists + (size_t)s * eppk::kListDwords;
    const uint32_t cnt = L[3];
    if (cnt > eppk::kListCap) { for (uint32_t i = 0; i < 64; ++i) pc += __popcll(rows[(size_t)s * 64u + i]); } else pc = cnt;   // (a listed set's row is all-zero)
    for (uint32_t q = 0; q < (cnt < eppk::kListCap ? cnt : eppk::kListCap); ++q) { const unsigned long long id = ((const uint16_t*)L)[eppk::list_pos(q)]; ids += (id + 1) * (id + 1); }
    unsigned long long z = k + 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z ^= z >> 27;
    const unsigned long long tag = ((const uint32_t*)keys)[eppk::meta_dword(s)] >> 24;       // the stamp: the tag in the key's meta dword
    acc += z * (1ull + pc + 7ull * tag + 13ull * cnt + 31ull * ids);
  }
  atomicAdd(out, acc);
}

int main(int argc, char** argv) {
  using LW = uint64_t;
  const uint32_t api_slots = 8u << 20, slots = 2u * api_slots /* physical words: five of a bucket's eight hold keys (protocol v5) */, R = 65536, B = 32, P = 4096, stride = 8 + 8 * B;
  uint32_t lg = 0; while ((1u << lg) < slots / eppk::kBucket) ++lg;
  const uint32_t shift = 32u - lg, limit = api_slots / 2u;
  const size_t rows_bytes = (((size_t)slots + 3u) * 64u * sizeof(LW) + 255u) & ~(size_t)255u, index_bytes = rows_bytes + ((size_t)slots + 2u) * 8u;
  void* bitmaps; uint32_t *stamps, *lists, *status; unsigned long long* ixc;
  CK(hipMalloc(&bitmaps, index_bytes)); CK(hipMemset(bitmaps, 0, index_bytes));
  uint64_t* keys = (uint64_t*)((uint8_t*)bitmaps + rows_bytes);
  CK(hipMalloc((void**)&stamps, 2u * 4u)); CK(hipMemset(stamps, 0, 2u * 4u));          // (exact stamps of the two reserved rows only)
  const size_t nd = ((size_t)slots + 4u) * eppk::kListDwords;
  const uint32_t sets_cap = slots / 16u;                      // the set table behind the lists (eppk.hip: eppk_create)
  CK(hipMalloc((void**)&lists, (nd + (size_t)sets_cap * eppk::kListDwords) * 4u));
  hipLaunchKernelGGL(eppk::lists_fill_kernel, dim3(1024), dim3(256), 0, 0, lists, nd);
  CK(hipMemset(lists + nd, 0, (size_t)sets_cap * eppk::kListDwords * 4u));
  uint32_t* set_ctl; CK(hipMalloc((void**)&set_ctl, 8)); CK(hipMemset(set_ctl, 0, 8));
  const eppk::SetTab settab{lists + nd, sets_cap - 1u, set_ctl};
  CK(hipMalloc((void**)&ixc, 256 * 64)); CK(hipMemset(ixc, 0, 256 * 64));   // (room for the 256-shard variant)
  CK(hipMalloc((void**)&status, 8)); CK(hipMemset(status, 0, 8));

  // batches (host): kind 0 = new, 1 = known, 2 = step; `gen` makes the unique hashes of different batches differ
  auto make = [&](int kind, uint64_t gen, std::vector<uint8_t>& rows, std::vector<int32_t>& picks) {
    rows.assign((size_t)R * stride, 0); picks.resize(R);
    for (uint32_t r = 0; r < R; ++r) {
      uint8_t* row = rows.data() + (size_t)r * stride;
      const uint32_t g = (uint32_t)(mix(r * 7919ull + 1) % 256u), nb = kind == 2 ? 32u : 16u;
      ((int32_t*)row)[0] = -1; ((uint32_t*)row)[1] = nb;
      uint64_t* h = (uint64_t*)(row + 8);
      uint32_t i = 0;
      if (kind != 0) for (; i < 16; ++i) h[i] = mix(0xABCD0000ull + g * 16u + i) | 2ull;
      for (; i < nb; ++i) h[i] = mix((gen << 40) + (uint64_t)r * 32u + i) | 2ull;
      picks[r] = (int32_t)((g * 8u + (r & 7u)) % P);      // one of the group's eight pods
    }
  };
  uint8_t* d_rows; int32_t* d_picks;
  CK(hipMalloc((void**)&d_rows, (size_t)R * stride)); CK(hipMalloc((void**)&d_picks, R * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  using Kern = void (*)(uint64_t*, void*, uint32_t*, uint32_t*, uint32_t, uint32_t, uint32_t, uint32_t, unsigned long long*, const uint8_t*, uint32_t, uint32_t, const int32_t*, uint32_t, uint32_t, uint32_t*, const LW*, eppk::SortWl, const eppk::IxLaunch*, const uint32_t*);
  eppk::IxLaunch* d_ixl; CK(hipMalloc((void**)&d_ixl, sizeof(eppk::IxLaunch)));
  uint32_t* d_wl; const uint32_t wl_cap = R * B; CK(hipMalloc((void**)&d_wl, (4u + (size_t)wl_cap) * 4u)); CK(hipMemset(d_wl, 0, 16));
  uint32_t sort_uses = 0;
  hipEvent_t e2; CK(hipEventCreate(&e2));
  Kern kern = eppk::index_insert_picks_kernel<LW>;
  auto run = [&](const char* what, int kind, uint64_t gen, uint32_t epoch, bool print) -> int {
    std::vector<uint8_t> rows; std::vector<int32_t> picks;
    make(kind, gen, rows, picks);
    CK(hipMemcpy(d_rows, rows.data(), rows.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(d_picks, picks.data(), R * 4, hipMemcpyHostToDevice));
    const uint64_t total = (uint64_t)R * B;
    CK(hipEventRecord(e0));
    const eppk::SortWl sw{d_wl, wl_cap, sort_uses & 1u};
    ++sort_uses;
    hipLaunchKernelGGL(eppk::index_budget_kernel, dim3(1), dim3(64), 0, 0, ixc, limit, slots, (unsigned long long)total, d_ixl);
    hipLaunchKernelGGL(kern, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, 0, keys, bitmaps, lists, stamps, slots, shift, limit,
                       epoch, ixc, d_rows, stride, B, d_picks, R, P, status, (const LW*)nullptr, sw, (const eppk::IxLaunch*)d_ixl, (const uint32_t*)nullptr);
    CK(hipEventRecord(e1));
    hipLaunchKernelGGL(eppk::index_canon_kernel, dim3(64), dim3(256), 0, 0, keys, lists, slots, settab, sw.wl, sw.cap, sw.which, 0u, (uint32_t*)nullptr);
    CK(hipEventRecord(e2)); CK(hipEventSynchronize(e2));
    float ms, ms2; CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipEventElapsedTime(&ms2, e1, e2));
    if (print) printf("%-8s epoch %u: %8.1f us  (+ sort pass %5.1f us)\n", what, epoch, ms * 1e3, ms2 * 1e3);
    return 0;
  };
  unsigned long long* d_dig; CK(hipMalloc((void**)&d_dig, 8));
  using EvictKern = void (*)(uint64_t*, void*, uint32_t*, const uint32_t*, uint32_t, uint32_t, uint32_t, unsigned long long*);
  struct V { const char* name; Kern k; EvictKern ev; uint32_t ev_grid = 4096; };
  std::vector<V> variants{{"library", eppk::index_insert_picks_kernel<LW>, eppk::index_evict_kernel<LW>}};
  for (const V& v : variants) {
    if (argc > 1) { bool want = false; for (int a = 1; a < argc; ++a) want = want || std::string(v.name).rfind(argv[a], 0) == 0; if (!want) continue; }
    printf("--- %s\n", v.name);
    kern = v.k;
    CK(hipMemset(bitmaps, 0, index_bytes)); CK(hipMemset(stamps, 0, 2u * 4u)); CK(hipMemset(ixc, 0, 256 * 64));
    hipLaunchKernelGGL(eppk::lists_fill_kernel, dim3(1024), dim3(256), 0, 0, lists, nd);
    if (run("warm", 1, 0, 2, false)) return 1;                 // the hot prefixes enter the index
    if (run("new", 0, 10, 2, true)) return 1;
    if (run("known", 1, 0, 2, true)) return 1;
    if (run("known+", 1, 0, 3, true)) return 1;
    if (run("step", 2, 20, 3, true)) return 1;               // (2.1 Mi + 2 Mi pairs stays below the 4 Mi limit: no exact-capacity mode)
    unsigned long long h[256 * 8]; CK(hipMemcpy(h, ixc, sizeof h, hipMemcpyDeviceToHost));
    unsigned long long live = 0, dropped = 0, lost = 0; for (uint32_t s2 = 0; s2 < 256u; ++s2) { live += h[s2 * 8 + eppk::kIxLive]; dropped += h[s2 * 8 + eppk::kIxDropped]; lost += h[s2 * 8 + eppk::kIxEvicted]; }
    CK(hipMemset(d_dig, 0, 8));
    hipLaunchKernelGGL(digest_kernel, dim3(4096), dim3(256), 0, 0, keys, (const uint64_t*)bitmaps, lists, slots, d_dig);
    unsigned long long dig; CK(hipMemcpy(&dig, d_dig, 8, hipMemcpyDeviceToHost));
    printf("live keys %llu (expected %u), dropped %llu, lost claims %llu, digest %016llx\n", live, 4096u + 2u * 1048576u, dropped, lost, dig);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(v.ev, dim3(v.ev_grid), dim3(256), 0, 0, keys, bitmaps, lists, (const uint32_t*)stamps, slots, 3u, 3u, ixc);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ems; CK(hipEventElapsedTime(&ems, e0, e1));
    CK(hipMemcpy(h, ixc, sizeof h, hipMemcpyDeviceToHost));
    unsigned long long live2 = 0, ev = 0; for (uint32_t s2 = 0; s2 < 256u; ++s2) { live2 += h[s2 * 8 + eppk::kIxLive]; ev += h[s2 * 8 + eppk::kIxEvicted]; }
    CK(hipMemset(d_dig, 0, 8));
    hipLaunchKernelGGL(digest_kernel, dim3(4096), dim3(256), 0, 0, keys, (const uint64_t*)bitmaps, lists, slots, d_dig);
    CK(hipMemcpy(&dig, d_dig, 8, hipMemcpyDeviceToHost));
    printf("evict    < epoch 3: %8.1f us   live keys after %llu, evicted %llu, digest %016llx\n", ems * 1e3, live2, ev - lost, dig);
  }
  return 0;
}
