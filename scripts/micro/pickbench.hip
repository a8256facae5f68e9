// Measurement tool (not product code): bench.py's timed region without Python or torch, for A/B runs that cost seconds of GPU budget.
//   pickbench <workload dir> <libeppk.so> [more libeppk.so ...] [--steps N] [--inflight 1|2] [--closed-loop] [--cl-slots N] [--profile]
// Every library named is dlopen()ed in turn (scripts/abq.sh leaves one per variant under ab/<name>/), run on the same workload
// (scripts/dump_workload.py), timed like bench.py (16 rotating batches resident in HBM, two in flight on two streams; closed loop:
// pick -> eppk_index_insert_picks_device -> next batch with fresh tail hashes, ageing every 2 steps, keep 2 epochs) and compared
// with the first one through a digest of the picks of the first 16 steps (open loop) -- a variant that changes a pick is flagged.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/eppk.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static std::vector<uint8_t> slurp(const std::string& p) {
  std::vector<uint8_t> v;
  FILE* f = std::fopen(p.c_str(), "rb");
  if (!f) { std::printf("cannot read %s\n", p.c_str()); std::exit(2); }
  std::fseek(f, 0, SEEK_END);
  v.resize((size_t)std::ftell(f));
  std::fseek(f, 0, SEEK_SET);
  if (std::fread(v.data(), 1, v.size(), f) != v.size()) std::exit(2);
  std::fclose(f);
  return v;
}

// batch b > 0 = batch 0 with its rows rotated by b * 4099 (other addresses, other order, the same population) and, in closed-loop
// mode, the second half of every row's hashes replaced by hashes nobody has seen (gen)
__global__ void derive_batch(const uint8_t* src, uint8_t* dst, uint32_t R, uint32_t stride, uint32_t rot, uint64_t gen) {
  const uint32_t words = stride / 8u;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)R * words; i += (size_t)gridDim.x * blockDim.x) {
    const uint32_t r = (uint32_t)(i / words), w = (uint32_t)(i % words);
    uint64_t v = ((const uint64_t*)src)[(size_t)((r + rot) % R) * words + w];
    if (gen && w >= 1u) {
      const uint32_t nb = (uint32_t)(((const uint64_t*)src)[(size_t)((r + rot) % R) * words] >> 32);
      const uint32_t blk = w - 1u;
      if (blk >= nb / 2u && blk < nb) {
        uint64_t z = (gen << 40) + (uint64_t)r * 64u + blk + 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; v = (z ^ (z >> 31)) | 2ull;
      }
    }
    ((uint64_t*)dst)[i] = v;
  }
}

struct Api {
  void* h = nullptr;
  decltype(&eppk_create) create; decltype(&eppk_destroy) destroy; decltype(&eppk_last_error) last_error;
  decltype(&eppk_snapshot_publish) publish; decltype(&eppk_index_insert) index_insert; decltype(&eppk_pick_batch_device) pick_device;
  decltype(&eppk_index_insert_picks_device) insert_picks; decltype(&eppk_index_advance_epoch) advance; decltype(&eppk_index_evict_older_device) evict_device;
  decltype(&eppk_quad_stats) quad_stats; decltype(&eppk_profile_enable) profile_enable; decltype(&eppk_index_size) index_size; decltype(&eppk_index_dropped) index_dropped;
  bool load(const char* path) {
    h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!h) { std::printf("dlopen %s: %s\n", path, dlerror()); return false; }
#define SYM(field, name) field = (decltype(field))dlsym(h, #name); if (!field) { std::printf("%s lacks %s\n", path, #name); return false; }
    SYM(create, eppk_create) SYM(destroy, eppk_destroy) SYM(last_error, eppk_last_error) SYM(publish, eppk_snapshot_publish)
    SYM(index_insert, eppk_index_insert) SYM(pick_device, eppk_pick_batch_device) SYM(insert_picks, eppk_index_insert_picks_device)
    SYM(advance, eppk_index_advance_epoch) SYM(evict_device, eppk_index_evict_older_device) SYM(index_size, eppk_index_size) SYM(index_dropped, eppk_index_dropped)
#undef SYM
    quad_stats = (decltype(quad_stats))dlsym(h, "eppk_quad_stats");       // (optional: older builds)
    profile_enable = (decltype(profile_enable))dlsym(h, "eppk_profile_enable");
    return true;
  }
};

int main(int argc, char** argv) {
  std::vector<std::string> libs;
  std::string dir;
  int steps = 200, warmup = 20, inflight = 2;
  bool closed = false, profile = false;
  uint32_t cl_slots = 1u << 24;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    if (a == "--steps") steps = std::atoi(argv[++i]);
    else if (a == "--warmup") warmup = std::atoi(argv[++i]);
    else if (a == "--inflight") inflight = std::atoi(argv[++i]);
    else if (a == "--closed-loop") closed = true;
    else if (a == "--profile") profile = true;          // completion events on every pick launch, as bench.py runs (eppk_profile_enable)
    else if (a == "--cl-slots") cl_slots = (uint32_t)std::strtoul(argv[++i], nullptr, 0);
    else if (dir.empty()) dir = a;
    else libs.push_back(a);
  }
  if (dir.empty() || libs.empty()) { std::printf("usage: pickbench <workload dir> <libeppk.so>... [--steps N] [--inflight 1|2] [--closed-loop]\n"); return 2; }
  if (closed) inflight = 1;
  uint32_t R, P, B, n_index, slots, n_sc;
  eppk_cfg cfg{};
  {
    FILE* f = std::fopen((dir + "/meta.txt").c_str(), "r");
    if (!f || std::fscanf(f, "%u %u %u %u %u %u", &R, &P, &B, &n_index, &slots, &n_sc) != 6) { std::printf("bad meta.txt\n"); return 2; }
    for (uint32_t i = 0; i < n_sc; ++i) if (std::fscanf(f, "%u %d", &cfg.chain[i].kind, &cfg.chain[i].weight) != 2) return 2;
    std::fclose(f);
  }
  const std::vector<uint8_t> pods = slurp(dir + "/pods.bin"), ih = slurp(dir + "/index_hashes.bin"), ip = slurp(dir + "/index_pods.bin"), reqs = slurp(dir + "/reqs.bin");
  const uint32_t stride = 8u + 8u * B, NB = 16;
  if (reqs.size() != (size_t)R * stride || pods.size() != (size_t)P * 64u) { std::printf("sizes do not match meta.txt\n"); return 2; }
  cfg.struct_size = sizeof cfg; cfg.device = 0; cfg.max_pods = P; cfg.max_blocks = B; cfg.max_batch = 4096; cfg.index_slots = closed ? cl_slots : slots; cfg.n_scorers = n_sc;

  uint8_t* d_batch[NB];
  for (uint32_t b = 0; b < NB; ++b) CK(hipMalloc((void**)&d_batch[b], reqs.size()));
  CK(hipMemcpy(d_batch[0], reqs.data(), reqs.size(), hipMemcpyHostToDevice));
  for (uint32_t b = 1; b < NB; ++b) hipLaunchKernelGGL(derive_batch, dim3(2048), dim3(256), 0, 0, d_batch[0], d_batch[b], R, stride, b * 4099u, 0ull);
  uint8_t* d_scratch; CK(hipMalloc((void**)&d_scratch, reqs.size()));
  int32_t* d_picks[2]; double* d_scores[2];
  for (int s = 0; s < 2; ++s) { CK(hipMalloc((void**)&d_picks[s], R * 4)); CK(hipMalloc((void**)&d_scores[s], R * 8)); }
  hipStream_t st[2]; for (int s = 0; s < 2; ++s) CK(hipStreamCreateWithFlags(&st[s], hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipDeviceSynchronize());

  std::vector<unsigned long long> ref_digest;
  for (size_t li = 0; li < libs.size(); ++li) {
    Api api;
    if (!api.load(libs[li].c_str())) return 1;
    eppk_ctx* ctx = nullptr;
    if (api.create(&cfg, &ctx) != EPPK_OK) { std::printf("eppk_create: %s\n", api.last_error(nullptr)); return 1; }
    if (api.publish(ctx, (const eppk_pod_row*)pods.data(), P, 1) != EPPK_OK || api.index_insert(ctx, (const uint64_t*)ih.data(), (const uint32_t*)ip.data(), n_index) != EPPK_OK) {
      std::printf("publish / index_insert: %s\n", api.last_error(ctx)); return 1;
    }
    // digest of the first NB steps' picks (open loop: a function of the inputs alone)
    std::vector<unsigned long long> digest;
    std::vector<int32_t> hp(R);
    if (!closed) for (uint32_t b = 0; b < NB; ++b) {
      if (api.pick_device(ctx, d_batch[b], R, nullptr, d_picks[0], d_scores[0], st[0]) != EPPK_OK) { std::printf("pick: %s\n", api.last_error(ctx)); return 1; }
      CK(hipStreamSynchronize(st[0]));
      CK(hipMemcpy(hp.data(), d_picks[0], R * 4, hipMemcpyDeviceToHost));
      unsigned long long d = 0;
      for (uint32_t r = 0; r < R; ++r) d += (unsigned long long)(uint32_t)hp[r] * (0x9E3779B97F4A7C15ull * (r + 1u) | 1ull);
      digest.push_back(d);
    }
    if (profile && api.profile_enable) api.profile_enable(ctx, 1);
    uint64_t gen = 1;
    uint32_t epoch = 1;
    auto step = [&](int i) -> int {
      const int s = i % inflight;
      const uint8_t* rows = d_batch[i % NB];
      if (closed) { hipLaunchKernelGGL(derive_batch, dim3(2048), dim3(256), 0, st[s], d_batch[i % NB], d_scratch, R, stride, 0u, gen++); rows = d_scratch; }
      if (api.pick_device(ctx, rows, R, nullptr, d_picks[s], d_scores[s], st[s]) != EPPK_OK) return 1;
      if (closed) {
        if (api.insert_picks(ctx, rows, d_picks[s], R, st[s]) != EPPK_OK) return 1;
        if (i % 2 == 1) {
          CK(hipStreamSynchronize(st[s]));                      // (eppk_index_advance_epoch is a host call: bench.py does the same)
          if (api.advance(ctx, &epoch) != EPPK_OK) return 1;
          if (epoch > 2u && api.evict_device(ctx, epoch - 2u, st[s]) != EPPK_OK) return 1;
        }
      }
      return 0;
    };
    for (int i = 0; i < warmup; ++i) if (step(i)) { std::printf("step: %s\n", api.last_error(ctx)); return 1; }
    for (int s = 0; s < 2; ++s) CK(hipStreamSynchronize(st[s]));
    CK(hipEventRecord(e0, st[0]));
    if (inflight == 2) CK(hipStreamWaitEvent(st[1], e0, 0));
    for (int i = 0; i < steps; ++i) if (step(warmup + i)) { std::printf("step: %s\n", api.last_error(ctx)); return 1; }
    if (inflight == 2) { hipEvent_t j; CK(hipEventCreate(&j)); CK(hipEventRecord(j, st[1])); CK(hipStreamWaitEvent(st[0], j, 0)); }
    CK(hipEventRecord(e1, st[0]));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    uint64_t ql = 0, qd = 0, dropped = 0; uint32_t size = 0;
    if (api.quad_stats) api.quad_stats(ctx, &ql, &qd);
    api.index_size(ctx, &size); api.index_dropped(ctx, &dropped);
    bool same = true;
    if (li == 0) ref_digest = digest; else same = digest == ref_digest;
    std::printf("%-40s %s inflight %d: %8.2f us per step, %7.3f G decisions/s   quad launches %llu deferred %llu  index %u keys, %llu dropped  %s\n", libs[li].c_str(),
                closed ? "closed loop" : "open loop", inflight, ms * 1e3 / steps, (double)R * steps / (ms * 1e-3) / 1e9, (unsigned long long)ql, (unsigned long long)qd, size,
                (unsigned long long)dropped, closed ? "" : (same ? "picks == first library" : "PICKS DIFFER FROM THE FIRST LIBRARY"));
    api.destroy(ctx);
  }
  return 0;
}
