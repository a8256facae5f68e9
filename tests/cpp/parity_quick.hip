// A parity check that costs seconds of GPU time (test infrastructure: links liboracle; driven by tests/test_gpu_parity_quick.py):
// the C5 workload (dumped by scripts/dump_workload.py) through the library's device entry points, against the oracle, bit for bit --
//   1. the open loop: batch 0, picks and binary64 scores;
//   2. the closed loop: G generations of  pick -> index_insert_picks -> (every 2nd: advance epoch, evict older than epoch - 1)  with
//      fresh tail hashes per generation; picks, scores and the number of live hashes after every generation.
// What the full pytest suite checks in minutes, for the two things a kernel edit breaks first.   parity_quick <workload dir> [G]
#ifndef PARITY_QUICK_SELFTEST
#include <hip/hip_runtime.h>
#endif

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/eppk.h"
#include "../../oracle/oracle.h"

#ifdef PARITY_QUICK_SELFTEST
// Self-test of THIS harness on a box without a GPU (tests/test_zz_parity_quick_gpu.py::test_parity_quick_selftest): the library's entry
// points and the HIP calls are replaced by stand-ins backed by a second oracle instance, so that the control flow -- file parsing,
// fresh tails, epochs, the eviction, the index-size bookkeeping -- is exercised end to end.  It says nothing about the library.
#include <map>
typedef int hipError_t;
enum { hipSuccess = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2 };
static const char* hipGetErrorString(hipError_t) { return "stand-in"; }
static hipError_t hipMalloc(void** p, size_t n) { *p = std::malloc(n); return *p ? hipSuccess : 1; }
static hipError_t hipMemcpy(void* d, const void* s, size_t n, int) { std::memcpy(d, s, n); return hipSuccess; }
struct eppk_ctx { eppk_cfg cfg; std::vector<eppk_pod_row> pods; orc_index* ix; };
extern "C" {
int eppk_create(const eppk_cfg* cfg, eppk_ctx** out) { *out = new eppk_ctx{*cfg, {}, orc_index_new()}; return EPPK_OK; }
void eppk_destroy(eppk_ctx* c) { orc_index_free(c->ix); delete c; }
const char* eppk_last_error(const eppk_ctx*) { return "stand-in"; }
int eppk_snapshot_publish(eppk_ctx* c, const eppk_pod_row* rows, uint32_t n, uint64_t) { c->pods.assign(rows, rows + n); return EPPK_OK; }
int eppk_index_insert(eppk_ctx* c, const uint64_t* h, const uint32_t* p, uint32_t n) { for (uint32_t i = 0; i < n; ++i) orc_index_insert(c->ix, h[i], p[i]); return EPPK_OK; }
int eppk_pick_batch_device(eppk_ctx* c, const void* reqs, uint32_t n, const uint64_t* mask, int32_t* picks, double* scores, void*) {
  return orc_pick_batch_mt(c->cfg.chain, c->cfg.n_scorers, c->pods.data(), (uint32_t)c->pods.size(), c->ix, reqs, c->cfg.max_blocks, n, mask, picks, scores, 4) ? EPPK_ERR_ARG : EPPK_OK;
}
int eppk_index_insert_picks_device(eppk_ctx* c, const void* reqs, const int32_t* picks, uint32_t n, void*) { orc_index_insert_picks(c->ix, reqs, c->cfg.max_blocks, n, picks); return EPPK_OK; }
int eppk_launch_status(eppk_ctx*, uint32_t* flags) { *flags = 0; return EPPK_OK; }
int eppk_index_advance_epoch(eppk_ctx* c, uint32_t* e) { *e = orc_index_advance_epoch(c->ix); return EPPK_OK; }
int eppk_index_evict_older(eppk_ctx* c, uint32_t min_epoch, uint32_t* n) { *n = orc_index_evict_older(c->ix, min_epoch); return EPPK_OK; }
int eppk_index_size(eppk_ctx* c, uint32_t* n) { *n = (uint32_t)orc_index_size(c->ix); return EPPK_OK; }
int eppk_index_dropped(eppk_ctx*, uint64_t* n) { *n = 0; return EPPK_OK; }
}
#endif

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define EK(x) do { int rc_ = (x); if (rc_ != EPPK_OK) { std::printf("%s -> %d: %s\n", #x, rc_, eppk_last_error(ctx)); return 1; } } while (0)

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
static uint64_t mix(uint64_t z) { z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }

int main(int argc, char** argv) {
  if (argc < 2) { std::printf("usage: parity_quick <workload dir> [generations]\n"); return 2; }
  const std::string dir = argv[1];
  const int G = argc > 2 ? std::atoi(argv[2]) : 4;
  uint32_t R, P, B, n_index, slots, n_sc;
  eppk_cfg cfg{};
  {
    FILE* f = std::fopen((dir + "/meta.txt").c_str(), "r");
    if (!f || std::fscanf(f, "%u %u %u %u %u %u", &R, &P, &B, &n_index, &slots, &n_sc) != 6) { std::printf("bad meta.txt\n"); return 2; }
    for (uint32_t i = 0; i < n_sc; ++i) if (std::fscanf(f, "%u %d", &cfg.chain[i].kind, &cfg.chain[i].weight) != 2) return 2;
    std::fclose(f);
  }
  const std::vector<uint8_t> pods = slurp(dir + "/pods.bin"), ih = slurp(dir + "/index_hashes.bin"), ip = slurp(dir + "/index_pods.bin");
  std::vector<uint8_t> reqs = slurp(dir + "/reqs.bin");
  const uint32_t stride = 8u + 8u * B;
  const int threads = (int)std::thread::hardware_concurrency();
  cfg.struct_size = sizeof cfg; cfg.device = 0; cfg.max_pods = P; cfg.max_blocks = B; cfg.max_batch = 4096; cfg.n_scorers = n_sc;

  uint8_t* d_reqs; int32_t* d_picks; double* d_scores;
  CK(hipMalloc((void**)&d_reqs, reqs.size())); CK(hipMalloc((void**)&d_picks, R * 4)); CK(hipMalloc((void**)&d_scores, R * 8));
  std::vector<int32_t> gp(R), op(R);
  std::vector<double> gs(R), os(R);
  int bad = 0;
  auto compare = [&](const char* what) {
    size_t dp = 0, ds = 0;
    for (uint32_t r = 0; r < R; ++r) { dp += gp[r] != op[r]; ds += std::memcmp(&gs[r], &os[r], 8) != 0; }
    std::printf("%-28s picks differ: %zu, scores differ (bitwise): %zu of %u\n", what, dp, ds, R);
    bad += dp != 0 || ds != 0;
  };

  for (int pass = 0; pass < 2; ++pass) {           // pass 0: open loop on the workload's own index; pass 1: closed loop on a large one
    cfg.index_slots = pass == 0 ? slots : (1u << 24);     // (the stand-ins of the self-test ignore it)
    eppk_ctx* ctx = nullptr;
    if (eppk_create(&cfg, &ctx) != EPPK_OK) { std::printf("eppk_create: %s\n", eppk_last_error(nullptr)); return 1; }
    EK(eppk_snapshot_publish(ctx, (const eppk_pod_row*)pods.data(), P, 1));
    EK(eppk_index_insert(ctx, (const uint64_t*)ih.data(), (const uint32_t*)ip.data(), n_index));
    orc_index* oix = orc_index_new();
    for (uint32_t i = 0; i < n_index; ++i) orc_index_insert(oix, ((const uint64_t*)ih.data())[i], ((const uint32_t*)ip.data())[i]);
    const int gens = pass == 0 ? 1 : G;
    uint32_t epoch = 1;
    for (int g = 0; g < gens; ++g) {
      if (pass == 1)                                 // fresh tails: the second half of every request's blocks is new in every generation
        for (uint32_t r = 0; r < R; ++r) {
          uint8_t* row = reqs.data() + (size_t)r * stride;
          const uint32_t nb = ((const uint32_t*)row)[1];
          for (uint32_t b = nb / 2u; b < nb; ++b) ((uint64_t*)(row + 8))[b] = mix(((uint64_t)(g + 1) << 40) + (uint64_t)r * 64u + b) | 2ull;
        }
      CK(hipMemcpy(d_reqs, reqs.data(), reqs.size(), hipMemcpyHostToDevice));
      EK(eppk_pick_batch_device(ctx, d_reqs, R, nullptr, d_picks, d_scores, nullptr));
      if (pass == 1) EK(eppk_index_insert_picks_device(ctx, d_reqs, d_picks, R, nullptr));
      uint32_t flags = 0;
      EK(eppk_launch_status(ctx, &flags));           // (synchronises)
      CK(hipMemcpy(gp.data(), d_picks, R * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(gs.data(), d_scores, R * 8, hipMemcpyDeviceToHost));
      if (orc_pick_batch_mt(cfg.chain, n_sc, (const eppk_pod_row*)pods.data(), P, oix, reqs.data(), B, R, nullptr, op.data(), os.data(), threads) != 0) { std::printf("oracle failed\n"); return 1; }
      char what[64];
      std::snprintf(what, sizeof what, pass == 0 ? "open loop, batch 0:" : "closed loop, generation %d:", g);
      compare(what);
      if (flags) { std::printf("launch status flags %u\n", flags); ++bad; }
      if (pass == 1) {
        orc_index_insert_picks(oix, reqs.data(), B, R, op.data());
        if (g % 2 == 1) {
          uint32_t e2 = 0, n_ev = 0;
          EK(eppk_index_advance_epoch(ctx, &e2));
          epoch = orc_index_advance_epoch(oix);
          if (e2 != epoch) { std::printf("epochs differ: %u vs %u\n", e2, epoch); ++bad; }
          if (epoch > 2u) {                            // keep one epoch: generation 3 drops what generations 0 and 1 learnt and nobody repeated
            EK(eppk_index_evict_older(ctx, epoch - 1u, &n_ev));
            const uint32_t o_ev = orc_index_evict_older(oix, epoch - 1u);
            std::printf("  evicted %u (oracle %u)\n", n_ev, o_ev);
            if (n_ev != o_ev) { std::printf("evicted: %u vs oracle %u\n", n_ev, o_ev); ++bad; }
          }
        }
        uint32_t size = 0; uint64_t dropped = 0;
        EK(eppk_index_size(ctx, &size)); EK(eppk_index_dropped(ctx, &dropped));
        const uint64_t osize = orc_index_size(oix);
        std::printf("  index: %u live hashes (oracle %llu), %llu dropped\n", size, (unsigned long long)osize, (unsigned long long)dropped);
        bad += size != osize || dropped != 0;
      }
    }
    orc_index_free(oix);
    eppk_destroy(ctx);
  }
  std::printf(bad ? "parity_quick FAILED (%d)\n" : "parity_quick ok\n", bad);
  return bad ? 1 : 0;
}
