// Measurement tool (CPU only): what the C++ host twin's request pipeline carries when the device costs nothing -- GpuPicker
// (host/eppk_host.hpp) with a backend that returns candidate 0 at once: T request threads call Pick() in a loop (each hashes its 2 KiB
// prompt and builds its candidate mask itself), one dispatcher thread batches.   hostpipe [threads] [endpoints] [picks per thread]
// Build: g++ -O2 -std=c++17 -pthread scripts/micro/hostpipe.cpp -o scripts/micro/hostpipe -Lgateway-api-inference-extension_amd -leppk -Wl,-rpath,$PWD/gateway-api-inference-extension_amd
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>

#include "../../gateway-api-inference-extension_amd/host/eppk_host.hpp"

using namespace eppk_host;

class NullBackend : public Backend {
 public:
  int Publish(const eppk_pod_row*, uint32_t, uint64_t) override { return EPPK_OK; }
  int PickBatch(const void*, uint32_t n, const uint64_t*, int32_t* picks, double* scores) override {
    for (uint32_t r = 0; r < n; ++r) { picks[r] = 0; if (scores) scores[r] = 0.0; }
    return EPPK_OK;
  }
  int PickTopK(const void*, uint32_t, const uint64_t*, uint32_t, int32_t*, double*) override { return EPPK_ERR_ARG; }
  int IndexInsert(const uint64_t*, const uint32_t*, uint32_t) override { return EPPK_OK; }
  int IndexRemovePod(uint32_t) override { return EPPK_OK; }
  int IndexAdvanceEpoch(uint32_t* e) override { *e = 1; return EPPK_OK; }
  int IndexEvictOlder(uint32_t, uint32_t* n) override { *n = 0; return EPPK_OK; }
  std::string LastError() const override { return ""; }
};

int main(int argc, char** argv) {
  const int T = argc > 1 ? std::atoi(argv[1]) : 64, P = argc > 2 ? std::atoi(argv[2]) : 256, N = argc > 3 ? std::atoi(argv[3]) : 2000;
  GpuPickerOptions o;
  o.max_pods = (uint32_t)P; o.max_blocks = 32; o.max_batch = 4096;
  o.window = std::chrono::microseconds(200);
  GpuPicker gp(std::unique_ptr<Backend>(new NullBackend()), o);
  std::vector<Endpoint> eps((size_t)P);
  std::vector<eppk_pod_row> rows((size_t)P);
  std::memset(rows.data(), 0, rows.size() * sizeof(eppk_pod_row));
  for (int i = 0; i < P; ++i) { eps[(size_t)i].address = "10.0." + std::to_string(i / 250) + "." + std::to_string(i % 250 + 1); eps[(size_t)i].port = "8000"; }
  if (!gp.PublishSnapshot(eps, rows, {}, 1).ok()) return 1;
  std::vector<const Endpoint*> all;
  for (auto& e : eps) all.push_back(&e);
  const auto t0 = std::chrono::steady_clock::now();
  std::vector<std::thread> th;
  for (int t = 0; t < T; ++t)
    th.emplace_back([&, t] {
      PickRequest rq;
      rq.model = "base";
      rq.body = std::string(2048, (char)('a' + t % 26));
      PickResult r;
      for (int i = 0; i < N; ++i) { rq.body[(size_t)(i % 2048)] = (char)('A' + i % 26); if (!gp.Pick(rq, all, &r).ok()) std::abort(); }
    });
  for (auto& x : th) x.join();
  const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  std::printf("%d request threads x %d picks, %d endpoints: %.0f picks/s, %llu batches (largest %llu), %.1f us per pick per thread\n", T, N, P, T * (double)N / s,
              (unsigned long long)gp.batches(), (unsigned long long)gp.largest_batch(), s / N * 1e6);
  return 0;
}
