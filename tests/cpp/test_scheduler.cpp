// GPU test of the C++ scheduling cycle (host/eppk_host.hpp: Scheduler, ProfileHandler, SchedulingResult) in the shape of
// docs/proposals/0845-scheduler-architecture-proposal/examples/example.yaml -- profiles `prefill` (filters, best-score) and `decode`
// (filter, random-top-3), profileSelection by prompt length -- end to end on the GPU against the ORACLE (oracle/oracle.h): every
// profile's picks and scores, bit for bit.  Test infrastructure: this file links liboracle, the product header does not.
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../gateway-api-inference-extension_amd/host/eppk_host.hpp"
#include "../../oracle/oracle.h"

using namespace eppk_host;

#define CHECK(x) do { if (!(x)) { std::fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #x); return 1; } } while (0)

int main() {
  const int P = 120, B = 8;
  std::vector<Endpoint> eps((size_t)P);
  std::vector<eppk_pod_row> rows((size_t)P);
  std::memset(rows.data(), 0, rows.size() * sizeof(eppk_pod_row));
  uint64_t x = 0x2545F4914F6CDD1Dull;
  auto rnd = [&] { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
  for (int i = 0; i < P; ++i) {
    eps[(size_t)i].address = "10.2.0." + std::to_string(i);
    eps[(size_t)i].port = "8000";
    eps[(size_t)i].labels["role"] = (i % 3 == 0) ? "prefill" : "decode";          // is-prefill / is-decode filters
    eps[(size_t)i].labels["accelerator"] = (i % 5 == 0) ? "none" : "mi355x";     // has-required-accelerator
    rows[(size_t)i].queue = (uint32_t)(rnd() % 12);
    rows[(size_t)i].kv_util = (double)(rnd() % 1025) / 1024.0;
    rows[(size_t)i].max_lora = 4;
  }
  std::vector<ProfileSpec> specs(2);
  specs[0].name = "prefill";
  specs[0].filter = [](const Endpoint& e) { return e.labels.at("role") == "prefill" && e.labels.at("accelerator") == "mi355x"; };
  specs[0].scorers = {{EPPK_SCORER_PREFIX, 3}, {EPPK_SCORER_QUEUE, 2}};           // (queue depth stands in for the example's latency-scorer)
  specs[0].picker = PickerKind::BestScore;
  specs[1].name = "decode";
  specs[1].filter = [](const Endpoint& e) { return e.labels.at("role") == "decode"; };
  specs[1].scorers = {{EPPK_SCORER_PREFIX, 3}, {EPPK_SCORER_KV, 5}};              // example.yaml:21-23
  specs[1].picker = PickerKind::RandomTopK;                                        // example.yaml:25 random-top-3
  specs[1].k = 3;
  DisaggTokenLengthHandler handler("prefill", "decode", 400);
  Scheduler sched;
  Scheduler::Options opt;
  opt.max_pods = 128; opt.max_blocks = B; opt.max_batch = 64;                      // (smaller than the batch: the groups are chunked)
  opt.index_slots = 1024;
  CHECK(sched.Configure(specs, &handler, opt).ok());
  CHECK(sched.PublishSnapshot(eps, rows, {}, 1).ok());

  // five system prompts, cached on a few pods of both roles
  std::vector<std::string> sys;
  for (int g = 0; g < 5; ++g) sys.push_back(std::string(256, (char)('A' + g)));
  orc_index* oix[2] = {orc_index_new(), orc_index_new()};
  std::vector<eppk_pod_row> prow[2] = {rows, rows};
  for (int pi = 0; pi < 2; ++pi)
    for (int i = 0; i < P; ++i)
      if (!specs[(size_t)pi].filter(eps[(size_t)i])) prow[pi][(size_t)i].flags |= EPPK_POD_INACTIVE;
  const std::string model = "base";
  for (int g = 0; g < 5; ++g) {
    uint64_t h[8];
    const int n = eppk_hash_prompt((const uint8_t*)model.data(), model.size(), (const uint8_t*)sys[(size_t)g].data(), sys[(size_t)g].size(), 64, h, 8);
    CHECK(n == 4);
    for (int pod : {g * 6, g * 6 + 1, g * 6 + 2, g * 6 + 3, 90 + g})
      for (int pi = 0; pi < 2; ++pi)
        for (int i = 0; i < n; ++i) {
          const uint32_t pp = (uint32_t)pod;
          CHECK(sched.IndexInsert(specs[(size_t)pi].name, &h[i], &pp, 1).ok());
          if (!(prow[pi][(size_t)pod].flags & EPPK_POD_INACTIVE)) orc_index_insert(oix[pi], h[i], pp);   // (a hole learns nothing: SEMANTICS.md 6b)
        }
  }

  const int N = 150;
  std::vector<Request> reqs((size_t)N);
  for (int i = 0; i < N; ++i) {
    reqs[(size_t)i].request_id = "req-" + std::to_string(i);
    reqs[(size_t)i].target_model = model;
    reqs[(size_t)i].prompt = sys[(size_t)(i % 5)] + std::string((size_t)(i % 2 ? 40 : 300), (char)('a' + i % 7)) + std::to_string(i);   // short / long
  }
  const uint64_t seed = 20260923ull;
  std::vector<SchedulingResult> res;
  std::vector<Status> st;
  CHECK(sched.ScheduleBatch(reqs, seed, &res, &st).ok());
  CHECK(res.size() == (size_t)N);

  // the oracle, profile by profile, over the same groups in the same order and chunks
  const size_t stride = 8u + 8u * (size_t)B;
  int n_prefill = 0, not_head = 0;
  for (int pi = 0; pi < 2; ++pi) {
    std::vector<int> group;
    for (int i = 0; i < N; ++i)
      if (pi == 1 || reqs[(size_t)i].prompt.size() >= 400) group.push_back(i);
    if (pi == 0) n_prefill = (int)group.size();
    eppk_weighted_scorer chain[2];
    for (int k = 0; k < 2; ++k) { chain[k].kind = (uint32_t)specs[(size_t)pi].scorers[(size_t)k].kind; chain[k].weight = specs[(size_t)pi].scorers[(size_t)k].weight; }
    for (size_t lo = 0; lo < group.size(); lo += opt.max_batch) {
      const uint32_t m = (uint32_t)std::min<size_t>(opt.max_batch, group.size() - lo);
      std::vector<uint8_t> rb((size_t)m * stride, 0);
      for (uint32_t i = 0; i < m; ++i) {
        const Request& rq = reqs[(size_t)group[lo + i]];
        eppk_req_hdr hdr; hdr.adapter = -1;
        hdr.n_blocks = (uint32_t)eppk_hash_prompt((const uint8_t*)model.data(), model.size(), (const uint8_t*)rq.prompt.data(), rq.prompt.size(), 64,
                                                  (uint64_t*)(rb.data() + (size_t)i * stride + 8), B);
        std::memcpy(rb.data() + (size_t)i * stride, &hdr, 8);
      }
      std::vector<int32_t> op(m), head(m);
      std::vector<double> os(m), hs(m);
      if (pi == 0) CHECK(orc_pick_batch(chain, 2, prow[pi].data(), P, oix[pi], rb.data(), B, m, nullptr, op.data(), os.data(), nullptr) == 0);
      else {
        CHECK(orc_pick_random_topk(chain, 2, prow[pi].data(), P, oix[pi], rb.data(), B, m, nullptr, 3, seed + lo, op.data(), os.data()) == 0);
        CHECK(orc_pick_batch(chain, 2, prow[pi].data(), P, oix[pi], rb.data(), B, m, nullptr, head.data(), hs.data(), nullptr) == 0);
      }
      for (uint32_t i = 0; i < m; ++i) {
        const SchedulingResult& sr = res[(size_t)group[lo + i]];
        auto it = sr.profile_results.find(specs[(size_t)pi].name);
        CHECK(it != sr.profile_results.end() && it->second.size() == 1);
        CHECK(op[i] >= 0 && it->second[0] != nullptr);
        CHECK(it->second[0]->address == eps[(size_t)op[i]].address);
        CHECK(specs[(size_t)pi].filter(*it->second[0]));                 // the profile's filters hold
        if (pi == 1 && op[i] != head[i]) ++not_head;
      }
    }
  }
  for (int i = 0; i < N; ++i) {
    CHECK(st[(size_t)i].ok() && res[(size_t)i].primary_profile_name == "decode");
    CHECK(res[(size_t)i].profile_results.count("prefill") == (reqs[(size_t)i].prompt.size() >= 400 ? 1u : 0u));
  }
  CHECK(n_prefill > 0 && n_prefill < N && not_head > 0);
  orc_index_free(oix[0]); orc_index_free(oix[1]);
  std::printf("scheduler ok: %d requests, %d through prefill + decode, %d decode picks off the head of their top-3; every profile equals the oracle\n", N, n_prefill, not_head);
  return 0;
}
