// host/eppk_scrape.hpp against real sockets: argv = ports of the fixture servers tests/test_metrics_cpp.py starts
//   argv[1] content-length server (vLLM body, queue 7)   argv[2] chunked server (queue 5)   argv[3] a server answering 503
//   argv[4] a server that accepts and never answers      argv[5] a closed port
#include <cstdio>
#include <string>

#include "../../gateway-api-inference-extension_amd/host/eppk_scrape.hpp"

using namespace eppk_host;

#define CHECK(x) do { if (!(x)) { std::fprintf(stderr, "%s:%d: CHECK(%s) failed\n", __FILE__, __LINE__, #x); return 1; } } while (0)

int main(int argc, char** argv) {
  if (argc != 6) { std::fprintf(stderr, "usage: test_scrape p_len p_chunked p_503 p_silent p_closed\n"); return 2; }
  std::string body, err;
  CHECK(HttpGet("127.0.0.1", argv[1], "/metrics", 2000, &body, &err));
  CHECK(body.find("vllm:num_requests_waiting") != std::string::npos && body.back() == '\n');
  CHECK(HttpGet("localhost", argv[2], "/metrics", 2000, &body, &err));
  CHECK(body.find("vllm:kv_cache_usage_perc") != std::string::npos && body.find("\r\n") == std::string::npos);
  CHECK(!HttpGet("127.0.0.1", argv[3], "/metrics", 2000, &body, &err) && err == "status 503");
  const auto t0 = SteadyClock::now();
  CHECK(!HttpGet("127.0.0.1", argv[4], "/metrics", 300, &body, &err) && err.find("timed out") != std::string::npos);
  const auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(SteadyClock::now() - t0).count();
  CHECK(ms >= 250 && ms < 1500);
  CHECK(!HttpGet("127.0.0.1", argv[5], "/metrics", 500, &body, &err) && err.find("connect") != std::string::npos);
  CHECK(!HttpGet("no-such-host.invalid", "80", "/metrics", 500, &body, &err) && err.find("resolve") != std::string::npos);

  // many exchanges on one thread: 400 GETs, every fourth to the closed port, 64 sockets at a time
  {
    std::vector<HttpRequest> reqs;
    for (int i = 0; i < 400; ++i) reqs.push_back({"127.0.0.1", argv[i % 4 == 3 ? 5 : 1 + (i & 1)], "/metrics"});
    const auto m0 = SteadyClock::now();
    std::vector<HttpResult> res = HttpGetMany(reqs, 3000, 64);
    const auto mms = std::chrono::duration_cast<std::chrono::milliseconds>(SteadyClock::now() - m0).count();
    CHECK(res.size() == 400);
    size_t ok = 0, refused = 0;
    for (int i = 0; i < 400; ++i) {
      if (i % 4 == 3) { CHECK(!res[(size_t)i].ok && res[(size_t)i].error.find("connect") != std::string::npos); ++refused; }
      else { CHECK(res[(size_t)i].ok && res[(size_t)i].body.find(i & 1 ? "vllm:kv_cache_usage_perc 0.5" : "vllm:num_requests_waiting{model_name=\"m\"} 7.0") != std::string::npos); ++ok; }
    }
    CHECK(ok == 300 && refused == 100);
    std::fprintf(stderr, "400 exchanges, 64 in flight: %lld ms\n", (long long)mms);
    CHECK(mms < 20000);
  }

  // the data source + the collector: five endpoints, two usable
  MetricsDataSource::Options o;
  o.interval_ms = 20;
  o.timeout_ms = 300;
  o.max_inflight = 4;
  MetricsDataSource src(o);
  auto rows = std::make_shared<PodRowCollector>(std::map<std::string, int32_t>{{"adapter1", 0}, {"adapter2", 1}});
  src.Subscribe(rows);
  std::vector<ScrapeTarget> eps;
  std::vector<std::string> ids;
  for (int i = 1; i <= 5; ++i) { eps.push_back({"pod-" + std::to_string(i), "127.0.0.1", argv[i], "/metrics"}); ids.push_back(eps.back().id); }
  src.UpdateEndpoints(eps);
  CHECK(src.Type() == "metrics");
  CHECK(src.ScrapeOnce() == 2);
  uint32_t holes = 0;
  std::vector<eppk_pod_row> r = rows->Rows(ids, 0, &holes);
  CHECK(r.size() == 5 && holes == 3);
  CHECK(r[0].flags == 0 && r[0].queue == 7 && r[0].running == 3 && r[0].kv_util == 0.4375 && r[0].max_lora == 4 && r[0].active[0] == 3ull);
  CHECK(r[1].flags == 0 && r[1].queue == 5 && r[1].kv_util == 0.5);
  CHECK(r[2].flags == EPPK_POD_INACTIVE && r[3].flags == EPPK_POD_INACTIVE && r[4].flags == EPPK_POD_INACTIVE);
  std::string why;
  CHECK(!rows->Latest("pod-3", nullptr, &why) && why == "status 503");
  CHECK(!rows->Latest("pod-9", nullptr));
  CHECK(rows->failures() == 3);

  // the timer loop: a few rounds, then the set of endpoints shrinks to the two good ones; rows age out
  src.UpdateEndpoints({eps[0], eps[1]});
  src.Start();
  src.Start();   // idempotent
  for (int i = 0; i < 200 && src.rounds() < 5; ++i) std::this_thread::sleep_for(std::chrono::milliseconds(10));
  src.Stop();
  CHECK(src.rounds() >= 5);
  r = rows->Rows({"pod-2", "pod-1"}, 1000, &holes);
  CHECK(holes == 0 && r[0].queue == 5 && r[1].queue == 7);
  std::this_thread::sleep_for(std::chrono::milliseconds(60));
  r = rows->Rows({"pod-2", "pod-1"}, 30, &holes);       // nothing scraped for 60 ms, 30 ms allowed: both are holes now
  CHECK(holes == 2);
  rows->Forget("pod-1");
  CHECK(!rows->Latest("pod-1", nullptr));
  std::printf("scrape ok\n");
  return 0;
}
