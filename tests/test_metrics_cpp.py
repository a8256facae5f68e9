"""The snapshot producer's input side, compiled with g++ and run on the CPU (no GPU, no libeppk):
  * host/eppk_metrics.hpp — extractor: Prometheus text of a model server -> eppk_pod_row (docs/proposals/003-model-server-protocol);
  * host/eppk_scrape.hpp  — data source: HTTP GET of every endpoint's /metrics + the collector that keeps the latest row per endpoint
    (the DataSource / DataCollection interfaces of docs/proposals/1023-data-layer-architecture), against fixture servers started here."""
import http.server
import os
import socket
import subprocess
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")


def _build(name, extra=()):
    exe = os.path.join(CPP, name)
    subprocess.run(["g++", "-O1", "-std=c++17", "-Wall", "-Wextra", "-Werror", exe + ".cpp", "-o", exe, *extra], check=True)
    return exe


def test_metrics_extractor():
    out = subprocess.run([_build("test_metrics")], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr + out.stdout
    assert "metrics ok" in out.stdout


def test_sample_parser_against_prometheus_client():
    """Differential test of ParseSample (the C++ extractor's line parser) against prometheus_client's text parser -- an independent
    implementation of the exposition format -- on random expositions: metric names with colons, label values with every escape,
    odd spacing, timestamps, NaN / +Inf / -Inf / exponents, comment and blank lines."""
    import math
    import random
    import struct
    from prometheus_client.parser import text_string_to_metric_families
    exe = _build("test_metrics")
    rnd = random.Random(20260923)
    alphabet = 'abcXYZ019 _-:/.,{}=#\u00e9'

    def esc(v):
        return v.replace("\\", "\\\\").replace('"', '\\"').replace("\n", "\\n")

    total = escaped = 0
    for round_ in range(30):
        lines, want = [], []
        for i in range(rnd.randrange(1, 40)):
            kind = rnd.random()
            if kind < 0.1:
                lines.append(rnd.choice(["", "# HELP m_%d some text {with} \"quotes\"" % i, "# TYPE m_%d gauge" % i, "   "]))
                continue
            name = rnd.choice(["vllm:num_requests_waiting", "vllm:kv_cache_usage_perc", "nv_trt_llm_request_metrics", "m_%d" % i, "a:b:c_%d" % i])
            labels = {}
            for j in range(rnd.choice([0, 0, 1, 2, 5])):
                val = "".join(rnd.choice(alphabet + '"\\\n') for _ in range(rnd.randrange(0, 12)))
                labels["l%d_%s" % (j, rnd.choice(["x", "model_name", "le"]))] = val
            value = rnd.choice([0.0, 1.0, -1.5, 7, 1.7123e9, 0.4375, 1e-300, 1e300, math.inf, -math.inf, math.nan, rnd.uniform(-1e6, 1e6)])
            vtxt = {math.inf: "+Inf", -math.inf: "-Inf"}.get(value, "NaN" if isinstance(value, float) and math.isnan(value) else repr(value))
            lab = ""
            if labels or rnd.random() < 0.1:
                sep = rnd.choice([",", ", ", " ,"])
                lab = "{" + sep.join('%s="%s"' % (k, esc(v)) for k, v in labels.items()) + rnd.choice(["", ","] if labels else [""]) + "}"
            ts = rnd.choice(["", "", " 1712345678000", " -5"])
            lines.append(name + lab + rnd.choice([" ", "  ", "\t"]) + vtxt + ts)
            want.append((name, labels, value))
        body = "\n".join(lines) + "\n"
        ref = [(s_.name, dict(s_.labels), s_.value) for fam in text_string_to_metric_families(body) for s_ in fam.samples]
        assert len(ref) == len(want)                                           # the generator and the reference parser agree first
        out = subprocess.run([exe, "--dump"], input=body.encode(), capture_output=True, timeout=60)
        assert out.returncode == 0
        got = []
        for ln in out.stdout.decode().splitlines():
            assert ln != "MALFORMED", body
            f = ln.split(" ")
            eq = f.index("=")
            strs = [bytes.fromhex(x[1:]).decode() for x in f[:eq]]
            got.append((strs[0], dict(zip(strs[1::2], strs[2::2])), struct.unpack("<d", struct.pack("<Q", int(f[eq + 1], 16)))[0]))
        assert len(got) == len(ref), body
        for g_, r_ in zip(got, ref):
            assert g_[0] == r_[0] and g_[1] == r_[1], (g_, r_)
            assert (math.isnan(g_[2]) and math.isnan(r_[2])) or g_[2] == r_[2], (g_, r_)
            total += 1
            escaped += any(c in v for v in g_[1].values() for c in '"\\\n')
    assert total > 300 and escaped > 30, (total, escaped)


BODY_A = (b"# TYPE vllm:num_requests_waiting gauge\n"
          b"vllm:num_requests_waiting{model_name=\"m\"} 7.0\n"
          b"vllm:num_requests_running{model_name=\"m\"} 3.0\n"
          b"vllm:kv_cache_usage_perc{model_name=\"m\"} 0.4375\n"
          b"vllm:lora_requests_info{max_lora=\"4\",running_lora_adapters=\"adapter1,adapter2\",waiting_lora_adapters=\"\"} 1.7e+09\n")
BODY_B = b"vllm:num_requests_waiting 5\nvllm:kv_cache_usage_perc 0.5\n" + b"# padding\n" * 3000      # > one chunk, > one recv


def _handler(kind):
    class H(http.server.BaseHTTPRequestHandler):
        protocol_version = "HTTP/1.1"

        def log_message(self, *a):
            pass

        def do_GET(self):
            if self.path != "/metrics" or kind == "503":
                self.send_response(503 if kind == "503" else 404)
                self.send_header("Content-Length", "0")
                self.end_headers()
                return
            self.send_response(200)
            self.send_header("Content-Type", "text/plain; version=0.0.4")
            if kind == "len":
                self.send_header("Content-Length", str(len(BODY_A)))
                self.end_headers()
                self.wfile.write(BODY_A)
            else:
                self.send_header("Transfer-Encoding", "chunked")
                self.end_headers()
                for i in range(0, len(BODY_B), 4093):
                    part = BODY_B[i:i + 4093]
                    self.wfile.write(b"%x\r\n" % len(part) + part + b"\r\n")
                self.wfile.write(b"0\r\n\r\n")
    return H


def _with_fixture_servers(exe):
    class Server(http.server.ThreadingHTTPServer):
        request_queue_size = 256                 # the engine opens dozens of connections at once
        daemon_threads = True

    servers = [Server(("127.0.0.1", 0), _handler(k)) for k in ("len", "chunked", "503")]
    for s in servers:
        threading.Thread(target=s.serve_forever, daemon=True).start()
    silent = socket.socket()                 # accepts (backlog) and never answers
    silent.bind(("127.0.0.1", 0))
    silent.listen(64)
    closed = socket.socket()                 # a port nobody listens on
    closed.bind(("127.0.0.1", 0))
    closed_port = closed.getsockname()[1]
    closed.close()
    try:
        ports = [str(s.server_address[1]) for s in servers] + [str(silent.getsockname()[1]), str(closed_port)]
        out = subprocess.run([exe, *ports], capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stderr + out.stdout
        return out.stdout
    finally:
        for s in servers:
            s.shutdown()
            s.server_close()
        silent.close()


def test_metrics_data_source():
    assert "scrape ok" in _with_fixture_servers(_build("test_scrape", ["-pthread"]))


def test_snapshot_producer_end_to_end():
    """list the pool -> scrape -> pod rows -> GpuPicker::PublishSnapshot -> Pick(), with a stand-in backend (host/eppk_producer.hpp)."""
    import __graft_entry__ as g
    g.build()                        # GpuPicker calls the library's host-side helpers (eppk_round_robin, eppk_hash_prompt): link libeppk
    pkg = os.path.join(ROOT, "gateway-api-inference-extension_amd")
    assert "producer ok" in _with_fixture_servers(_build("test_producer", ["-pthread", f"-L{pkg}", "-leppk", f"-Wl,-rpath,{pkg}"]))
