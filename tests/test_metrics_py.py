"""metrics.py (the Python mirror of the snapshot producer) on the CPU: its parser against prometheus_client's and against the C++
extractor's on the same bodies; the producer against fixture servers with a recording `publish`."""
import http.server
import math
import os
import random
import subprocess
import threading

import numpy as np

from test_metrics_cpp import BODY_A, BODY_B, _build, _handler

VLLM = ("# TYPE vllm:num_requests_waiting gauge\n"
        'vllm:num_requests_waiting{model_name="a"} 2\nvllm:num_requests_waiting{model_name="b"} 5.0 1712345678000\n'
        'vllm:num_requests_running{model_name="a"} 3\n'
        'vllm:kv_cache_usage_perc{model_name="a"} 0.25\nvllm:kv_cache_usage_perc{model_name="b"} 0.4375\n'
        'vllm:lora_requests_info{max_lora="4",running_lora_adapters="adapter1",waiting_lora_adapters=""} 1.7123e+09\n'
        'vllm:lora_requests_info{max_lora="8",running_lora_adapters="adapter2, sql-lora",waiting_lora_adapters="big,nobody"} 1.7124e+09\n'
        "garbage line\n")


def test_extractor_rows(pkg):
    m = pkg.metrics
    row, complete, unknown = m.parse_model_server_metrics(VLLM, {"adapter1": 0, "adapter2": 1, "sql-lora": 64, "big": 127})
    assert complete and unknown == ["nobody"]
    assert (int(row["queue"]), int(row["running"]), float(row["kv_util"]), int(row["max_lora"]), int(row["flags"])) == (7, 3, 0.4375, 8, 0)
    assert row["active"].tolist() == [2, 1] and row["waiting"].tolist() == [0, 1 << 63]
    tr = m.MetricNames(queued="nv_trt_llm_request_metrics{request_type=waiting}", running='nv_trt_llm_request_metrics{request_type="scheduled"}',
                       kv_util="nv_trt_llm_kv_cache_block_metrics{kv_cache_block_type=fraction}")
    body = ('nv_trt_llm_request_metrics{model="m",request_type="waiting"} 11\nnv_trt_llm_request_metrics{model="m",request_type="scheduled"} 5\n'
            'nv_trt_llm_request_metrics{model="m",request_type="max"} 512\nnv_trt_llm_kv_cache_block_metrics{kv_cache_block_type="fraction"} 0.25\n')
    row, complete, _ = m.parse_model_server_metrics(body, {}, tr)
    assert complete and (int(row["queue"]), int(row["running"]), float(row["kv_util"])) == (11, 5, 0.25)
    row, complete, _ = m.parse_model_server_metrics("vllm:num_requests_waiting -4\nvllm:kv_cache_usage_perc NaN\nvllm:num_requests_running 1e12\n", {})
    assert not complete and int(row["queue"]) == 0 and int(row["running"]) == 0xFFFFFFFF


def test_three_parsers_agree(pkg):
    """parse_sample (Python), ParseSample (C++, through tests/cpp/test_metrics --dump) and prometheus_client on random expositions."""
    from prometheus_client.parser import text_string_to_metric_families
    import struct
    exe = _build("test_metrics")
    rnd = random.Random(7)
    lines = []
    for i in range(400):
        labels = {f"l{j}": "".join(rnd.choice('ab ,=}{"\\\n:é') for _ in range(rnd.randrange(0, 9))) for j in range(rnd.choice([0, 1, 3]))}
        esc = lambda v: v.replace("\\", "\\\\").replace('"', '\\"').replace("\n", "\\n")
        value = rnd.choice([0.0, 7, -1.5, 1.7e9, math.inf, -math.inf, rnd.uniform(-1e3, 1e3)])
        vtxt = {math.inf: "+Inf", -math.inf: "-Inf"}.get(value, repr(value))
        lab = "{" + ",".join(f'{k}="{esc(v)}"' for k, v in labels.items()) + "}" if labels else ""
        lines.append(f"vllm:m_{i}{lab} {vtxt}" + rnd.choice(["", " 1712345678000"]))
    body = "\n".join(lines) + "\n"
    ref = [(s.name, dict(s.labels), s.value) for fam in text_string_to_metric_families(body) for s in fam.samples]
    py = [pkg.metrics.parse_sample(ln) for ln in lines]
    assert py == ref
    out = subprocess.run([exe, "--dump"], input=body.encode(), capture_output=True, timeout=60)
    cpp = []
    for ln in out.stdout.decode().splitlines():
        f = ln.split(" ")
        eq = f.index("=")
        strs = [bytes.fromhex(x[1:]).decode() for x in f[:eq]]
        cpp.append((strs[0], dict(zip(strs[1::2], strs[2::2])), struct.unpack("<d", struct.pack("<Q", int(f[eq + 1], 16)))[0]))
    assert cpp == ref


def test_producer_against_fixture_servers(pkg):
    m = pkg.metrics

    class Server(http.server.ThreadingHTTPServer):
        daemon_threads = True

    servers = [Server(("127.0.0.1", 0), _handler(k)) for k in ("len", "chunked", "503")]
    for s in servers:
        threading.Thread(target=s.serve_forever, daemon=True).start()
    try:
        pool = [m.ScrapedEndpoint("127.0.0.1", str(s.server_address[1])) for s in servers]
        published = []
        prod = m.SnapshotProducer(publish=lambda rows, epoch: published.append((rows.copy(), epoch)), list_endpoints=lambda: list(pool), timeout_s=2.0)
        assert prod.refresh() == 2                                   # the 503 server is not in the snapshot
        rows, epoch = published[-1]
        assert epoch == 1 and rows.shape == (2,) and rows["queue"].tolist() == [7, 5] and rows["flags"].tolist() == [0, 0]
        assert prod.adapters == {"adapter1": 0, "adapter2": 1} and int(rows["active"][0][0]) == 3 and int(rows["max_lora"][0]) == 4
        assert prod.slots == {pool[0].key: 0, pool[1].key: 1}
        del pool[0]                                                  # churn: slot 0 becomes a hole, the other endpoint keeps index 1
        assert prod.refresh() == 1
        rows, epoch = published[-1]
        assert epoch == 2 and rows["flags"].tolist() == [m.POD_INACTIVE, 0] and int(rows["queue"][1]) == 5
        pool.insert(0, m.ScrapedEndpoint("127.0.0.1", str(servers[0].server_address[1])))   # a newcomer takes the lowest free slot
        assert prod.refresh() == 2 and prod.slots[pool[0].key] == 0
        stop = threading.Event()
        t = threading.Thread(target=prod.run, args=(stop, 0.01))
        t.start()
        import time
        t_end = time.monotonic() + 30
        while prod.epoch < 6 and time.monotonic() < t_end:
            time.sleep(0.005)
        stop.set()
        t.join()
        assert published[-1][1] >= 6
    finally:
        for s in servers:
            s.shutdown()
            s.server_close()
