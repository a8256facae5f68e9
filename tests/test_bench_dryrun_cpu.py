"""bench.py's control flow, end to end on CPU: torch.cuda, the process group and the HIP picker are replaced by stand-ins
(streams and events that do nothing, gloo, the oracle writing picks through the same raw pointers), so that the N>1 code
path -- ring of pick buffers, bucketed all-gather, flush before the fence, result checks, the JSON line -- runs here,
where there is no GPU.  Nothing about speed is asserted: the numbers of such a run are meaningless by construction."""
import ctypes
import io
import json
import os
import sys
from contextlib import redirect_stdout

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _Stream:
    _next = 1

    def __init__(self, device=None):
        _Stream._next += 1
        self.cuda_stream = 0x1000 + _Stream._next

    def wait_event(self, ev):
        assert ev.recorded, "waiting for an event that was never recorded"

    def wait_stream(self, other):
        assert other is not self

    def synchronize(self):
        pass

    def query(self):          # (bench.py polls the streams ahead of the closing synchronize of a timed region)
        return True


class _Event:
    def __init__(self, enable_timing=False):
        self.recorded = False

    def record(self, stream=None):
        self.recorded = True

    def elapsed_time(self, other):
        assert self.recorded and other.recorded
        return 0.05


def _fake_picker_class(pkg, orc, log):
    class FakePicker:
        def __init__(self, chain, max_pods, max_blocks, max_batch, index_slots=0, device=0):
            self.chain, self.B, self.oix, self.pods = chain, max_blocks, orc.OracleIndex(), None
            self.stride = 8 + 8 * max_blocks
            self.prof, self.launches, self.last_stream = False, 0, None

        def publish(self, pods):
            self.pods = pods

        def index_insert(self, h, p):
            self.oix.insert(h, p)

        def _pick(self, reqs):
            p, s, _ = orc.pick_batch(self.chain, self.pods, self.oix, reqs, self.B)
            return p, s

        def pick(self, reqs):
            return self._pick(reqs)

        def pick_device(self, p_reqs, n, mask, p_pick, p_score, stream=0):
            assert mask is None and stream != 0
            raw = (ctypes.c_uint8 * (n * self.stride)).from_address(p_reqs)
            reqs = np.frombuffer(raw, dtype=np.uint64).reshape(n, 1 + self.B).copy()
            picks, scores = self._pick(reqs)
            ctypes.memmove(p_pick, picks.astype(np.int32).ctypes.data, 4 * n)
            if p_score:
                ctypes.memmove(p_score, scores.astype(np.float64).ctypes.data, 8 * n)
            self.launches += self.prof
            self.last_stream = stream
            log.append(("pick", p_pick, stream))

        def pick_learn_device(self, p_reqs, n, mask, p_pick, p_score, stream=0):
            self.pick_device(p_reqs, n, mask, p_pick, p_score, stream)
            self.index_insert_picks_device(p_reqs, p_pick, n, stream)

        def index_insert_picks_device(self, p_reqs, p_picks, n, stream=0):
            raw = (ctypes.c_uint8 * (n * self.stride)).from_address(p_reqs)
            reqs = np.frombuffer(raw, dtype=np.uint64).reshape(n, 1 + self.B).copy()
            picks = np.frombuffer((ctypes.c_int32 * n).from_address(p_picks), dtype=np.int32).copy()
            self.oix.insert_picks(reqs, self.B, picks)
            log.append(("learn", n))

        def index_advance_epoch(self):
            return self.oix.advance_epoch()

        def index_evict_older(self, min_epoch):
            return self.oix.evict_older(min_epoch)

        def index_evict_older_device(self, min_epoch, stream=0):
            self.oix.evict_older(min_epoch)

        def index_size(self):
            return self.oix.size()

        def index_dropped(self):
            return 0

        def launch_status(self):
            return 0

        def chain_is_fused(self):
            return 1

        def stream_wait_pick(self, waiting_stream):
            assert self.last_stream is not None and waiting_stream != self.last_stream
            log.append(("wait", waiting_stream))

        def profile(self, on):
            self.prof = bool(on)

        def profile_drain(self):
            n, self.launches = self.launches, 0
            return [0.05] * n

        def quad_stats(self):
            return 0, 0

        def profile_bytes(self):
            return 1000 * max(self.launches, 1), 17 * max(self.launches, 1), max(self.launches, 1)

        def close(self):
            pass

    return FakePicker


@pytest.mark.parametrize("argv", [["--force-dist", "--steps", "11", "--warmup", "3", "--p99-samples", "40"],
                                  ["--force-dist", "--steps", "8", "--warmup", "8", "--gather-every", "1", "--inflight", "1", "--scaling", "weak"],
                                  ["--steps", "5", "--warmup", "2", "--batches", "3"],
                                  ["--steps", "4", "--warmup", "1", "--batches", "2", "--revisit-leg"],
                                  ["--steps", "6", "--warmup", "2", "--batches", "4", "--closed-loop", "--cl-slots", "4096", "--cl-verify", "3"]])
def test_bench_control_flow_on_cpu(pkg, orc, monkeypatch, argv):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    import bench

    log = []
    real_device = torch.device
    monkeypatch.setattr(torch, "device", lambda *a, **k: real_device("cpu"))
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "set_device", lambda *a, **k: None)
    monkeypatch.setattr(torch.cuda, "set_stream", lambda *a, **k: None)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    monkeypatch.setattr(torch.cuda, "Stream", _Stream)
    monkeypatch.setattr(torch.cuda, "Event", _Event)
    real_init = dist.init_process_group
    monkeypatch.setattr(dist, "init_process_group", lambda backend=None, device_id=None, **k: real_init("gloo", rank=0, world_size=1))
    monkeypatch.setattr(pkg, "BatchedPicker", _fake_picker_class(pkg, orc, log))
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", str(_free_port()))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    base = ["bench.py", "--config", "3", "--requests", "96", "--host-path", "3"]
    if "--p99-samples" not in argv:
        base += ["--p99-samples", "0"]
    monkeypatch.setattr(sys, "argv", base + argv)
    out = io.StringIO()
    try:
        with redirect_stdout(out):
            bench.main()
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()
    lines = [ln for ln in out.getvalue().splitlines() if ln.strip()]
    d = json.loads(lines[-1])                                           # the JSON line is the last thing on stdout
    closed = "--closed-loop" in argv
    # `config`: at most 20 scalar keys (what the driver's stored record keeps), the workload first; the rest is in `config_detail`
    assert 5 <= len(d["config"]) <= 20 and all(not isinstance(v, (dict, list)) for v in d["config"].values()), d["config"]
    assert list(d["config"])[:5] == ["workload", "requests_per_step", "pods", "blocks_per_request", "chain"]
    assert d["config_detail"]["workload"] == d["config"]["workload"]
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline") + (() if closed else ("cpu_baseline",)):
        assert key in d, key
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "bytes_per_launch", "model_frac", "kernel_p99_ms", "kernel_samples"):
        assert key in d["roofline"], key
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["traffic"] is None      # no stamped PMC file for this fake build
    assert d["n_gpus"] == 1 and d["unit"] == "decisions/s" and d["vs_baseline"] is None
    assert d["steps"] == int(argv[argv.index("--steps") + 1])
    n_picks = sum(1 for e in log if e[0] == "pick")
    if closed:
        assert d["closed_loop"]["picks_equal_oracle"] and d["closed_loop"]["scores_bitwise_equal_oracle"]
        assert d["closed_loop"]["generations_verified"] == 3 and d["config_detail"]["closed_loop"] is True
        rc = d["roofline_closed_loop"]
        assert rc["bound"] == "hbm-random-lines" and abs(rc["frac"] - rc["achieved"] / rc["peak"]) < 1e-12
        assert set(rc["step_parts_ms"]) >= {"pick", "index_update", "ageing_per_step"} and rc["per_step"]["new_keys"] >= 0
        # every pick is followed by its index update (3 verified generations + the timed loop + the 24 instrumented steps of the roofline object)
        assert sum(1 for e in log if e[0] == "learn") == n_picks == 3 + d["steps"] + d["warmup"] + 24
        return
    for key in ("value", "unit", "cores", "kind", "sample", "algorithm", "single_thread_value", "per_request_loop_value"):
        assert key in d["cpu_baseline"], key
    assert d["parity"]["picks_equal_oracle"] and d["parity"]["scores_bitwise_equal_oracle"]
    if "--force-dist" in argv:
        both = "--scaling" not in argv
        assert d["scaling"] == "weak" and d["config_detail"]["ranks_seen"] == 1          # N > 1: weak scaling is the headline, strong beside it
        assert ("strong" in d) == both and "weak" not in d
        extra = max(0, 40 - d["steps"]) if "--p99-samples" in argv else 0
        steps = d["steps"] + d["warmup"]
        if both:
            # strong scaling scores the shards of a whole gather bucket (16 batches) with ONE launch; weak scaling one launch per batch
            assert "ONE launch" in d["strong"]["note"] and d["strong"]["requests_per_launch"] == 16 * 96 and d["config_detail"]["requests_per_launch"] == 96
            assert "int16" not in d["config_detail"]["sharding"]                                   # (--pack16 is opt-in)
            assert n_picks >= (steps + 15) // 16 + steps + extra                              # (extra = launches beyond the timed region: samples)
        else:
            assert n_picks >= steps + extra
        # cross-stream dependencies: the collective stream waits for the last launch on every compute stream of a bucket, not for every launch
        n_waits = sum(1 for e in log if e[0] == "wait")
        assert (steps + 15) // 16 <= n_waits <= n_picks
        if extra:
            assert d["roofline"]["kernel_samples"] >= 40
    else:
        assert d["scaling"] == "weak" and n_picks >= d["steps"] + d["warmup"]
        if "--revisit-leg" in argv:
            # the returning-requests leg: batch 0 learned, then batches with 0 / 25 / 50 / 100 % of its rows coming back, each against the oracle
            rv = d["revisit"]
            assert rv["learned_batch_picks_equal_oracle"] and rv["index_size_equal_oracle"]
            assert list(rv["by_fraction"]) == ["0.0", "0.25", "0.5", "1.0"]
            assert all(v["picks_and_scores_equal_oracle"] for v in rv["by_fraction"].values())
            assert rv["by_fraction"]["1.0"]["returning_requests"] == 96 and rv["by_fraction"]["0.0"]["returning_requests"] == 0
            assert d["config"]["revisit_50_equal_oracle"] is True and d["config"]["revisit_50_value"] > 0
        elif "--batches" in argv:
            assert d["config_detail"]["distinct_batches"] == 3


def _bench_worker(rank, world, port, outdir, extra_args=("--batches", "5"), requests=64):
    """One rank of a world-size-2 dry run: the same stand-ins, patched by hand (no pytest fixtures in a spawned process)."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import __graft_entry__ as g
    import bench
    pkg, orc = g.load_package(), g.load_oracle()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    real_device, real_init = torch.device, dist.init_process_group
    torch.device = lambda *a, **k: real_device("cpu")
    torch.cuda.is_available = lambda: True
    torch.cuda.set_device = torch.cuda.set_stream = torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.Stream, torch.cuda.Event = _Stream, _Event
    dist.init_process_group = lambda backend=None, device_id=None, **k: real_init("gloo", rank=rank, world_size=world)
    log = []
    pkg.BatchedPicker = _fake_picker_class(pkg, orc, log)
    sys.argv = ["bench.py", "--gpus", str(world), "--config", "3", "--requests", str(requests), "--steps", "10", "--warmup", "3", "--p99-samples", "0",
                *extra_args]
    out = io.StringIO()
    with redirect_stdout(out):
        bench.main()
    with open(os.path.join(outdir, f"bench_rank{rank}.out"), "w") as f:
        f.write(out.getvalue())


@pytest.mark.parametrize("extra,grouped", [(("--batches", "5", "--pack16"), False),            # 5 batches do not tile into buckets: a launch per shard
                                           (("--batches", "8", "--gather-every", "4"), True)])  # one launch per bucket of 4 shards
def test_bench_two_ranks_on_cpu(tmp_path, extra, grouped):
    """World size 2 over gloo: both ranks run bench.py's N>1 path to the end, rank 0 alone prints the JSON line, with the
    whole-job aggregate (requests of BOTH ranks) in it."""
    world = 2
    mp.spawn(_bench_worker, args=(world, _free_port(), str(tmp_path), extra), nprocs=world, join=True)
    out0 = [ln for ln in open(tmp_path / "bench_rank0.out").read().splitlines() if ln.strip()]
    out1 = [ln for ln in open(tmp_path / "bench_rank1.out").read().splitlines() if ln.strip()]
    assert not any(ln.lstrip().startswith("{") for ln in out1)
    d = json.loads(out0[-1])
    assert d["n_gpus"] == 2 and d["steps"] == 10 and d["warmup"] == 3
    # headline = weak scaling: a whole 64-request batch per rank and step, the aggregate counts BOTH ranks; every rank ends up with the
    # picks of both (rank 0 checks each rank's part against the oracle on the batch that rank scored)
    assert d["scaling"] == "weak" and d["config_detail"]["requests_per_gpu"] == 64 and d["config_detail"]["requests_per_step"] == 128
    assert "cpu_baseline" not in d and "one whole batch per rank" in d["config_detail"]["sharding"] and d["config_detail"]["ranks_seen"] == 2
    assert ("int16" in d["config_detail"]["sharding"]) == ("--pack16" in extra) and d["config_detail"]["requests_per_launch"] == 64
    # what `value` is: the metric names the scaling, the note quotes north_star with rank 0's measured launch time, the communicator is asked
    assert "closed loop" not in d["metric"]        # (a reduced workload here: the headline's names are checked in test_metric_names_say_which_scaling)
    assert "outgrows one device" in d["scaling_note"] and "weak" in d["scaling_note"]
    assert d["config_detail"]["per_rank_kernel_us"] > 0 and d["config_detail"]["collective_us"]["collectives"] >= 1 and d["config_detail"]["collective_us"]["p50"] >= 0
    cl = d["completion_latency"]                  # N > 1: when a batch's picks exist on every rank (per gather bucket)
    assert cl["buckets"] >= 1 and cl["p50_ms"] <= cl["p99_ms"] <= cl["max_ms"] and cl["batches_per_bucket"] >= 1
    assert d["parity"] == {"gathered_picks_equal_oracle": True, "ranks_checked": 2}
    assert abs(d["value"] - 2 * 64 * 10 / (d["ms_per_step"] * 1e-3 * 10)) < 1e-6 * d["value"]
    # strong scaling timed beside it: each 64-request batch is split 32 + 32
    st = d["strong"]
    assert st["requests_per_gpu"] == 32 and st["requests_per_step"] == 64 and abs(st["value"] - 64 * 10 / (st["ms_per_step"] * 1e-3 * 10)) < 1e-6 * st["value"]
    assert ("ONE launch" in st["note"]) == grouped and st["requests_per_launch"] == (4 * 32 if grouped else 32)
    assert st["completion_latency_p50_ms"] <= st["completion_latency_p99_ms"]
    # BASELINE.json configs[4] first class in `config` (the driver's stored record keeps `config` in full): the strong-scaled single batch
    c = d["config"]
    assert c["strong_value"] == st["value"] and c["strong_ms_per_step"] == st["ms_per_step"]
    assert c["strong_completion_latency_p99_ms"] == st["completion_latency_p99_ms"] and c["completion_latency_p99_ms"] == cl["p99_ms"]
    # ... and `config` is what the driver's record can hold: a handful of scalars, the N > 1 figures among them
    assert len(c) <= 20 and all(not isinstance(v, (dict, list)) for v in c.values())
    assert c["ranks_seen"] == 2 and c["collective_us"] >= 0 and c["per_rank_kernel_us"] > 0 and c["requests_per_step"] == 128


def test_bench_two_ranks_ragged_shards_grouped(tmp_path):
    """71 requests over 2 ranks: shards of 36 and 35 rows; the short shard is padded with filler rows inside the grouped launches and
    the gathered batch still equals the oracle's."""
    world = 2
    mp.spawn(_bench_worker, args=(world, _free_port(), str(tmp_path), ("--batches", "8", "--gather-every", "4", "--scaling", "strong"), 71), nprocs=world, join=True)
    out0 = [ln for ln in open(tmp_path / "bench_rank0.out").read().splitlines() if ln.strip()]
    d = json.loads(out0[-1])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config_detail"]["requests_per_gpu"] == 36 and d["config_detail"]["requests_per_step"] == 71
    assert "ONE launch" in d["config_detail"]["sharding"] and d["config_detail"]["requests_per_launch"] == 4 * 36
    assert "strong" in d["scaling_note"] and d["config_detail"]["ranks_seen"] == 2 and d["config_detail"]["per_rank_kernel_us"] > 0
    assert d["completion_latency"]["p50_ms"] <= d["completion_latency"]["p99_ms"]
    assert d["parity"]["gathered_picks_equal_oracle"] is True


def test_metric_names_say_which_scaling():
    """At N > 1 the line's `metric` must not be readable as BASELINE.json configs[4]'s ONE 64k batch when `value` is the aggregate of N
    replicas, nor the other way round (round-3 verdict item 4)."""
    import types
    import bench
    wl = types.SimpleNamespace(name="C5 64kx4096 full chain + prefix B=32")
    args = types.SimpleNamespace(groups=256, zipf=1.0, closed_loop=False)
    one = bench.metric_name(True, "single", 1, wl, args)
    weak = bench.metric_name(True, "weak", 8, wl, args)
    strong = bench.metric_name(True, "strong", 8, wl, args)
    assert one == "routing decisions/sec, 64k-req x 4096-pod batch"
    assert "8 x 64k-req" in weak and "weak scaling" in weak and "replicas" in weak
    assert "ONE 64k-req" in strong and "8 ranks" in strong and "strong scaling" in strong
    assert len({one, weak, strong}) == 3
    assert "C5" in bench.metric_name(False, "weak", 8, wl, args)          # a reduced workload never carries the headline's name
