#!/usr/bin/env python3
"""bench.py — routing decisions/s of the batched endpoint pick on MI355X (BASELINE.json metric).

A "step" = one pass of the hot path over one batch: request rows already resident in HBM ->
fused pick kernel -> picks in HBM (+ for N>1 an RCCL all-gather of the per-rank picks).
N=1 workload: BASELINE.json configs[4] ("64k req x 4096 pods, full scorer chain + prefix-cache"), the
configuration the metric is quoted on; it fits one GPU.  N>1: weak scaling, every rank scores its own
64k-request shard against the replicated snapshot + prefix index, picks are all-gathered.

Batches are independent, so by default two of them are in flight (`--inflight 2`: consecutive launches alternate between two
streams, which hides the dispatch gap between back-to-back kernels); every launch still processes one whole batch and
`roofline.kernel_avg_ms` is the per-launch duration measured in the timed region -- about twice the kernel's duration
with the GPU to itself (`--inflight 1`, or `--alone-ref` for both in one run), because two launches share the GPU.

Contract: `python bench.py --gpus N --steps K --warmup W`; N>1 is launched by torch.distributed.run
(one rank per GPU).  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def cpu_baseline(pkg, wl, orc, passes: int = 2):
    """The C restatement (oracle/) timed on this box's host cores, on the SAME workload.

    Single thread on a 4096-request sample (the shape of the reference's per-request loop) and all
    cores on the whole batch.  Bounded: ~7 core-seconds per full pass at 64k x 4096."""
    oix = orc.OracleIndex()
    oix.insert(wl.index_hashes, wl.index_pods)
    cores = os.cpu_count() or 1
    n1 = min(wl.R, 4096)
    t0 = time.perf_counter()
    p1, _, _ = orc.pick_batch(wl.chain, wl.pods, oix, wl.reqs[:n1], wl.B)
    t1 = time.perf_counter() - t0
    best = None
    for _ in range(passes):
        t0 = time.perf_counter()
        pm, sm, _ = orc.pick_batch(wl.chain, wl.pods, oix, wl.reqs, wl.B, threads=cores)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    assert np.array_equal(p1, pm[:n1])
    return dict(value=wl.R / best, unit="decisions/s", cores=cores, kind="port",
                sample=f"{passes} passes of the full {wl.R} x {wl.P} batch on {cores} threads (C restatement, not Go); "
                       f"single thread on the first {n1} requests",
                single_thread_value=n1 / t1), pm, sm


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", type=int, default=5, help="BASELINE.json config number (1-based); 5 = headline")
    ap.add_argument("--requests", type=int, default=None, help="override requests per rank")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--groups", type=int, default=256, help="shared-prefix groups (256 = BASELINE workload; 65536 = cold-cache index variant)")
    ap.add_argument("--zipf", type=float, default=1.0, help="Zipf exponent over groups (0 = uniform)")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed even at world size 1 (exercises the N>1 code path on one GPU)")
    ap.add_argument("--inflight", type=int, default=2, choices=(1, 2), help="batches in flight: consecutive (independent) batches alternate between this many compute streams; 1 = strictly back-to-back launches on one stream")
    ap.add_argument("--gather-every", type=int, default=4, help="N>1: all-gather the picks of this many batches with one RCCL call (1 = one collective per batch)")
    ap.add_argument("--alone-ref", action="store_true", help="after the timed region also time 50 launches back to back on one stream (the kernel with the GPU to itself) and report them as roofline.kernel_alone_*")
    ap.add_argument("--host-path", type=int, default=40, help="also time N batches through the host-buffer entry point (H2D + kernel + D2H): the p99 pick latency a host caller observes; 0 = skip")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the pick has no CPU path (libeppk fails loudly without HIP)")
    torch.cuda.set_device(local_rank)
    dist = None
    use_dist = world > 1 or args.force_dist
    if use_dist:
        import torch.distributed as dist  # noqa: F811
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    pkg = graft.load_package()
    # replicated snapshot + index (seed of the config), this rank's own request shard (weak scaling)
    wl = pkg.workload.make_workload(args.config, R=args.requests, req_seed=(0x5EED0000 + args.config) ^ (0xA5A5 * rank) if rank else None,
                                    n_groups=args.groups, zipf_s=args.zipf)
    R = wl.R
    pk = pkg.BatchedPicker(wl.chain, max_pods=wl.P, max_blocks=wl.B, max_batch=R, index_slots=wl.index_slots, device=local_rank)
    pk.publish(wl.pods)
    if wl.index_slots:
        pk.index_insert(wl.index_hashes, wl.index_pods)

    dev = torch.device("cuda", local_rank)
    d_reqs = torch.from_numpy(wl.reqs.view(np.int64)).to(dev)
    # A ring of NBUF pick buffers (one contiguous tensor).  The picks are all-gathered in BUCKETS of `--gather-every` batches
    # (default 4): one RCCL call moves the picks of four batches -- a 256 KiB per-rank message is pure latency on xGMI, so
    # fewer, larger collectives it is -- on the `comm` stream, overlapping the kernels of the following batches.  The
    # compute streams wait for the collectives once per trip around the ring.
    ring = pkg.distributed.GatherRing(nbuf=8, gather_every=args.gather_every)   # the bookkeeping (tests/test_distributed_cpu.py)
    NBUF, G = ring.nbuf, ring.gather_every
    d_picks_all = torch.empty(NBUF * R, dtype=torch.int32, device=dev)
    d_picks = [d_picks_all[i * R:(i + 1) * R] for i in range(NBUF)]
    d_scores = [torch.empty(R, dtype=torch.float64, device=dev) for _ in range(NBUF)]
    d_alls = [torch.empty(world * G * R, dtype=torch.int32, device=dev) for _ in range(NBUF // G)] if use_dist else None
    # Explicit streams: the kernel and its HIP-event brackets are ordered on `compute`, the collective on `comm`.
    # (torch's legacy default stream has handle 0, which the C ABI reads as "the context's own stream".)
    # Consecutive batches are independent, so their kernels alternate between TWO compute streams: the next launch is already
    # queued when a kernel drains and its workgroups start as CUs free up (hides most of the ~5 us dispatch gap between
    # back-to-back launches on one stream).  Every kernel still processes one whole batch; kernel time is per launch.
    computes = [torch.cuda.Stream(device=dev) for _ in range(args.inflight)]
    compute = computes[0]
    comm = torch.cuda.Stream(device=dev)
    # torch's CURRENT stream is `comm` (c10d enqueues the collective on the current stream); the pick kernel gets `compute`
    # by handle.  No per-step stream context manager: at 35 us per kernel the host enqueue path is what limits the N > 1 rate.
    torch.cuda.set_stream(comm if use_dist else compute)
    streams = [c.cuda_stream for c in computes]
    assert all(h != 0 for h in streams)
    comm_handle = comm.cuda_stream
    ev_gather = torch.cuda.Event()                           # the all-gather of the last bucket of a trip finished (ring reusable)
    last_gather = [None]                                     # (bucket tensor view, slots in it) of the most recent all-gather

    def gather(due):
        if due is None:
            return
        b0, n, closes_trip = due
        out = d_alls[ring.bucket_of(b0)][: world * n * R]
        dist.all_gather_into_tensor(out, d_picks_all[b0 * R:(b0 + n) * R])       # on `comm`, the current stream
        last_gather[0] = (out, n)
        if closes_trip:
            ev_gather.record(comm)

    p_reqs, p_scores, p_picks = d_reqs.data_ptr(), [t.data_ptr() for t in d_scores], [t.data_ptr() for t in d_picks]

    def step():
        if use_dist and ring.begins_trip():
            for c in computes:
                c.wait_event(ev_gather)                  # every all-gather of the previous trip is done: the ring is free again
        b = ring.next_slot()
        pk.pick_device(p_reqs, R, None, p_picks[b], p_scores[b], streams[b % len(streams)])
        due = ring.after_batch()
        if use_dist:
            pk.stream_wait_pick(comm_handle)                       # comm waits for the kernel's own completion event
            gather(due)

    def fence():
        due = ring.flush()
        if use_dist:
            gather(due)                                  # a partly filled bucket is flushed: every step's picks are gathered inside the timed region
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    pk.profile(True)           # HIP events around every pick launch on the launch stream + probe counts
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kern_ms = pk.profile_drain()
    abytes, lookups, launches = pk.profile_bytes()
    alone_ms = None
    if args.alone_ref and world == 1:
        # reference figure, outside the timed region: the same kernel with the GPU to itself (launches back to back on one stream)
        for _ in range(60):
            pk.pick_device(p_reqs, R, None, p_picks[0], p_scores[0], streams[0])
        torch.cuda.synchronize()
        alone_ms = np.asarray(pk.profile_drain(), dtype=np.float64)[10:]
    pk.profile(False)

    last = (ring.steps - 1) % NBUF
    picks = d_picks[last].cpu().numpy()
    scores = d_scores[last].cpu().numpy()
    if use_dist:
        out, n = last_gather[0]                          # layout [world][n][R]; the last step is the last slot of this rank's slab
        allp = out.cpu().numpy().reshape(world, n, R)
        assert np.array_equal(allp[rank, n - 1], picks), "all-gather returned a different shard"

    if rank == 0:
        out = {
            "metric": "routing decisions/sec, 64k-req x 4096-pod batch" if args.config == 5 and args.requests is None and args.groups == 256 else f"routing decisions/sec ({wl.name}, groups={args.groups}, zipf={args.zipf})",
            "value": world * R * args.steps / elapsed,
            "unit": "decisions/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": wl.name, "requests_per_gpu": R, "pods": wl.P, "adapters": wl.A, "blocks_per_request": wl.B,
                       "chain": "queue:2,kv:2,lora:1,prefix:3" if args.config in (3, 5) else str(wl.chain),
                       "index_entries": int(wl.index_hashes.shape[0]), "sharding": f"requests/{world} per rank, RCCL all-gather of picks (buckets of {G} batches) overlapped with the following kernels" if use_dist else "single GPU",
                       "batches_in_flight": args.inflight,
                       "p99_step_ms": None},
        }
        k = np.asarray(kern_ms, dtype=np.float64)
        per_launch_bytes = abytes / max(launches, 1)
        avg_ms = float(k.mean()) if k.size else float("nan")
        achieved = per_launch_bytes / (avg_ms * 1e-3) / 1e9 if k.size else float("nan")
        traffic = None
        kname = "pick_fast_kernel" if (wl.mask is None) else "pick_generic_kernel"
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")   # measured separately with rocprofv3 --pmc
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("workload") == wl.name and args.groups == 256 and args.zipf == 1.0:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        # What the library's OWN layout moves through L2 -> L1 per launch (an estimate from the probe counts):
        # request rows + pod rows + picks, one 64-byte key bucket per probed hash (all of a request's first 32 are gathered),
        # one 64-byte pod list per hit (the dense 64 * sizeof(LW)-byte row when the lists are off), the adapter tables.
        # `achieved` / `frac` above stay on SURVEY §8(d)'s byte model (u64 key + P/8-byte bitmap per index entry).
        lw_bytes = 2 if wl.P <= 1024 else 4 if wl.P <= 2048 else 8
        stride = 8 + 8 * wl.B
        fixed = wl.P * 64 + R * (stride + 4)
        lk = lookups / max(launches, 1)
        row_model = 8 + 64 * lw_bytes
        hits = max(per_launch_bytes - fixed - 8 * lk, 0.0) / (row_model - 8) if wl.B else 0.0
        lists_on = os.environ.get("EPPK_LISTS", "1") != "0"
        layout_bytes = fixed + (R * min(wl.B, 32) * 64 + hits * (64 if lists_on else 64 * lw_bytes) if wl.B else 0) + R * (16 * 12 + 2 * 64 * lw_bytes)
        out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                           "traffic": traffic, "kernel": "pick_fast_kernel", "kernel_avg_ms": avg_ms,
                           "kernel_p99_ms": float(np.percentile(k, 99)) if k.size else None,
                           "algorithmic_bytes_per_launch": per_launch_bytes, "index_lookups_per_launch": lk,
                           "byte_model": "SURVEY 8(d): u64 key + P/8-byte bitmap per index entry (the reference-shaped index)",
                           "layout_bytes_per_launch": layout_bytes,
                           "layout_GBps": layout_bytes / (avg_ms * 1e-3) / 1e9 if k.size else None,
                           "launches_in_flight": args.inflight}
        if alone_ms is not None and alone_ms.size:
            # with two batches in flight the kernels share the GPU, so each launch lasts about twice as long as it does alone
            # while two of them finish per that time; `achieved` / `frac` above use the duration measured in the timed region
            a = float(alone_ms.mean())
            out["roofline"]["kernel_alone_avg_ms"] = a
            out["roofline"]["kernel_alone_p99_ms"] = float(np.percentile(alone_ms, 99))
            out["roofline"]["frac_alone"] = per_launch_bytes / (a * 1e-3) / 1e9 / HBM_PEAK_GBS
        out["config"]["p99_step_ms"] = out["roofline"]["kernel_p99_ms"]
        if args.host_path and world == 1:
            # host-observed pick latency: request rows in host memory -> pinned staging -> H2D -> kernel -> D2H (PCIe-inclusive;
            # never `value`)
            lat = []
            for _ in range(args.host_path):
                t0 = time.perf_counter()
                pk.pick(wl.reqs)
                lat.append(time.perf_counter() - t0)
            lat = np.asarray(lat[2:] or lat) * 1e3
            out["host_path"] = {"batches": int(lat.size), "p50_ms": float(np.percentile(lat, 50)), "p99_ms": float(np.percentile(lat, 99)),
                                "decisions_per_s_p50": R / (float(np.percentile(lat, 50)) * 1e-3)}
        if world == 1 and not args.no_cpu_baseline:
            orc = graft.load_oracle()
            cb, opicks, oscores = cpu_baseline(pkg, wl, orc)
            out["cpu_baseline"] = cb
            out["parity"] = {"picks_equal_oracle": bool(np.array_equal(picks, opicks)),
                             "scores_bitwise_equal_oracle": bool(np.array_equal(scores.view(np.uint64), oscores.view(np.uint64)))}
        # RCCL writes a version banner through C stdio, which a pipe buffers until exit: push it out first so that the JSON
        # line is the last thing on stdout
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    pk.close()


if __name__ == "__main__":
    main()
