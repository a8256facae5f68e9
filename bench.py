#!/usr/bin/env python3
"""bench.py — routing decisions/s of the batched endpoint pick on MI355X (BASELINE.json metric).

A "step" = one pass of the hot path over one batch of 64k requests: request rows already resident in HBM -> fused pick
kernel -> picks in HBM (+ for N>1 an RCCL all-gather of the per-rank picks).  Workload: BASELINE.json configs[4]
("64k req x 4096 pods, full scorer chain + prefix-cache"), the configuration the metric is quoted on; it fits one GPU.

What is timed (and what is not):
  * the timed region ROTATES through `--batches` (default 16) DISTINCT request batches -- 16 x 17.3 MB = 277 MB, more than the
    256 MB Infinity Cache -- so request rows stream from HBM every step (a single resident batch would be served by the cache);
  * N=1: the whole batch on one GPU.  N>1 (BASELINE.json configs[4] "request-sharded 8xMI355X with RCCL all-gather of picks"):
    the headline is WEAK scaling -- every rank scores a whole 64k batch per step (the shard size the metric is quoted on; N x 64k
    requests per step), the picks of all ranks are all-gathered in buckets of 16 batches so that every rank holds all of them;
    the same invocation then also times STRONG scaling (each step's ONE 64k batch split R/N per rank, a rank's shards of a
    bucket scored by one launch) and prints it beside (`"strong": {...}`).  `--scaling strong` makes that the headline instead.
    (Until round 3 strong was the headline: a strong-scaled step is 2 us of kernel per rank at 8 GPUs, so the driver's 20-step
    timed region -- 0.4 ms on one GPU -- would time one collective's latency and a fence: profiles/r03_y_short_runs.txt.)
  * the closing fence reads the clock when this rank's device is idle (every step's picks gathered), BEFORE the barrier; the
    region's time is the MAX of that over the ranks;
  * `--closed-loop`: pick -> eppk_index_insert_picks_device (the post-route index update, SEMANTICS.md §6) -> next, DIFFERENT
    batch, with ageing every `--age-every` steps; its first generations are checked against the oracle at full size.
  * batches are independent, so two of them are in flight (`--inflight 2`: consecutive launches alternate between two streams);
    the closed loop is strictly ordered (one stream).

Roofline objects of the JSON line:
  roofline        the headline workload, as the contract defines it: bound "hbm", achieved = COMPULSORY HBM bytes of a launch under the
                  library's own layout (request rows + outputs + every distinct index line / table the launch touches, once) / the
                  kernel's average duration (HIP events on its dispatch packet), frac = achieved / 8 TB/s -- about 0.08: the headline
                  index (4 096 distinct keys) lives in L2, so the launch is NOT HBM-bound.  `traffic` = exact HBM bytes per launch from
                  the stamped PMC passes (profiles/pmc_traffic.json; null when the stamp does not match the sources).  Beside it:
                  `vmem_pipe` (what does bound it: the CU's texture-address unit busy / kernel clocks -- a unit-busy ratio, labelled as
                  such), `issue` (instruction counts per decision), `l2_side_*` (what the layout requests from L2), and `model_frac`
                  (SURVEY §8(d)'s byte model -- u64 key + P/8-byte bitmap per index entry -- for reference only: the kernel does not
                  move those bytes, so it may exceed 1).
  roofline_cold   the same kernel on a COLD index (262 144 prefix groups, uniform: 4.2 M distinct hashes), where HBM is the bound:
                  bytes = what the layout reads per launch (rows, one 64-byte bucket per gathered hash, one 64-byte list per hit);
                  `frac_strict` = SURVEY §8(d) to the letter (probes = matched + 1).
  closed_loop, roofline_closed_loop, host_path.pipelined_learn
                  the path's steady state -- pick -> the index learns the picks -> next batch, ageing every other step -- in a context
                  of its own, its first generations verified against the oracle at full size (closed_loop_leg).

Contract: `python bench.py --gpus N --steps K --warmup W`; N>1 is launched by torch.distributed.run (one rank per GPU).
Rank 0 prints ONE JSON line (the last line on stdout).
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

DBG = os.environ.get("EPPK_BENCH_DBG", "")      # measurement switches of the N > 1 path (nolat / nogather; with or without --pack16): timing experiments only (scripts/gpu_r3_z4.sh)
HOSTTIME = os.environ.get("EPPK_BENCH_HOSTTIME", "0") == "1"   # stderr: where the host's time of a timed region goes (N > 1 path)
HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)
N_SIMD = 256 * 4          # 256 CUs x 4 SIMD16; a wave64 VALU instruction occupies its SIMD for 4 cycles


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def kernel_source_hash() -> str:
    csrc = os.path.join(graft.PKG_DIR, "csrc")
    files = [os.path.join(csrc, f) for f in ("eppk_kernels.hip.h", "eppk_pick_inst.hip.h", "eppk.hip")] + [os.path.join(ROOT, "include", "eppk.h")]
    return graft._digest(files, " ".join(graft.COMPILE_FLAGS))[:16]


def stamped_json(name: str, khash: str):
    """profiles/<name> if it carries this build's kernel-source hash, else None."""
    path = os.path.join(ROOT, "profiles", name)
    try:
        with open(path) as f:
            j = json.load(f)
        return j if j.get("kernel_src_sha16") == khash else None
    except Exception:
        return None


def cpu_baseline(wl, orc, reqs, all_batches=None, passes: int = 3):
    """The CPU side of the comparison, on this box's host cores and the SAME workload (C restatement under oracle/, not Go).

    `value`: the batch algorithm a careful CPU implementation would use (oracle.c: orc_pick_batch_sparse — per-snapshot tables,
    then per request only the pods its prefix walk names), all rotating batches in one call so that thread start-up is amortised,
    best of `passes` and of {all, half} of the hardware threads.  `per_request_loop_value`: the shape of the reference's
    per-request Schedule() (every scorer over every candidate: O(R x P)), all cores on one batch; `single_thread_*` beside both.
    The two algorithms must agree bit for bit on the whole batch (asserted); the loop's picks are what `parity` compares with.
    Bounded: ~7 core-seconds for the O(R x P) pass at 64k x 4096, ~0.5 core-seconds per pass of the table algorithm."""
    oix = orc.OracleIndex()
    oix.insert(wl.index_hashes, wl.index_pods)
    cores = os.cpu_count() or 1
    n1 = min(wl.R, 4096)
    t0 = time.perf_counter()
    p1, _, _ = orc.pick_batch(wl.chain, wl.pods, oix, reqs[:n1], wl.B)
    t1 = time.perf_counter() - t0
    t0 = time.perf_counter()
    pm, sm, _ = orc.pick_batch(wl.chain, wl.pods, oix, reqs, wl.B, threads=cores)
    t_loop = time.perf_counter() - t0
    assert np.array_equal(p1, pm[:n1])
    loop = dict(per_request_loop_value=wl.R / t_loop, per_request_loop_cores=cores, per_request_loop_single_thread_value=n1 / t1,
                per_request_loop_what=f"every scorer over every candidate per request (the shape of the reference's Schedule()): one {wl.R} x {wl.P} "
                                      f"batch on {cores} threads; single thread on its first {n1} requests")
    try:
        # the table algorithm: snapshot tables once (as the GPU's snapshot kernels are outside its timed region), then the batches
        t0 = time.perf_counter()
        tb = orc.OracleTables(wl.chain, wl.pods)
        ps, ss = tb.pick_batch(oix, reqs, wl.B, threads=cores)
        t_tables = time.perf_counter() - t0
        assert np.array_equal(ps, pm) and np.array_equal(ss.view(np.uint64), sm.view(np.uint64)), "the two CPU algorithms disagree"
        big = np.concatenate(all_batches) if all_batches is not None and len(all_batches) > 1 else reqs
        best, best_threads = None, cores
        for th in sorted({cores, max(1, cores // 2)}, reverse=True):
            for _ in range(passes):
                t0 = time.perf_counter()
                tb.pick_batch(oix, big, wl.B, threads=th)
                dt = time.perf_counter() - t0
                if best is None or dt < best:
                    best, best_threads = dt, th
        t0 = time.perf_counter()
        tb.pick_batch(oix, reqs, wl.B, threads=1)
        t_s1 = time.perf_counter() - t0
        return dict(value=big.shape[0] / best, unit="decisions/s", cores=best_threads, kind="port",
                    algorithm="snapshot tables + the pods each request's prefix walk names (oracle.c orc_pick_batch_sparse; bit-identical "
                              "to the per-request loop on this batch)",
                    sample=f"best of {passes} passes over {big.shape[0]} requests ({max(1, big.shape[0] // wl.R)} batches of {wl.R} x {wl.P}) in one call, "
                           f"on {cores} and {max(1, cores // 2)} threads (C restatement, not Go); tables built once before ({t_tables * 1e3:.1f} ms incl. one batch)",
                    single_thread_value=wl.R / t_s1,
                    **loop), pm, sm
    except Exception as e:          # the bench line must not die with its baseline leg: fall back to the loop's figure and say so
        return dict(value=wl.R / t_loop, unit="decisions/s", cores=cores, kind="port", algorithm="per-request loop (the table algorithm failed)",
                    sample=loop["per_request_loop_what"], single_thread_value=n1 / t1, error=f"{type(e).__name__}: {e}", **loop), pm, sm


def metric_name(headline: bool, mode: str, world: int, wl, args) -> str:
    """BASELINE.json's metric on one GPU; at N > 1 the name says which scaling `value` is (so that an N x 64k-per-step aggregate cannot be
    read as configs[4]'s one 64k batch, or the other way round)."""
    if not headline:
        return f"routing decisions/sec ({wl.name}, groups={args.groups}, zipf={args.zipf}{', closed loop' if args.closed_loop else ''})"
    if mode == "weak":
        return f"routing decisions/sec, {world} x 64k-req x 4096-pod batches per step (request-sharded replicas, one whole batch per rank: weak scaling)"
    if mode == "strong":
        return f"routing decisions/sec, ONE 64k-req x 4096-pod batch per step split over {world} ranks (strong scaling)"
    return "routing decisions/sec, 64k-req x 4096-pod batch"


def make_batches(pkg, wl, args, n: int):
    """`n` distinct request batches against ONE snapshot + index (batch 0 = the workload's own requests)."""
    out = [wl.reqs]
    base = (0x5EED0000 + args.config)
    for b in range(1, n):
        out.append(pkg.workload.make_requests(wl, base ^ (0x9E3779B1 * b & 0x7FFFFFFF)))
    return out


class Runner:
    """One context + NB resident request batches + the stream / ring plumbing of a timed region."""

    def __init__(self, pkg, torch, dist, wl, batches, args, rank, world, local_rank, index_slots=None, closed_loop=False):
        self.pkg, self.torch, self.dist, self.wl, self.args = pkg, torch, dist, wl, args
        self.rank, self.world = rank, world
        self.use_dist = dist is not None
        self.R, self.NB = wl.R, len(batches)
        self.h_batches = batches
        slots = wl.index_slots if index_slots is None else index_slots
        # strong scaling scores a rank's shards of a whole gather bucket with ONE launch (setup()): up to gather_every x R/world rows
        per_strong = (wl.R + world - 1) // world
        max_batch = max(wl.R, args.gather_every * per_strong) if self.use_dist else wl.R
        self.pk = pkg.BatchedPicker(wl.chain, max_pods=wl.P, max_blocks=wl.B, max_batch=max_batch, index_slots=slots, device=local_rank)
        self.pk.publish(wl.pods)
        if slots:
            self.pk.index_insert(wl.index_hashes, wl.index_pods)
        self.dev = dev = torch.device("cuda", local_rank)
        self.stride = 8 + 8 * wl.B
        self.d_batches = [torch.from_numpy(b.view(np.int64)).to(dev) for b in batches]
        self.p_batches = [t.data_ptr() for t in self.d_batches]
        inflight = 1 if closed_loop else args.inflight
        self.computes = [torch.cuda.Stream(device=dev) for _ in range(inflight)]
        self.comm = torch.cuda.Stream(device=dev)
        # torch's CURRENT stream is `comm` (c10d enqueues the collective on the current stream); the pick kernel gets a compute
        # stream by handle.  No per-step stream context manager: the host enqueue path is what limits the N > 1 rate.
        torch.cuda.set_stream(self.comm if self.use_dist else self.computes[0])
        self.streams = [c.cuda_stream for c in self.computes]
        assert all(h != 0 for h in self.streams)
        self.comm_handle = self.comm.cuda_stream
        self.closed_loop = closed_loop
        self.learn_api = hasattr(self.pk, "pick_learn_device") and not getattr(args, "cl_two_calls", False)
        self.profile_every = max(1, int(getattr(args, "profile_every", 1)))
        # N > 1: completion latency of a gather bucket -- a timing event on the compute stream right before the first launch that scores
        # a bucket's batches, one on `comm` behind the collective that delivers their picks to every rank (timed region only)
        self.lat_on, self.lat_start, self.lat_pairs, self.coll_pairs = False, None, [], []
        self.host_t = {}
        self.ev_pool, self.ev_pool_used = [], 0

    # -- a timed region ---------------------------------------------------------------------------------------------------
    def setup(self, mode: str, gather_every: int):
        """mode: 'single' (N=1), 'strong' (the step's batch split R/N per rank), 'weak' (a whole batch per rank)."""
        torch, R, W = self.torch, self.R, self.world
        self.mode = mode
        self.per = (R + W - 1) // W if mode == "strong" else R          # requests this rank scores per step
        self.lo = min(self.rank * self.per, R) if mode == "strong" else 0
        self.n_mine = max(0, min(self.lo + self.per, R) - self.lo) if mode == "strong" else R
        # (strong scaling with launch groups: four buckets, so that launches run ahead of the collectives that free their buckets)
        want_groups = (mode == "strong" and self.use_dist and not self.closed_loop and gather_every > 1 and self.NB % gather_every == 0
                       and self.n_mine > 0 and not getattr(self.args, "no_launch_groups", False))
        # Four gather buckets in the ring: a launch waits only for the collective that last read ITS bucket (ev_bucket), so launches run
        # up to three buckets ahead of the collectives.  (The ring used to be one trip of 8 / 16 slots that every compute stream
        # re-entered only behind the LAST collective of the previous trip: a pipeline bubble per trip -- 35 us per step instead of 21 in a
        # 20-step run of the weak mode on one GPU, gpurun_out/r3y.)
        self.ring = self.pkg.distributed.GatherRing(nbuf=4 * gather_every if self.use_dist else 8, gather_every=gather_every)
        NBUF, G = self.ring.nbuf, self.ring.gather_every
        # weak scaling gathers WHOLE batches (world x G x R picks per bucket: 32 MiB of int32 at 8 GPUs and G = 16).  `--pack16` sends pod
        # indices below 2^15 as int16 (EPPK_NO_PICK = -1 stays -1), which halves what the collective moves over xGMI -- OFF by default:
        # the cast is one more small kernel on the collective stream, and on one GPU every such kernel between the persistent pick
        # kernels opened a 25-50 us hole in their timeline (gpurun_out/r3z4_gaps.txt: 20-step region 437 us with the cast, 389 without)
        self.pack16 = self.use_dist and mode == "weak" and self.wl.P <= 32767 and bool(getattr(self.args, "pack16", False))
        self.d_picks_all = torch.full((NBUF * self.per,), -1, dtype=torch.int32, device=self.dev)
        self.d_picks = [self.d_picks_all[i * self.per:(i + 1) * self.per] for i in range(NBUF)]
        self.d_scores_all = torch.empty(NBUF * self.per, dtype=torch.float64, device=self.dev)
        self.d_scores = [self.d_scores_all[i * self.per:(i + 1) * self.per] for i in range(NBUF)]
        # Strong scaling: a rank's shard of one batch (R / world rows: 8192 at 8 GPUs) is a launch-bound kernel, and the host's enqueue
        # cost per launch then caps the rate.  The batches are independent and a bucket of G of them is all-gathered by one collective
        # anyway, so the rank keeps ITS rows of every batch contiguous (d_shards[b]) and scores the G shards of a bucket with ONE
        # launch over G x R/world contiguous rows (8 x 8192 = one 64k launch at 8 GPUs) right before the bucket's collective.
        self.grouped = want_groups and G == gather_every
        self.ev_bucket = [None] * (NBUF // G)        # grouped: the event behind the collective that last read bucket k (its slots are free again)
        if self.grouped:
            rows = self.stride // 8
            self.d_shards = torch.empty((self.NB, self.per, rows), dtype=torch.int64, device=self.dev)
            for b, t in enumerate(self.d_batches):
                v = t.view(-1, rows)
                self.d_shards[b, : self.n_mine] = v[self.lo:self.lo + self.n_mine]
                if self.n_mine < self.per:                       # a ragged last shard: valid filler rows (their picks are never looked at)
                    self.d_shards[b, self.n_mine:] = v[self.lo:self.lo + 1]
            self.p_shards = self.d_shards.data_ptr()
            # one full-size launch per compute stream now: the library sizes a stream's work-list buffer at its first launch of that size
            # (a hipMalloc behind a stream synchronize), and with 16 steps per launch the second stream's first launch would otherwise
            # fall into the timed region
            for c in self.computes:
                self.pk.pick_device(self.p_shards, G * self.per, None, self.d_picks[0].data_ptr(), self.d_scores[0].data_ptr(), c.cuda_stream)
            torch.cuda.synchronize()
        self.launch_requests = (G * self.per) if self.grouped else self.n_mine
        self.p_picks = [t.data_ptr() for t in self.d_picks]
        self.p_scores = [t.data_ptr() for t in self.d_scores]
        pdt = torch.int16 if self.pack16 else torch.int32
        self.d_alls = [torch.empty(W * G * self.per, dtype=pdt, device=self.dev) for _ in range(NBUF // G)] if self.use_dist else None
        self.d_p16 = [torch.empty(G * self.per, dtype=torch.int16, device=self.dev) for _ in range(NBUF // G)] if self.pack16 else None
        self.gather_views = {}
        self.last_gather = None
        self.step_no = 0
        self.last_batch = 0

    def _gather(self, due):
        if due is None:
            return
        if HOSTTIME:
            t_h = time.perf_counter()
            self._gather_body(due)
            self.host_t["gather"] = self.host_t.get("gather", 0.0) + time.perf_counter() - t_h
            self.host_t["gathers"] = self.host_t.get("gathers", 0) + 1
            return
        self._gather_body(due)

    def _gather_body(self, due):
        b0, n, closes_trip = due
        k = self.ring.bucket_of(b0)
        views = self.gather_views.get((b0, n))           # (slicing and re-viewing five tensors costs the host ~15 us per collective)
        if views is None:
            out = self.d_alls[k][: self.world * n * self.per]
            src = self.d_picks_all[b0 * self.per:(b0 + n) * self.per]
            p16 = self.d_p16[k][: n * self.per] if self.pack16 else None
            views = self.gather_views[(b0, n)] = (out, src, p16, out.view(self.torch.uint8) if self.pack16 else None,
                                                  p16.view(self.torch.uint8) if self.pack16 else None)
        out, src, p16, out8, p16_8 = views
        e_c0 = None
        if self.lat_on:                                  # the collective alone: events around it on `comm` (timed region only)
            e_c0 = self._timing_event()
            e_c0.record(self.comm)
        if self.pack16:                                  # (on `comm`, the current stream, behind the kernels' completion events)
            p16.copy_(src)
            if DBG != "nogather":
                self.dist.all_gather_into_tensor(out8, p16_8)   # (as bytes: gloo has no int16)
        elif DBG != "nogather":
            self.dist.all_gather_into_tensor(out, src)  # on `comm`, the current stream
        if e_c0 is not None:
            e_c1 = self._timing_event()
            e_c1.record(self.comm)
            self.coll_pairs.append((e_c0, e_c1))
        self.last_gather = (out, n)
        if self.lat_on and self.lat_start is not None:
            e_end = self._timing_event()
            e_end.record(self.comm)
            self.lat_pairs.append((self.lat_start, e_end, n))
            self.lat_start = None
        if self.ev_bucket[k] is None:
            self.ev_bucket[k] = self.torch.cuda.Event()
        self.ev_bucket[k].record(self.comm)              # bucket k's slots may be overwritten behind this

    def _timing_event(self):
        """Timing events for the bucket latencies come from a pool created outside the timed region (creating one costs the host ~5 us)."""
        if self.ev_pool_used == len(self.ev_pool):
            self.ev_pool.append(self.torch.cuda.Event(enable_timing=True))
        e = self.ev_pool[self.ev_pool_used]
        self.ev_pool_used += 1
        return e

    def batch_of(self, step: int) -> int:
        # weak scaling: ranks walk the same ring of batches at different offsets, so that no two ranks score the same batch at
        # the same step (every rank holds all NB batches)
        return (step + (self.rank if self.mode == "weak" else 0)) % self.NB

    def step(self):
        ring = self.ring
        if self.grouped:
            # Launch groups: 15 of 16 steps only advance the ring (the rank's shard of the batch is scored by the bucket's ONE launch), so
            # this path is kept to a handful of bytecodes -- at 8 GPUs a step is 2-3 us of device time and the host loop is what bounds
            # the rate (measured on one GPU with --force-dist: 3.4 us per step through the general path below, whatever the batch size).
            slot = ring.steps % ring.nbuf
            b = self.step_no % self.NB                        # (strong scaling: every rank walks the ring of batches in step)
            self.step_no += 1
            self.last_batch, self.last_slot = b, slot         # (fence() flushes a partly filled bucket from here)
            if (slot + 1) % ring.gather_every:
                ring.steps += 1
                ring._count += 1
                return
            due = ring.after_batch()
            self._launch_group(due, b)
            if self.n_mine:
                self.pk.stream_wait_pick(self.comm_handle)   # comm waits for the kernel's own completion event
            self._gather(due)
            return
        slot = ring.next_slot()
        if self.use_dist and slot % ring.gather_every == 0:
            ev = self.ev_bucket[slot // ring.gather_every]
            if ev is not None:
                for c in self.computes:
                    c.wait_event(ev)                         # the collective that read this bucket on its previous trip is done
        b = self.batch_of(self.step_no)
        st = self.streams[slot % len(self.streams)]
        if self.n_mine and not self.grouped:
            if self.lat_on and self.use_dist and self.lat_start is None:      # first batch of a bucket
                self.lat_start = self._timing_event()
                self.lat_start.record(self.computes[slot % len(self.computes)])
            if self.closed_loop and self.learn_api:           # pick + post-route index update in one call: the next pick sees the update
                self.pk.pick_learn_device(self.p_batches[b] + self.lo * self.stride, self.n_mine, None, self.p_picks[slot], self.p_scores[slot], st)
            else:
                self.pk.pick_device(self.p_batches[b] + self.lo * self.stride, self.n_mine, None, self.p_picks[slot], self.p_scores[slot], st)
                if self.closed_loop:                          # (a picker without the fused entry point: the two calls, same stream)
                    self.pk.index_insert_picks_device(self.p_batches[b] + self.lo * self.stride, self.p_picks[slot], self.n_mine, st)
        due = ring.after_batch()
        if self.grouped and due is not None:
            self._launch_group(due, b)
        if self.use_dist:
            # comm waits for the completion event of the LAST launch on every compute stream before a bucket's collective: the last
            # `inflight` launches of the bucket (consecutive launches alternate between the streams; each stream runs its own in order).
            # (A wait per launch was 2 HIP calls = ~4 us of host time per step on a path whose host side is what bounds a short run.)
            if self.n_mine and (due is not None or (slot % ring.gather_every) >= ring.gather_every - len(self.streams)):
                self.pk.stream_wait_pick(self.comm_handle)
            self._gather(due)
        self.last_batch, self.last_slot = b, slot
        self.step_no += 1

    def _launch_group(self, due, last_b):
        """One launch over this rank's shards of the `n` batches of a bucket (slots first .. first + n - 1; batches last_b - n + 1 ..
        last_b: buckets never wrap the ring, and NB is a multiple of the bucket size, so both ranges are contiguous)."""
        first, n, _ = due
        b0 = last_b - n + 1
        assert b0 >= 0 and b0 + n <= self.NB
        k = self.ring.bucket_of(first)
        cs = self.computes[k % len(self.computes)]
        st = cs.cuda_stream
        if self.ev_bucket[k] is not None:
            cs.wait_event(self.ev_bucket[k])                 # the collective that read this bucket on its previous trip is done
        if self.lat_on and self.lat_start is None:
            self.lat_start = self._timing_event()
            self.lat_start.record(cs)
        self.pk.pick_device(self.p_shards + b0 * self.per * self.stride, n * self.per, None, self.p_picks[first], self.p_scores[first], st)

    def fence(self, t0=None):
        """Flush + device synchronize (+ barrier).  With `t0` (the closing fence of a timed region) returns this rank's elapsed time, read
        when ITS device is idle -- every step's picks gathered -- and BEFORE the barrier: the region's time is the MAX over ranks of
        these (timed()), so the barrier's own cost (a device all-reduce plus a host round trip, tens of microseconds against a
        20-step region of 0.4 ms) is not charged to the steps; it still separates the region from whatever follows."""
        due = self.ring.flush()
        if self.grouped and due is not None:
            self._launch_group(due, self.last_batch)
            self.pk.stream_wait_pick(self.comm_handle)
        if self.use_dist:
            if due is not None and not self.grouped:
                for c in self.computes:
                    self.comm.wait_stream(c)                 # (a partly filled bucket: its last launches may sit on either compute stream)
            self._gather(due)                                # a partly filled bucket is flushed: every step's picks are gathered inside the timed region
        # (Polling the streams with hipStreamQuery ahead of this synchronize -- so that the host stands at the fence when the last kernel
        # ends instead of waking up from an interrupt -- was measured and dropped: the polling itself slows the launches down, 21.7 vs
        # 20.2 us per step at --steps 20.  The GPU's own span of that region is 361 of its 402 us: profiles/r04_trace20_timeline.txt.)
        t_sync = time.perf_counter()
        self.torch.cuda.synchronize()
        if HOSTTIME and t0 is not None:
            self.host_t["synchronize"] = time.perf_counter() - t_sync
        elapsed = None if t0 is None else time.perf_counter() - t0
        if self.use_dist:
            self.dist.barrier()
            self.torch.cuda.synchronize()
        return elapsed

    def timed(self, steps: int, warmup: int, age=None):
        """`warmup` untimed steps, then exactly `steps` timed ones between fences; returns (elapsed max over ranks, kernel ms list,
        (bytes, lookups, launches))."""
        for i in range(warmup):
            self.step()
            if age:
                age(i)
        # HIP events + probe counts on every `profile_every`-th pick launch only: events on every launch cost the host ~3 us per step
        # (21.4 vs 18.1 us against the C harness on the same library, profiles/r02_micro_pickbench.txt) -- instrumentation is not
        # part of what the headline times.  The p99 samples (more_kernel_samples, outside the timed region) instrument every launch.
        self.pk.profile(self.profile_every)
        self.fence()
        self.lat_on, self.lat_start, self.lat_pairs, self.coll_pairs = self.use_dist and DBG != "nolat", None, [], []
        if self.use_dist:
            while len(self.ev_pool) < 128:
                self.ev_pool.append(self.torch.cuda.Event(enable_timing=True))
        self.ev_pool_used = 0
        self.host_t = {}
        t0 = time.perf_counter()
        for i in range(steps):
            self.step()
            if age:
                age(warmup + i)
        if HOSTTIME:
            self.host_t["issue_all_steps"] = time.perf_counter() - t0
        elapsed = self.fence(t0)
        if HOSTTIME:
            log(f"[bench] host time, mode {self.mode}: " + ", ".join(f"{k} {v * 1e6:.1f} us" if isinstance(v, float) else f"{k} {v}" for k, v in self.host_t.items())
                + f", region {elapsed * 1e6:.1f} us")
        self.lat_on = False
        self.bucket_latency_ms = np.asarray([a.elapsed_time(b) for a, b, _ in self.lat_pairs], dtype=np.float64)
        self.bucket_batches = [n for _, _, n in self.lat_pairs]
        self.collective_ms = np.asarray([a.elapsed_time(b) for a, b in self.coll_pairs], dtype=np.float64)
        if self.use_dist:
            t = self.torch.tensor([elapsed], dtype=self.torch.float64, device=self.dev)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            elapsed = float(t.item())
        kern_ms = np.asarray(self.pk.profile_drain(), dtype=np.float64)
        stats = self.pk.profile_bytes()
        return elapsed, kern_ms, stats

    def more_kernel_samples(self, have: int, want: int):
        """Launch durations beyond the timed region (same launch pattern, profiling still on) until `want` samples exist."""
        out = []
        if have < want:
            self.pk.profile(True)      # every launch from here on (resets the probe statistics: timed() has read them already)
        while have + sum(len(x) for x in out) < want:
            n = min(256, want - have - sum(len(x) for x in out))
            for _ in range(n):
                self.step()
            self.fence()
            out.append(np.asarray(self.pk.profile_drain(), dtype=np.float64))
        return np.concatenate(out) if out else np.zeros(0)

    def last_outputs(self):
        picks = self.d_picks[self.last_slot].cpu().numpy()[: self.n_mine]
        scores = self.d_scores[self.last_slot].cpu().numpy()[: self.n_mine]
        return picks, scores

    def check_gather(self):
        """The most recent all-gather handed this rank its own shard back, and (strong) a full batch."""
        if not self.use_dist or self.last_gather is None:
            return None
        out, n = self.last_gather
        allp = out.cpu().numpy().reshape(self.world, n, self.per)
        mine = self.d_picks[self.last_slot].cpu().numpy()
        assert np.array_equal(allp[self.rank, n - 1], mine), "all-gather returned a different shard"
        # strong: the whole batch in request order; weak: [world, R] -- row r = the picks of the batch rank r scored in the last step
        return allp[:, n - 1, :].reshape(-1)[: self.R] if self.mode == "strong" else allp[:, n - 1, :].astype(np.int32)

    def close(self):
        self.pk.close()


WEAK_BUCKET = 16               # weak scaling: batches per all-gather (16 x 64k picks per rank as int16 = 2 MiB per rank and collective)

L2_GATHER_PEAK_GBS = 16900.0   # measured ceiling of 64-byte line gathers out of an L2-resident table, one quad per line (the quad kernel's
                               # access pattern): 0.515 lines / clock / CU (profiles/r02_g_micro_l2gather.txt)


def byte_models(wl, R, kern_stats, khash, lists_on=True, quad=False):
    """Per-launch byte counts from the device-counted probe statistics: (SURVEY model, layout L2-side bytes, compulsory HBM bytes)."""
    abytes, lookups, launches = kern_stats
    launches = max(launches, 1)
    model = abytes / launches
    lk = lookups / launches
    lw_bytes = 2 if wl.P <= 1024 else 4 if wl.P <= 2048 else 8
    stride = 8 + 8 * wl.B
    fixed = wl.P * 64 + R * (stride + 4)
    row_model = 8 + 64 * lw_bytes
    hits = max(model - fixed - 8 * lk, 0.0) / (row_model - 8) if wl.B else 0.0
    out_bytes = R * 12                                                         # i32 pick + f64 score
    tables = 129 * 64 * (12 + 2 * lw_bytes) + wl.P * 8 + (wl.B + 1) ** 2 * 8   # per-adapter tables + base[] + pterm (read through L2 by every workgroup)
    # L2-side: what the layout requests per launch (all of a request's first 32 key buckets are gathered, one list or row per hit)
    per_hit = 64 if lists_on else 64 * lw_bytes
    n_sets = int((wl.meta or {}).get("n_groups", 0)) or 1          # distinct pod sets of the pre-populated index: one per prefix group
    with_hits = R if hits > 0 else 0                              # (every request of the bench workloads finds its group's shared blocks)
    if quad:      # pick_quad_kernel (protocol v5): 17 key buckets gathered ahead (the rest only behind 17 hits) -- a hit's set id comes with its
        # bucket line -- ONE 64-byte set line per request, one 64-byte line of interleaved tier planes per listed pod (~8), 16 table entries
        probed = min(wl.B, 17) if hits < 17 * R else min(wl.B, 32)
        l2_side = R * stride + out_bytes + (R * probed * 64 + with_hits * 64 if wl.B else 0) + R * (16 * 12 + 8 * 64)
    else:
        l2_side = R * stride + out_bytes + (R * min(wl.B, 32) * 64 + hits * per_hit if wl.B else 0) + R * (16 * 12 + 2 * 64 * lw_bytes)
    # compulsory HBM: streams (rows in, picks/scores out) + every DISTINCT index line once + the tables once
    n_keys = int(np.unique(wl.index_hashes).size) if wl.B else 0
    distinct_buckets = min(R * min(wl.B, 32), wl.index_slots // 4) if wl.B else 0          # (a bucket per four API slots)
    distinct_sets = min(with_hits, n_sets) if quad else min(hits, n_keys)
    compulsory = R * stride + out_bytes + distinct_buckets * 64 + distinct_sets * per_hit + tables
    # an index far beyond the caches: every gathered bucket comes from HBM (and, counted here although a 16 MiB set table mostly stays in
    # the Infinity Cache, the request's one set line); the adapter tables do not
    probed_cold = (min(wl.B, 17) if hits < 17 * R else min(wl.B, 32)) if quad else min(wl.B, 32)
    cold_hbm = R * stride + out_bytes + (R * probed_cold * 64 + (with_hits * 64 if quad else hits * per_hit) if wl.B else 0)
    # SURVEY 8(d) strictly: only the probes the sequential walk needs (matched + 1 per request, device-counted) and the pod-set lines the
    # layout needs for them (v5: one per request; the fast kernel: one per hit)
    cold_strict = R * stride + out_bytes + (lk * 64 + (with_hits * 64 if quad else hits * per_hit) if wl.B else 0)
    return dict(model=model, lookups=lk, hits=hits, l2_side=l2_side, compulsory=compulsory, cold_hbm=cold_hbm, cold_strict=cold_strict)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", type=int, default=5, help="BASELINE.json config number (1-based); 5 = headline")
    ap.add_argument("--requests", type=int, default=None, help="override requests per batch")
    ap.add_argument("--batches", type=int, default=16, help="distinct request batches the timed region rotates through (16 x 17.3 MB > the 256 MB Infinity Cache)")
    ap.add_argument("--scaling", choices=("strong", "weak", "both"), default="both", help="N>1: 'both' = weak scaling is the headline `value`, strong scaling is timed too and printed beside it")
    ap.add_argument("--group", type=int, default=0, metavar="MEMBERS",
                    help="ONE process, the C-ABI device group (eppk_group_*) instead of torch.distributed: MEMBERS contexts over the visible GPUs (round "
                         "robin; all on device 0 of a one-GPU box), device-resident shards, one launch per member and gather bucket, peer all-gather")
    ap.add_argument("--closed-loop", action="store_true", help="pick -> insert_picks -> next different batch, ageing every --age-every steps (N=1; one stream)")
    ap.add_argument("--age-every", type=int, default=2, help="closed loop: tick the index epoch and evict every this many steps")
    ap.add_argument("--keep-epochs", type=int, default=2, help="closed loop: hashes not re-inserted during this many epochs are evicted")
    ap.add_argument("--cl-slots", type=int, default=1 << 24, help="closed loop: index slots (live keys ~ age_every * keep_epochs * R * B/2)")
    ap.add_argument("--cl-verify", type=int, default=3, help="closed loop: generations checked against the oracle at full size before timing")
    ap.add_argument("--cl-two-calls", action="store_true", help="closed loop: eppk_pick_batch_device + eppk_index_insert_picks_device instead of eppk_pick_learn_device")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-resident-leg", action="store_true", help="skip the small-batch latencies through the resident workgroup (EPPK_RESIDENT=1)")
    ap.add_argument("--no-closed-loop-leg", action="store_true", help="skip the closed-loop / pipelined-LEARN sub-run of the default line")
    ap.add_argument("--closed-loop-leg", action="store_true", help="run that sub-run for a non-headline workload too")
    ap.add_argument("--cl-steps", type=int, default=200, help="timed steps of the default line's closed-loop sub-run (warm-up: a quarter of it)")
    ap.add_argument("--pl-batches", type=int, default=512, help="batches of the default line's pipelined-LEARN sub-run (its p99 is a percentile of that many)")
    ap.add_argument("--no-cold-ref", action="store_true", help="skip the cold-index sub-run (roofline_cold)")
    ap.add_argument("--no-revisit-leg", action="store_true", help="skip the returning-requests sub-run (`revisit`)")
    ap.add_argument("--revisit-leg", action="store_true", help="run that sub-run for a non-headline workload too")
    ap.add_argument("--groups", type=int, default=256, help="shared-prefix groups (256 = BASELINE workload; 65536 = cold index)")
    ap.add_argument("--zipf", type=float, default=1.0, help="Zipf exponent over groups (0 = uniform)")
    ap.add_argument("--pods-per-group", type=int, default=8, help="pods that hold each group's shared blocks in the pre-populated index")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed even at world size 1 (exercises the N>1 code path on one GPU)")
    ap.add_argument("--inflight", type=int, default=2, choices=(1, 2, 3, 4), help="batches in flight: consecutive (independent) batches alternate between this many compute streams")
    ap.add_argument("--no-launch-groups", action="store_true", help="N>1 strong scaling: one launch per batch shard instead of one per gather bucket")
    ap.add_argument("--gather-every", type=int, default=16, help="N>1: all-gather the picks of this many batches with one RCCL call (1 = one collective per batch)")
    ap.add_argument("--pack16", action="store_true", help="N>1 weak scaling: all-gather the picks as int16 (one cast kernel per bucket; see Runner.setup)")
    ap.add_argument("--profile-every", type=int, default=8, help="inside the timed region only every Nth pick launch carries HIP events and probe counters (1 = all)")
    ap.add_argument("--p99-samples", type=int, default=1000, help="kernel durations collected for the p99 (beyond the timed region if it has fewer launches)")
    ap.add_argument("--host-path", type=int, default=1000, help="also time N batches through the host-buffer entry point (H2D + kernel + D2H): the pick latency a host caller observes; 0 = skip")
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
    if args.group:
        return group_leg(graft.load_package(), torch, args)
    if args.closed_loop and use_dist:
        raise SystemExit("--closed-loop is a one-GPU mode (the group API applies the gathered update on every device: tests/test_gpu_group.py)")

    pkg = graft.load_package()
    khash = kernel_source_hash()
    t_gen = time.perf_counter()
    # replicated snapshot + index; the SAME batches on every rank (strong: a rank scores its slice of each; weak: ranks walk the
    # ring of batches at different offsets)
    wl = pkg.workload.make_workload(args.config, R=args.requests, n_groups=args.groups, zipf_s=args.zipf, pods_per_group=args.pods_per_group)
    R = wl.R
    batches = make_batches(pkg, wl, args, max(1, args.batches))
    log(f"[bench] rank {rank}: {len(batches)} batches of {R} requests generated in {time.perf_counter() - t_gen:.1f} s")
    headline = args.config == 5 and args.requests is None and args.groups == 256 and args.zipf == 1.0 and args.pods_per_group == 8 and not args.closed_loop

    run = Runner(pkg, torch, dist, wl, batches, args, rank, world, local_rank,
                 index_slots=(args.cl_slots if args.closed_loop else None), closed_loop=args.closed_loop)
    # N > 1: WEAK scaling is the headline (every rank scores a whole 64k batch per step -- the shard the metric is quoted on -- and the
    # picks of all ranks are all-gathered); strong scaling (the step's ONE 64k batch split R/N) is timed in the same invocation and
    # printed beside it.  Why: a strong-scaled step is 2 us of kernel per rank at 8 GPUs, so a short timed region (the driver's
    # --steps 20: 0.4 ms on one GPU, 50 us at ideal 8-GPU strong scaling) measures one collective's latency and one fence, not the
    # path (one-GPU dry run of both at --steps 20 in profiles/r03_y_short_runs.txt).
    modes = ["single"] if not use_dist else (["weak", "strong"] if args.scaling == "both" else [args.scaling])
    results = {}
    cl_info = cl_state = None
    for mode in modes:
        run.setup(mode, args.gather_every if mode != "weak" else min(args.gather_every, WEAK_BUCKET))
        age = None
        if args.closed_loop:
            cl_info = closed_loop_verify(run, wl, args)
            state = cl_state = {"epoch": cl_info["epoch"]}

            def age(i, state=state):      # stream-ordered: behind the inserts of this step, ahead of the next pick
                if (i + 1) % args.age_every == 0:
                    state["epoch"] = run.pk.index_advance_epoch()
                    if state["epoch"] > args.keep_epochs:
                        run.pk.index_evict_older_device(state["epoch"] - args.keep_epochs + 1, run.streams[0])
        elapsed, kern_ms, stats = run.timed(args.steps, args.warmup, age)
        results[mode] = dict(elapsed=elapsed, kern_ms=kern_ms, stats=stats, per=run.per, launch_requests=run.launch_requests, grouped=run.grouped,
                             bucket_latency_ms=run.bucket_latency_ms, bucket_batches=run.bucket_batches, collective_ms=run.collective_ms,
                             bucket=run.ring.gather_every, pack16=run.pack16,
                             value=(world if mode == "weak" else 1) * R * args.steps / elapsed, ms_per_step=1e3 * elapsed / args.steps)
        if mode == modes[0]:
            extra_ms = run.more_kernel_samples(len(kern_ms), args.p99_samples) if not args.closed_loop else np.zeros(0)
            picks, scores = run.last_outputs()
            last_batch, lo, n_mine = run.last_batch, run.lo, run.n_mine
            gathered = run.check_gather()
            last_step = run.step_no - 1
        else:
            run.check_gather()
    run.pk.profile(False)
    status = run.pk.launch_status()
    assert status == 0, f"launch status {status}"

    if rank == 0:
        main_mode = modes[0]
        res = results[main_mode]
        G = res["bucket"]
        sharding = {"single": "single GPU",
                    "strong": (f"each 64k batch split R/{world} per rank, RCCL all-gather of picks (buckets of {G} batches) overlapped with the following kernels" +
                               (f"; a rank scores its {G} shards of a bucket with ONE launch ({G} x {run.per} contiguous rows)" if res.get("grouped") else "")),
                    "weak": (f"one whole batch per rank per step ({world} x {R} requests per step), RCCL all-gather of all ranks' picks (buckets of {G} batches" +
                             (", int16 payload" if res.get("pack16") else "") + ") overlapped with the following kernels")}[main_mode]
        out = {
            "metric": metric_name(headline, main_mode, world if use_dist else 1, wl, args),
            "value": res["value"],
            "unit": "decisions/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": res["ms_per_step"],
            "higher_is_better": True,
            "scaling": "weak" if main_mode in ("single", "weak") else "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": wl.name, "requests_per_step": R if main_mode != "weak" else R * world, "requests_per_gpu": res["per"], "pods": wl.P,
                       "adapters": wl.A, "blocks_per_request": wl.B,
                       "chain": "queue:2,kv:2,lora:1,prefix:3" if args.config in (3, 5) else str(wl.chain),
                       "index_entries": int(wl.index_hashes.shape[0]), "sharding": sharding,
                       "distinct_batches": len(batches), "batch_bytes_resident": int(len(batches) * R * run.stride),
                       "batches_in_flight": len(run.streams), "closed_loop": bool(args.closed_loop), "p99_step_ms": None},
        }
        if "weak" in results and main_mode != "weak":
            w = results["weak"]
            out["weak"] = {"value": w["value"], "ms_per_step": w["ms_per_step"], "requests_per_gpu": R,
                           "note": "weak scaling timed in the same invocation: every rank scores one whole batch per step, picks all-gathered"}
        if "strong" in results and main_mode != "strong":
            st = results["strong"]
            out["strong"] = {"value": st["value"], "ms_per_step": st["ms_per_step"], "requests_per_gpu": st["per"], "requests_per_step": R,
                             "requests_per_launch": int(st["launch_requests"]), "batches_per_bucket": int(st["bucket"]),
                             "note": "strong scaling timed in the same invocation: each step's ONE batch split R/N per rank, picks all-gathered per bucket" +
                                     ("; a rank scores its shards of a whole bucket with ONE launch" if st.get("grouped") else "")}
            sbl = st["bucket_latency_ms"]
            if sbl.size:
                out["strong"]["completion_latency_p50_ms"] = float(np.percentile(sbl, 50))
                out["strong"]["completion_latency_p99_ms"] = float(np.percentile(sbl, 99))
        if use_dist:
            out["config"]["ranks_seen"] = int(dist.get_world_size())
            cm = res.get("collective_ms")
            out["config"]["per_rank_kernel_us"] = None      # (filled in below, once the launch durations are reduced)
            out["config"]["collective_us"] = ({"p50": float(np.percentile(cm, 50)) * 1e3, "max": float(cm.max()) * 1e3, "collectives": int(cm.size),
                                               "what": "one all-gather of a bucket's picks, events around it on the collective stream (rank 0, timed region)"}
                                              if cm is not None and cm.size else None)
            bl = res["bucket_latency_ms"]
            if bl.size:
                # BASELINE's metric names p99 pick latency: at N > 1 a batch's picks exist on every rank when the collective of its
                # gather bucket has finished -- with launch groups that is the whole bucket's launch + one all-gather later
                out["completion_latency"] = {"p50_ms": float(np.percentile(bl, 50)), "p99_ms": float(np.percentile(bl, 99)), "max_ms": float(bl.max()),
                                             "buckets": int(bl.size), "batches_per_bucket": int(max(res["bucket_batches"])),
                                             "definition": "device time from the start of the (first) launch that scores a gather bucket's batches on this rank to the end of the "
                                                           "all-gather that hands their picks to every rank (events on the compute and the collective stream, rank 0, timed region)"}
        k_timed = res["kern_ms"]
        k_all = np.concatenate([k_timed, extra_ms]) if extra_ms.size else k_timed
        # launch duration: the sampled launches of the timed region plus the every-launch samples taken behind it (same launch pattern)
        avg_ms = float(k_all.mean()) if k_all.size else float("nan")
        q_launches, q_deferred = run.pk.quad_stats()
        quad = q_launches > 0
        bm = byte_models(wl, res["launch_requests"], res["stats"], khash, lists_on=os.environ.get("EPPK_LISTS", "1") != "0", quad=quad)
        out["config"]["requests_per_launch"] = int(res["launch_requests"])
        kname = ("pick_quad_kernel" if quad else "pick_fast_kernel" if run.pk.chain_is_fused() else "pick_generic_kernel")
        out["config"]["quad_route"] = {"launches": q_launches, "requests_deferred_to_pick_fast_kernel": q_deferred,
                                       "note": "four requests per wavefront; a launch = pick_quad_kernel + the work-list pass of pick_fast_kernel over what it deferred (both inside kernel_*_ms)"}
        achieved = bm["compulsory"] / (avg_ms * 1e-3) / 1e9
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                "kernel": kname, "kernel_avg_ms": avg_ms,
                "kernel_p50_ms": float(np.percentile(k_all, 50)) if k_all.size else None,
                "kernel_p99_ms": float(np.percentile(k_all, 99)) if k_all.size else None,
                "kernel_samples": int(k_all.size), "launches_in_flight": len(run.streams),
                "bytes_per_launch": bm["compulsory"],
                "bytes_definition": "compulsory HBM bytes of one launch under libeppk's layout: request rows in + picks/scores out + each distinct key bucket / pod list / table touched, once",
                "l2_side_bytes_per_launch": bm["l2_side"], "l2_side_GBps": bm["l2_side"] / (avg_ms * 1e-3) / 1e9,
                "index_lookups_per_launch": bm["lookups"],
                "l2_frac_of_gather_ceiling": bm["l2_side"] / (avg_ms * 1e-3) / 1e9 / L2_GATHER_PEAK_GBS,
                "limiter": ("not HBM: the L2-resident index is gathered line by line -- vector memory pipe (one access per 16-byte piece and clock and CU, "
                            "an L2 line per two clocks and CU: DESIGN.md 3.1) plus VALU issue of the per-request skeleton; see `issue`; the HBM-bound case is `roofline_cold`"),
                "model_bytes_per_launch": bm["model"], "model_frac": bm["model"] / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "model_note": "SURVEY 8(d) byte model (u64 key + P/8-byte bitmap per index entry, the reference-shaped index): NOT what this kernel moves; reference figure only, may exceed 1",
                "kernel_src_sha16": khash}
        tj = stamped_json("pmc_traffic.json", khash)
        if tj and tj.get("workload") == wl.name and headline:
            roof["traffic"] = tj.get("hbm_bytes_per_launch")
            roof["traffic_frac"] = roof["traffic"] / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
            roof["traffic_source"] = ("profiles/pmc_traffic.json: rocprofv3 --pmc passes of THIS kernel build (stamped with its source hash), exact bytes = "
                                      "64 * RDREQ_64B + 128 * RDREQ_128B + WRITE_SIZE (every HBM read of this GPU is a 128-byte request; FETCH_SIZE tallies it at 64: profiles/r02_fetchcal.txt)")
        ij = stamped_json("pmc_issue.json", khash)
        if ij and headline:
            roof["issue"] = {k: ij[k] for k in ij if k not in ("kernel_src_sha16",)}
            if "valu_per_decision" in ij:      # wave-instructions issued per second / (SIMDs x clock / 4)
                clk = ij.get("sclk_hz", 2.4e9)
                # VALU wave-instructions per second over the whole step / (SIMDs x clock / 4 cycles per wave64 instruction)
                roof["issue"]["valu_issue_frac_live"] = ij["valu_per_decision"] * res["per"] / (res["ms_per_step"] * 1e-3) / (N_SIMD * clk / 4.0)
        # the same bytes over the time of one STEP (with two launches in flight a launch lasts about twice as long as it does alone,
        # while two of them finish per that time): the rate the whole GPU sustains
        roof["step_GBps"] = bm["compulsory"] / (res["ms_per_step"] * 1e-3) / 1e9
        roof["step_frac"] = roof["step_GBps"] / HBM_PEAK_GBS
        # The primary object is what the contract asks for: bound "hbm", achieved = the launch's algorithmic (compulsory) HBM bytes over the
        # kernel's duration, frac = achieved / 8 TB/s -- 0.08 for the headline, and it says so.  WHY it is that low sits beside it under
        # `vmem_pipe`: the headline index (4 096 distinct keys) is L2-resident by construction, so this launch is bound by the CU's vector
        # memory pipe (texture-address unit busy clocks over kernel clocks from the stamped PMC pass of this build, or -- while no stamp
        # matches the sources -- the L2-side gather rate over its measured ceiling), not by HBM; the HBM-bound variant of the kernel
        # is `roofline_cold` (a different, cold index: not the BASELINE workload).
        if headline or quad:
            roof["hbm_achieved_GBps"], roof["hbm_peak_GBps"], roof["hbm_frac"] = roof["achieved"], roof["peak"], roof["frac"]      # (aliases kept for round-3 readers)
            iss = roof.get("issue") or {}
            ta = (iss["ta_busy_cycles_avg"] / iss["kernel_cycles_tcc_busy_avg"]) if iss.get("ta_busy_cycles_avg") and iss.get("kernel_cycles_tcc_busy_avg") else None
            roof["vmem_pipe"] = ({"what": "texture-address unit busy clocks / kernel clocks (TA_BUSY_avr / TCC_BUSY_avr, profiles/pmc_issue.json): a unit-busy ratio, NOT bytes / time / peak",
                                  "busy_clocks": iss["ta_busy_cycles_avg"], "kernel_clocks": iss["kernel_cycles_tcc_busy_avg"], "busy_frac": ta} if ta is not None else
                                 {"what": "L2-side 64-byte line gathers vs their measured ceiling (no stamped PMC pass for this build)",
                                  "l2_side_GBps": roof["l2_side_GBps"], "ceiling_GBps": L2_GATHER_PEAK_GBS, "busy_frac": roof["l2_frac_of_gather_ceiling"]})
        out["roofline"] = roof
        out["config"]["p99_step_ms"] = roof["kernel_p99_ms"]
        if use_dist:
            out["config"]["per_rank_kernel_us"] = avg_ms * 1e3
            # what `value` means at N > 1, in north_star's own terms
            out["scaling_note"] = ("north_star shards the request batch across GPUs \"only when |requests| x |pods| outgrows one device\": rank 0 scores a launch of "
                                   f"{int(res['launch_requests'])} requests x {wl.P} pods in {avg_ms * 1e3:.1f} us here, so the 64k x 4096 batch does not outgrow one MI355X.  " +
                                   ("`value` is therefore the aggregate of N request-sharded replicas -- every rank scores a whole 64k batch per step and all picks are "
                                    "all-gathered (weak scaling: the unit per GPU is the shard the metric is quoted on); the strong-scaled single batch (BASELINE.json "
                                    "configs[4]: ONE 64k batch split R/N per rank) is timed in the same invocation and printed under `strong`."
                                    if main_mode == "weak" else
                                    "`value` is the strong-scaled single batch (BASELINE.json configs[4]: ONE 64k batch split R/N per rank, picks all-gathered); a step is "
                                    "then a few microseconds of kernel per rank and the collective's latency bounds it; weak scaling is printed under `weak` when timed."))
        if cl_info:
            out["closed_loop"] = cl_info
            out["roofline_closed_loop"] = closed_loop_roofline(run, wl, args, cl_state, res["ms_per_step"])
        if args.host_path and world == 1 and not args.closed_loop:
            # host-observed pick latency: request rows in host memory -> pinned staging -> H2D -> kernel -> D2H (PCIe-inclusive;
            # never `value`)
            lat = []
            for i in range(args.host_path + 2):
                t0 = time.perf_counter()
                run.pk.pick(batches[i % len(batches)])
                lat.append(time.perf_counter() - t0)
            lat = np.asarray(lat[2:] or lat) * 1e3
            out["host_path"] = {"batches": int(lat.size), "p50_ms": float(np.percentile(lat, 50)), "p99_ms": float(np.percentile(lat, 99)),
                                "decisions_per_s_p50": R / (float(np.percentile(lat, 50)) * 1e-3),
                                "what": "eppk_pick_batch on pageable caller rows: chunked copy into pinned staging overlapped with the H2D DMA, row check + pick kernel writing the pinned results"}
            if hasattr(run.pk, "staging"):
                # the same with the rows BUILT in the library's pinned staging buffer (eppk_host_staging / eppk_pick_batch_staged): what a
                # dispatcher that writes its request rows straight into that buffer sees; the fill is the caller's row construction, not timed
                st_reqs, _ = run.pk.staging()
                lat2 = []
                for i in range(args.host_path + 2):
                    np.copyto(st_reqs[:R], batches[i % len(batches)])
                    t0 = time.perf_counter()
                    run.pk.pick_staged(R)
                    lat2.append(time.perf_counter() - t0)
                lat2 = np.asarray(lat2[2:] or lat2) * 1e3
                out["host_path"]["staged"] = {"p50_ms": float(np.percentile(lat2, 50)), "p99_ms": float(np.percentile(lat2, 99)),
                                              "decisions_per_s_p50": R / (float(np.percentile(lat2, 50)) * 1e-3),
                                              "what": "eppk_pick_batch_staged: rows already in the pinned staging buffer: one H2D, row check on the device, pick kernel writing the pinned results"}
            if hasattr(run.pk, "staging"):
                # BASELINE's metric names the p99 pick latency: what a dispatcher that drains a few dozen to a few thousand pending requests
                # per call observes -- one batch of n requests through eppk_pick_batch_staged, fresh rows written into the pinned buffer
                # before every call (not timed).  Up to 3072 requests the library runs these zero-copy (one launch, no upload / download); above, one upload and no download (the kernel writes the pinned results).
                by_n = {}
                for n in (16, 128, 2048, 8192):
                    if n > R:
                        continue
                    l4 = []
                    o_p, o_s = np.empty(n, dtype=np.int32), np.empty(n, dtype=np.float64)     # (result arrays and their addresses once: the timed
                    a_p, a_s = o_p.ctypes.data, o_s.ctypes.data                               #  call is the ctypes call alone, as a cgo caller's would be)
                    for i in range(min(args.host_path, 200) + 10):
                        off = (i * n) % max(1, R - n + 1)
                        np.copyto(st_reqs[:n], batches[i % len(batches)][off:off + n])
                        t0 = time.perf_counter()
                        run.pk.pick_staged_into(n, a_p, a_s)
                        l4.append(time.perf_counter() - t0)
                    l4 = np.asarray(l4[10:] or l4) * 1e6
                    by_n[str(n)] = {"p50_us": float(np.percentile(l4, 50)), "p99_us": float(np.percentile(l4, 99))}
                out["host_path"]["latency_by_batch"] = {"requests": by_n, "what": "eppk_pick_batch_staged, one batch at a time, host-observed (call -> picks and scores in "
                                                        "caller memory); zero-copy up to EPPK_ZERO_COPY_MAX (default 3072) requests"}
            if hasattr(run.pk, "resident_stats") and not args.no_resident_leg:
                # the same small batches through the RESIDENT workgroup (EPPK_RESIDENT=1, opt-in: include/eppk.h): no launch, no completion
                # signal -- a doorbell in pinned memory and a polling host.  A context of its own (the switch is read at eppk_create).
                try:
                    res_leg = resident_latency_leg(pkg, wl, batches, min(args.host_path, 200))
                    out["host_path"]["latency_by_batch_resident"] = res_leg
                    # latency_by_batch[n] LEADS with the library's default configuration (p50_us / p99_us, path "launched"); the opt-in
                    # resident workgroup's figures for the sizes it serves stand beside them (resident_p50_us / resident_p99_us)
                    lbb = out["host_path"].get("latency_by_batch", {})
                    lb = lbb.get("requests", {})
                    for v in lb.values():
                        v["path"] = "launched (default)"
                    if res_leg.get("picks_and_scores_equal_oracle") and res_leg.get("batches_answered_by_the_resident_workgroup", 0) > 0:
                        for n_s, v in res_leg.get("requests", {}).items():
                            e = lb.setdefault(n_s, {"path": "resident only (size not timed on the launched path)"})
                            e["resident_p50_us"], e["resident_p99_us"] = v["p50_us"], v["p99_us"]
                        lbb["what"] = (lbb.get("what", "") + "; resident_p50_us / resident_p99_us: the same call with EPPK_RESIDENT=1 (opt-in latency path: a resident "
                                       "workgroup behind a doorbell, no launch; include/eppk.h)")
                except Exception as e:
                    out["host_path"]["latency_by_batch_resident"] = {"error": repr(e)}
            if hasattr(run.pk, "resident_stats") and not args.no_resident_leg:
                try:
                    out["host_path"]["latency_dispatcher_calls"] = dispatcher_latency_leg(pkg, wl, batches, min(args.host_path, 200))
                except Exception as e:
                    out["host_path"]["latency_dispatcher_calls"] = {"error": repr(e)}
            if hasattr(run.pk, "stage_begin"):
                # PIPELINED: two staging sets -- the rows of batch k + 1 cross PCIe while batch k is scored (eppk_pick_stage_*).  Same
                # convention as `staged`: the rows are in the pinned buffers already (two different batches, one per set; building them
                # is the caller's per-request work, timed separately below as one numpy copy per batch).  Throughput = batches over the
                # wall time of the whole loop; latency = begin -> end of a batch while the other set's upload shares the link.
                sb = [run.pk.stage_buffers(0)[0], run.pk.stage_buffers(1)[0]]
                np.copyto(sb[0][:R], batches[0])
                t0 = time.perf_counter()
                np.copyto(sb[1][:R], batches[1 % len(batches)])
                fill_ms = (time.perf_counter() - t0) * 1e3
                nb_p = max(args.host_path, 8) + 2
                lat3, t_begin = [], [0.0, 0.0]
                t_begin[0] = time.perf_counter(); run.pk.stage_begin(0, R)
                t_loop = None
                for i in range(1, nb_p + 1):
                    cur, prev = i & 1, (i - 1) & 1
                    if i == 2:
                        t_loop = time.perf_counter()
                    if i < nb_p:
                        t_begin[cur] = time.perf_counter()
                        run.pk.stage_begin(cur, R)
                    p_picks, _ = run.pk.stage_end(prev)
                    lat3.append(time.perf_counter() - t_begin[prev])
                t_all = time.perf_counter() - t_loop
                lat3 = np.asarray(lat3[2:]) * 1e3
                out["host_path"]["pipelined"] = {"batches": int(lat3.size), "decisions_per_s": R * (nb_p - 1) / t_all, "ms_per_batch": 1e3 * t_all / (nb_p - 1),
                                                 "p50_ms": float(np.percentile(lat3, 50)), "p99_ms": float(np.percentile(lat3, 99)),
                                                 "pcie_floor_ms": R * run.stride / 55e9 * 1e3, "caller_fill_ms_numpy_one_thread": fill_ms,
                                                 "what": "eppk_pick_stage_begin / _end over two staging sets, rows already in the pinned sets: validate + upload of one batch under "
                                                         "the kernel and download of the other; latency = begin -> end of a batch; the floor is the 17 MB of rows at ~55 GB/s of PCIe"}
        if world == 1 and not args.no_cpu_baseline and not args.closed_loop:
            orc = graft.load_oracle()
            cb, opicks, oscores = cpu_baseline(wl, orc, batches[last_batch], batches)
            out["cpu_baseline"] = cb
            out["parity"] = {"picks_equal_oracle": bool(np.array_equal(picks, opicks[lo:lo + n_mine])),
                             "scores_bitwise_equal_oracle": bool(np.array_equal(scores.view(np.uint64), oscores[lo:lo + n_mine].view(np.uint64))),
                             "batch": int(last_batch)}
        elif use_dist and gathered is not None and not args.no_cpu_baseline and R <= 4096:
            orc = graft.load_oracle()
            oix = orc.OracleIndex()
            oix.insert(wl.index_hashes, wl.index_pods)
            if main_mode == "strong":
                op, _, _ = orc.pick_batch(wl.chain, wl.pods, oix, batches[last_batch], wl.B)
                out["parity"] = {"gathered_picks_equal_oracle": bool(np.array_equal(gathered, op))}
            else:       # weak: rank r scored batch (step + r) % NB in the last step; rank 0 holds everybody's picks
                ok = True
                for r in range(world):
                    op, _, _ = orc.pick_batch(wl.chain, wl.pods, oix, batches[(last_step + r) % len(batches)], wl.B)
                    ok = ok and bool(np.array_equal(gathered[r][:R], op))
                out["parity"] = {"gathered_picks_equal_oracle": ok, "ranks_checked": world}
    run.close()

    # The path's STEADY STATE in the line the driver runs (rank 0, N = 1, headline runs): pick -> the index learns the picks -> next,
    # different batch, ageing every other step -- its first generations verified against the oracle at full size -- and the same loop
    # through the pipelined host entry points (EPPK_PICK_LEARN).  A context of its own (16 Mi index slots); ~3 s.
    if rank == 0 and world == 1 and (headline or args.closed_loop_leg) and not args.closed_loop and not use_dist and not args.no_closed_loop_leg:
        try:
            cl = closed_loop_leg(pkg, torch, args, wl, batches)
            out["closed_loop"] = cl["closed_loop"]
            out["roofline_closed_loop"] = cl["roofline_closed_loop"]
            if "host_path" in out and cl.get("pipelined_learn"):
                out["host_path"]["pipelined_learn"] = cl["pipelined_learn"]
        except Exception as e:  # never lose the headline line to a sub-run
            out["closed_loop"] = {"error": repr(e)}

    # roofline_cold: the same kernel on an index that does not fit the caches (rank 0, N=1, headline runs only)
    if rank == 0 and world == 1 and headline and not args.no_cold_ref and not use_dist:
        try:
            out["roofline_cold"] = cold_reference(pkg, torch, args, khash)
        except Exception as e:  # never lose the headline line to the reference sub-run
            out["roofline_cold"] = {"error": repr(e)}

    # revisit: batches of RETURNING requests (rank 0, N = 1, headline runs)
    if rank == 0 and world == 1 and (headline or args.revisit_leg) and not args.closed_loop and not use_dist and not args.no_revisit_leg and wl.B:
        try:
            out["revisit"] = revisit_leg(pkg, torch, args, wl, batches)
        except Exception as e:  # never lose the headline line to a sub-run
            out["revisit"] = {"error": repr(e)}

    if rank == 0:
        steady_state_into_config(out)
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


CONFIG_MAX_KEYS = 20     # the driver's stored record keeps about this many SCALAR `config` keys and no nested objects (BENCH_r05.json: 22)

# `config` of the printed line, most valuable first: (key, path into the line).  The first CONFIG_MAX_KEYS that exist are kept, in this
# order; everything bench.py knows about the run stays in `config_detail` (and in the side objects of the line).
CONFIG_KEYS = [
    ("workload", ("config_detail", "workload")),
    ("requests_per_step", ("config_detail", "requests_per_step")),
    ("pods", ("config_detail", "pods")),
    ("blocks_per_request", ("config_detail", "blocks_per_request")),
    ("chain", ("config_detail", "chain")),
    # N > 1 (absent at N = 1): BASELINE.json configs[4] -- ONE 64k batch split R/N per rank -- beside the weak-scaled `value`
    ("strong_value", ("strong", "value")),
    ("strong_ms_per_step", ("strong", "ms_per_step")),
    ("ranks_seen", ("config_detail", "ranks_seen")),
    ("collective_us", ("config_detail", "collective_us", "p50")),
    ("per_rank_kernel_us", ("config_detail", "per_rank_kernel_us")),
    ("requests_per_gpu_strong", ("strong", "requests_per_gpu")),
    ("strong_completion_latency_p99_ms", ("strong", "completion_latency_p99_ms")),
    ("completion_latency_p99_ms", ("completion_latency", "p99_ms")),
    ("weak_value", ("weak", "value")),
    # N = 1: the path's steady state and what a host caller observes
    ("closed_loop_value", ("closed_loop", "value")),
    ("closed_loop_ms_per_step", ("closed_loop", "ms_per_step")),
    ("closed_loop_picks_equal_oracle", ("closed_loop", "picks_equal_oracle")),
    ("host_staged_p99_ms", ("host_path", "staged", "p99_ms")),
    ("host_pipelined_decisions_per_s", ("host_path", "pipelined", "decisions_per_s")),
    ("host_pipelined_learn_decisions_per_s", ("host_path", "pipelined_learn", "decisions_per_s")),
    ("latency_16_pick_launched_p50_us", ("host_path", "latency_dispatcher_calls", "launched", "pick", "p50_us")),
    ("latency_16_pick_learn_resident_p50_us", ("host_path", "latency_dispatcher_calls", "resident", "pick_learn", "p50_us")),
    ("cold_value", ("roofline_cold", "value")),
    ("cold_frac_strict", ("roofline_cold", "frac_strict")),
    ("revisit_50_value", ("revisit", "by_fraction", "0.5", "value")),
    ("revisit_50_equal_oracle", ("revisit", "by_fraction", "0.5", "picks_and_scores_equal_oracle")),
    ("parity_picks_equal_oracle", ("parity", "picks_equal_oracle")),
    ("parity_scores_bitwise_equal_oracle", ("parity", "scores_bitwise_equal_oracle")),
    ("cpu_baseline_value", ("cpu_baseline", "value")),
    ("closed_loop_scores_bitwise_equal_oracle", ("closed_loop", "scores_bitwise_equal_oracle")),
    ("p99_step_ms", ("config_detail", "p99_step_ms")),
    ("requests_per_launch", ("config_detail", "requests_per_launch")),
    ("sharding", ("config_detail", "sharding")),
    ("distinct_batches", ("config_detail", "distinct_batches")),
    ("closed_loop", ("config_detail", "closed_loop")),
    ("requests_per_gpu", ("config_detail", "requests_per_gpu")),
]


def steady_state_into_config(out) -> None:
    """`config` as the driver's record can hold it: at most CONFIG_MAX_KEYS scalar keys, the path's steady-state, host-path, latency, cold
    and returning-request figures in front (CONFIG_KEYS).  What `config` held so far moves to `config_detail` unchanged."""
    out["config_detail"] = out["config"]
    cfg = {}
    for key, path in CONFIG_KEYS:
        v = out
        for k in path:
            v = v.get(k) if isinstance(v, dict) else None
            if v is None:
                break
        if v is None or isinstance(v, (dict, list, tuple)):
            continue
        cfg[key] = v
        if len(cfg) == CONFIG_MAX_KEYS:
            break
    out["config"] = cfg


def group_leg(pkg, torch, args):
    """`--group M`: the strong-scaling step through the C ABI's device group -- what a Go host links -- in ONE process: M member
    contexts (one per visible GPU, round robin), every member holding ITS rows of the 16 rotating batches; a step scores one batch,
    a bucket of `--gather-every` steps is ONE eppk_group_pick_device call (one launch per member + the peer all-gather of the picks).
    Prints one JSON line (not the driver's headline: `n_gpus` is the number of distinct devices used)."""
    M = args.group
    n_dev = max(1, torch.cuda.device_count())
    devices = [i % n_dev for i in range(M)]
    wl = pkg.workload.make_workload(args.config, R=args.requests, n_groups=args.groups, zipf_s=args.zipf, pods_per_group=args.pods_per_group)
    batches = make_batches(pkg, wl, args, args.batches)
    R, NB, G = wl.R, len(batches), max(1, args.gather_every)
    assert NB % G == 0
    per = (R + M - 1) // M
    g = pkg.DeviceGroup(wl.chain, devices, max_pods=max(wl.P, 64), max_blocks=wl.B, max_batch=max(R, G * per), index_slots=wl.index_slots, gather=pkg.picker.GATHER_PEER)
    g.publish(wl.pods)
    g.index_insert(wl.index_hashes, wl.index_pods)
    rows_w = 1 + wl.B
    shards, n_rows = [], []
    for i in range(M):
        lo, hi = min(i * per, R), min((i + 1) * per, R)
        with torch.cuda.device(devices[i]):
            t = torch.from_numpy(np.concatenate([b[lo:hi] for b in batches]).view(np.int64)).cuda()      # [NB * n_i, rows_w]
        shards.append(t)
        n_rows.append(hi - lo)
    tot = sum(n_rows) * G
    d_picks = [[torch.empty(G * n_rows[i], dtype=torch.int32, device=f"cuda:{devices[i]}") for i in range(M)] for _ in range(2)]
    d_all = [[torch.empty(tot, dtype=torch.int32, device=f"cuda:{devices[i]}") for i in range(M)] for _ in range(2)]

    def bucket(k, slot):                       # batches k*G .. k*G + G - 1 of the ring
        b0 = (k * G) % NB
        g.pick_device([shards[i].data_ptr() + b0 * n_rows[i] * rows_w * 8 for i in range(M)], [G * n for n in n_rows],
                      [t.data_ptr() for t in d_picks[slot]], None, [t.data_ptr() for t in d_all[slot]])

    n_buckets_w, n_buckets = max(1, args.warmup // G), max(1, args.steps // G)
    for k in range(n_buckets_w):
        bucket(k, k & 1)
    g.sync()
    lat = []
    t0 = time.perf_counter()
    for k in range(n_buckets):
        bucket(n_buckets_w + k, k & 1)
    g.sync()
    elapsed = time.perf_counter() - t0
    for k in range(min(n_buckets, 50)):        # completion latency of a bucket, one at a time
        t1 = time.perf_counter()
        bucket(k, 0)
        g.sync()
        lat.append(time.perf_counter() - t1)
    # parity: the gathered picks of the last bucket's first batch against the oracle (member 0's copy; every member holds the same)
    orc = graft.load_oracle()
    oix = orc.OracleIndex()
    oix.insert(wl.index_hashes, wl.index_pods)
    k_last = min(n_buckets, 50) - 1
    b_first = (k_last * G) % NB
    op, _, _ = orc.pick_batch(wl.chain, wl.pods, oix, batches[b_first], wl.B, threads=os.cpu_count() or 1)
    allp = d_all[0][0].cpu().numpy()
    got, off = [], 0
    for i in range(M):
        got.append(allp[off:off + n_rows[i]])              # member i's rows of the bucket's FIRST batch come first in its shard
        off += G * n_rows[i]
    same = bool(np.array_equal(np.concatenate(got), op))
    same_everywhere = all(bool(torch.equal(d_all[0][0].cpu(), d_all[0][i].cpu())) for i in range(M))
    steps = n_buckets * G
    lat = np.asarray(lat) * 1e3
    out = {"metric": f"routing decisions/sec ({wl.name}), C-ABI device group", "value": R * steps / elapsed, "unit": "decisions/s", "n_gpus": len(set(devices)),
           "steps": steps, "warmup": n_buckets_w * G, "ms_per_step": 1e3 * elapsed / steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
           "dtype": "f64", "data": "synthetic",
           "config": {"workload": wl.name, "members": M, "devices": devices, "ranks_seen": int(g.ranks_seen), "requests_per_step": R, "requests_per_member": per,
                      "batches_per_launch": G, "gather": "peer copies (hipMemcpyPeerAsync; device-to-device copies between members that share a GPU)",
                      "entry_point": "eppk_group_pick_device + eppk_group_sync (device-resident shards, no host staging)"},
           "completion_latency": {"p50_ms": float(np.percentile(lat, 50)), "p99_ms": float(np.percentile(lat, 99)), "batches_per_bucket": G,
                                  "definition": "host time of one bucket alone: eppk_group_pick_device -> eppk_group_sync"},
           "parity": {"gathered_picks_equal_oracle": same, "every_member_holds_the_same_picks": same_everywhere}}
    g.close()
    try:
        out["host_path"] = group_pipelined_leg(pkg, wl, batches, devices, args)
        out["config"]["group_pipelined_decisions_per_s"] = out["host_path"]["pipelined"]["decisions_per_s"]
        out["config"]["group_pipelined_learn_decisions_per_s"] = out["host_path"]["pipelined_learn"]["decisions_per_s"]
        out["config"]["group_pipelined_learn_first_batches_equal_oracle"] = out["host_path"]["pipelined_learn"]["first_batches_equal_oracle"]
    except Exception as e:
        out["host_path"] = {"error": repr(e)}
    print(json.dumps(out), flush=True)


def group_pipelined_leg(pkg, wl, batches, devices, args, n_batches: int = 200):
    """What a host caller of a device group sees: eppk_group_pick_stage_begin / _end over the group's two staging sets (rows already in the
    pinned sets), plain and with EPPK_PICK_LEARN + the shim's ageing between begins (eppk_group_index_evict_older_device).  The first
    LEARN batches are checked against the oracle replaying the call order.  A group of its own (closed-loop index size)."""
    R, M = wl.R, len(devices)
    orc = graft.load_oracle()
    g = pkg.DeviceGroup(wl.chain, devices, max_pods=max(wl.P, 64), max_blocks=wl.B, max_batch=R, index_slots=args.cl_slots, gather=pkg.picker.GATHER_PEER)
    try:
        g.publish(wl.pods)
        g.index_insert(wl.index_hashes, wl.index_pods)
        sb = [g.stage_buffers(0)[0], g.stage_buffers(1)[0]]
        np.copyto(sb[0][:R], batches[0])
        np.copyto(sb[1][:R], batches[1 % len(batches)])

        def loop(n, learn, tick=None):
            lat, t_begin = [], [0.0, 0.0]
            t0 = time.perf_counter()
            t_begin[0] = t0
            g.stage_begin(0, R, learn=learn)
            for i in range(1, n + 1):
                cur, prev = i & 1, (i - 1) & 1
                if i < n:
                    t_begin[cur] = time.perf_counter()
                    g.stage_begin(cur, R, learn=learn)
                    if tick is not None:
                        tick(i)
                g.stage_end(prev)
                lat.append(time.perf_counter() - t_begin[prev])
            for i in range(M):
                g.member_index_size(i)                      # (synchronises every member: the last updates are inside the wall time)
            t_all = time.perf_counter() - t0
            lat = np.asarray(lat) * 1e3
            return {"batches": n, "decisions_per_s": R * n / t_all, "ms_per_batch": 1e3 * t_all / n, "p50_ms": float(np.percentile(lat, 50)), "p99_ms": float(np.percentile(lat, 99))}

        loop(4, False)
        plain = loop(n_batches, False)
        plain["what"] = f"eppk_group_pick_stage_begin / _end over two staging sets, {M} members: every member uploads and scores its shard of each batch"
        # LEARN: the first generations against the oracle (set 0 / 1 hold batches 0 / 1: the oracle replays begin order)
        oix = orc.OracleIndex()
        oix.insert(wl.index_hashes, wl.index_pods)
        ok = True
        cores = os.cpu_count() or 1
        for k in range(3):
            g.stage_begin(k & 1, R, learn=True)
            picks, scores = g.stage_end(k & 1)
            op, osc, _ = orc.pick_batch(wl.chain, wl.pods, oix, batches[(k & 1) % len(batches)], wl.B, threads=cores)
            ok = ok and bool(np.array_equal(picks, op)) and bool(np.array_equal(scores.view(np.uint64), osc.view(np.uint64)))
            oix.insert_picks(batches[(k & 1) % len(batches)], wl.B, op)
        sizes = [g.member_index_size(i) for i in range(M)]
        ok = ok and all(sz == oix.size() for sz in sizes)
        every = args.age_every
        state = {"epoch": 1}

        def tick(i):
            if i % every == 0:
                state["epoch"] = g.index_advance_epoch()
                if state["epoch"] > args.keep_epochs:
                    g.index_evict_older_device(state["epoch"] - args.keep_epochs + 1)
        learn = loop(n_batches, True, tick)
        learn.update({"first_batches_equal_oracle": ok, "ageing_every_batches": every, "launch_status": [int(g.member_launch_status(i)) for i in range(M)],
                      "what": f"the same with EPPK_PICK_LEARN: every member uploads the WHOLE batch, the picks are gathered (peer copies) and each member applies the post-route "
                              f"update to its replica; epoch tick + eppk_group_index_evict_older_device every {every} batches between two begins; {args.cl_slots} index slots"})
        return {"pipelined": plain, "pipelined_learn": learn, "members": M}
    finally:
        g.close()


def resident_latency_leg(pkg, wl, batches, calls: int):
    """Host-observed latency of eppk_pick_batch_staged for 1 / 4 / 16 / 32 / 64 requests with EPPK_RESIDENT=1 (fresh rows written into the
    pinned staging buffer before every call, not timed), each size checked against the oracle once.  Below 8 requests the resident
    workgroup with pick_fast_kernel's body answers, from 8 on the one with pick_quad_kernel's."""
    orc = graft.load_oracle()
    oix = orc.OracleIndex()
    oix.insert(wl.index_hashes, wl.index_pods)
    old = os.environ.get("EPPK_RESIDENT")
    os.environ["EPPK_RESIDENT"] = "1"
    try:
        pk = pkg.BatchedPicker(wl.chain, max_pods=wl.P, max_blocks=wl.B, max_batch=64, index_slots=wl.index_slots)
    finally:
        if old is None:
            os.environ.pop("EPPK_RESIDENT", None)
        else:
            os.environ["EPPK_RESIDENT"] = old
    try:
        pk.publish(wl.pods)
        pk.index_insert(wl.index_hashes, wl.index_pods)
        st_reqs, _ = pk.staging()
        by_n, ok = {}, True
        for n in (1, 4, 16, 32, 64):
            lat = []
            p, s = np.empty(n, dtype=np.int32), np.empty(n, dtype=np.float64)
            a_p, a_s = p.ctypes.data, s.ctypes.data
            for i in range(calls + 10):
                off = (i * n) % max(1, wl.R - n + 1)
                np.copyto(st_reqs[:n], batches[i % len(batches)][off:off + n])
                t0 = time.perf_counter()
                pk.pick_staged_into(n, a_p, a_s)
                lat.append(time.perf_counter() - t0)
            op, osc, _ = orc.pick_batch(wl.chain, wl.pods, oix, st_reqs[:n].copy(), wl.B)
            ok = ok and bool(np.array_equal(p, op)) and bool(np.array_equal(s.view(np.uint64), osc.view(np.uint64)))
            lat = np.asarray(lat[10:]) * 1e6
            by_n[str(n)] = {"p50_us": float(np.percentile(lat, 50)), "p99_us": float(np.percentile(lat, 99))}
        on, served, starts = pk.resident_stats()
        return {"requests": by_n, "picks_and_scores_equal_oracle": ok, "batches_answered_by_the_resident_workgroup": served, "kernel_starts": starts,
                "what": "eppk_pick_batch_staged with EPPK_RESIDENT=1 (opt-in): a resident workgroup polls a doorbell in pinned host memory, scores the batch "
                        "(below 8 requests with pick_fast_kernel's body, from 8 on with pick_quad_kernel's: one workgroup of each form) and raises a completion "
                        "word the call polls -- no launch, no completion signal; a CU is held per form in use"}
    finally:
        pk.close()


def dispatcher_latency_leg(pkg, wl, batches, calls: int, n: int = 16):
    """Host-observed latency of what the Go / C++ dispatcher really issues for a small batch (integration patch: eppk_pick_stage_begin with
    EPPK_PICK_LEARN, eppk_pick_topk for PickResult.Fallbacks, masked batches), `n` requests per call, with the library's defaults
    ("launched") and with EPPK_RESIDENT=1 ("resident": a resident workgroup per variant behind a doorbell).  Rows are written into the
    pinned buffers before every call (not timed).  Every variant is checked against the oracle once; the LEARN loop against the oracle's
    index size at the end."""
    orc = graft.load_oracle()
    out = {}
    P, B = wl.P, wl.B
    W = (P + 63) // 64
    rng = np.random.default_rng(5)
    mask = rng.integers(0, 2**63, (n, W), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, (n, W), dtype=np.uint64)
    for mode in ("launched", "resident"):
        old = os.environ.get("EPPK_RESIDENT")
        if mode == "resident":
            os.environ["EPPK_RESIDENT"] = "1"
        else:
            os.environ.pop("EPPK_RESIDENT", None)
        try:
            pk = pkg.BatchedPicker(wl.chain, max_pods=P, max_blocks=B, max_batch=256, index_slots=1 << 20)
        finally:
            if old is None:
                os.environ.pop("EPPK_RESIDENT", None)
            else:
                os.environ["EPPK_RESIDENT"] = old
        try:
            pk.publish(wl.pods)
            pk.index_insert(wl.index_hashes, wl.index_pods)
            oix = orc.OracleIndex()
            oix.insert(wl.index_hashes, wl.index_pods)
            st, stm = pk.staging(with_mask=True)
            sb, sbm = pk.stage_buffers(0, with_mask=True)
            res, ok = {}, True

            def rows(i):
                off = (i * n) % max(1, wl.R - n + 1)
                return batches[i % len(batches)][off:off + n]

            def timed(fill, call):
                lat, t_start = [], 0.0
                for i in range(calls + 10):
                    if i == 10:
                        t_start = time.perf_counter()
                    fill(i)
                    t0 = time.perf_counter()
                    call()
                    lat.append(time.perf_counter() - t0)
                cycle = (time.perf_counter() - t_start) * 1e6 / calls
                lat = np.asarray(lat[10:]) * 1e6
                return {"p50_us": float(np.percentile(lat, 50)), "p99_us": float(np.percentile(lat, 99)), "cycle_us": cycle}

            p, sc = np.empty(n * 4, dtype=np.int32), np.empty(n * 4, dtype=np.float64)
            a_p, a_s = p.ctypes.data, sc.ctypes.data
            lib, ctx = pk._lib, pk._ctx
            res["pick"] = timed(lambda i: np.copyto(st[:n], rows(i)), lambda: pk.pick_staged_into(n, a_p, a_s))
            op, osc, _ = orc.pick_batch(wl.chain, wl.pods, oix, st[:n].copy(), B)
            ok = ok and bool(np.array_equal(p[:n], op)) and bool(np.array_equal(sc[:n].view(np.uint64), osc.view(np.uint64)))
            stm[:n * W] = mask.reshape(-1)
            res["pick_masked"] = timed(lambda i: np.copyto(st[:n], rows(i)), lambda: pk.pick_staged_into(n, a_p, a_s, use_mask=True))
            op, osc, _ = orc.pick_batch(wl.chain, wl.pods, oix, st[:n].copy(), B, mask)
            ok = ok and bool(np.array_equal(p[:n], op)) and bool(np.array_equal(sc[:n].view(np.uint64), osc.view(np.uint64)))
            # what the reference's subset filter leaves (request.go:104-133): a handful of endpoints -- such a request nearly always needs its
            # own QUEUE normalisers (its candidates miss a pod at the snapshot-wide minimum or maximum)
            mask8 = np.zeros((n, W), dtype=np.uint64)
            for r_ in range(n):
                for p_ in rng.choice(P, size=min(8, P), replace=False):
                    mask8[r_, int(p_) // 64] |= np.uint64(1) << np.uint64(int(p_) % 64)
            stm[:n * W] = mask8.reshape(-1)
            res["pick_subset_8_endpoints"] = timed(lambda i: np.copyto(st[:n], rows(i)), lambda: pk.pick_staged_into(n, a_p, a_s, use_mask=True))
            op, osc, _ = orc.pick_batch(wl.chain, wl.pods, oix, st[:n].copy(), B, mask8)
            ok = ok and bool(np.array_equal(p[:n], op)) and bool(np.array_equal(sc[:n].view(np.uint64), osc.view(np.uint64)))
            st_ptr = st.ctypes.data
            res["top4"] = timed(lambda i: np.copyto(st[:n], rows(i)), lambda: lib.eppk_pick_topk(ctx, st_ptr, n, None, 4, a_p, a_s))
            op, osc = orc.pick_topk_batch(wl.chain, wl.pods, oix, st[:n].copy(), B, 4)
            ok = ok and bool(np.array_equal(p.reshape(n, 4), op)) and bool(np.array_equal(sc.view(np.uint64).reshape(n, 4), osc.view(np.uint64)))
            # ... and PickResult.Fallbacks within a subset filter of 8 endpoints (server.go:72-77 behind request.go:104-133)
            m8 = np.ascontiguousarray(mask8)
            m8_ptr = m8.ctypes.data
            res["top4_subset_8_endpoints"] = timed(lambda i: np.copyto(st[:n], rows(i)), lambda: lib.eppk_pick_topk(ctx, st_ptr, n, m8_ptr, 4, a_p, a_s))
            op, osc = orc.pick_topk_batch(wl.chain, wl.pods, oix, st[:n].copy(), B, 4, m8)
            ok = ok and bool(np.array_equal(p.reshape(n, 4), op)) and bool(np.array_equal(sc.view(np.uint64).reshape(n, 4), osc.view(np.uint64)))
            # pick + LEARN through a staging set: begin -> end is what the request waits for; the update runs on behind it, and the
            # next begin waits for it on the device (launched) / on the completion word of the resident update
            learned = []

            def learn_call():
                lib.eppk_pick_stage_begin(ctx, 0, n, 0, 1)
                lib.eppk_pick_stage_end(ctx, 0, a_p, a_s)

            def learn_fill(i):
                np.copyto(sb[:n], rows(i))
                learned.append(rows(i))
            res["pick_learn"] = timed(learn_fill, learn_call)      # (cycle_us: back to back: fill, begin, end, fill, ...)
            t_sync = time.perf_counter()
            pk.index_size()
            res["pick_learn"]["sync_after_loop_us"] = (time.perf_counter() - t_sync) * 1e6
            # the same with the device idle between two batches (a dispatcher at 10-1000 QPS: docs/proposals/006-scheduler/README.md:133):
            # the previous batch's index update is long over when the next one arrives
            def idle_fill(i):
                learn_fill(i + calls + 16)              # (other rows than the loop above has just taught the index: not revisits)
                t_w = time.perf_counter()
                while time.perf_counter() - t_w < 60e-6:
                    pass
            res["pick_learn_after_60us_idle"] = timed(idle_fill, learn_call)
            for rws in learned:                           # the oracle replays the loop: same index at the end
                o_p, _, _ = orc.pick_batch(wl.chain, wl.pods, oix, rws, B)
                oix.insert_picks(rws, B, o_p)
            ok = ok and pk.index_size() == oix.size() and bool(np.array_equal(p[:n], o_p)) and pk.index_selfcheck() == 0
            res["equal_oracle"] = ok
            if mode == "resident":
                res["batches_answered_by_resident_workgroups"] = int(pk.resident_stats()[1])
            out[mode] = res
        finally:
            pk.close()
    out["requests"] = n
    out["what"] = ("eppk_pick_batch_staged (plain / masked), eppk_pick_topk (k = 4) and eppk_pick_stage_begin(EPPK_PICK_LEARN) + _end for one small batch, host-observed; "
                   "\"launched\" = the library's defaults, \"resident\" = EPPK_RESIDENT=1 (opt-in; a resident workgroup per variant)")
    return out


def closed_loop_leg(pkg, torch, args, wl, batches):
    """The closed loop as a sub-run of the default line: `closed_loop` {value, ms_per_step, generations verified against the oracle,
    index sizes}, `roofline_closed_loop` (step parts, HBM lines per step) and `pipelined_learn` (the same loop through
    eppk_pick_stage_begin(EPPK_PICK_LEARN) / _end: what a host caller sees)."""
    import copy
    a = copy.copy(args)
    a.closed_loop, a.inflight, a.profile_every = True, 1, 1 << 30
    t_all = time.perf_counter()
    run = Runner(pkg, torch, None, wl, batches, a, 0, 1, int(os.environ.get("LOCAL_RANK", "0")), index_slots=a.cl_slots, closed_loop=True)
    try:
        run.setup("single", 1)
        info = closed_loop_verify(run, wl, a)
        state = {"epoch": info["epoch"]}

        def age(i, state=state):      # stream-ordered: behind the update of this step, ahead of the next pick
            if (i + 1) % a.age_every == 0:
                state["epoch"] = run.pk.index_advance_epoch()
                if state["epoch"] > a.keep_epochs:
                    run.pk.index_evict_older_device(state["epoch"] - a.keep_epochs + 1, run.streams[0])
        steps, warm = max(8, a.cl_steps), max(4, a.cl_steps // 4)
        elapsed, _, _ = run.timed(steps, warm, age)
        run.pk.profile(False)
        ms = 1e3 * elapsed / steps
        roof = closed_loop_roofline(run, wl, a, state, ms, steps=12)
        info.update({"value": wl.R * steps / elapsed, "unit": "decisions/s", "ms_per_step": ms, "steps": steps, "warmup": warm,
                     "step_parts_ms": roof["step_parts_ms"], "launch_status": int(run.pk.launch_status()), "index_dropped_after": int(run.pk.index_dropped()),
                     "what": "pick -> post-route index update (index[hash[r][i]] U= {pick[r]}) -> next, DIFFERENT batch on one stream; epoch tick + eviction of "
                             f"hashes not re-inserted for {a.keep_epochs} epochs every {a.age_every} steps; {a.cl_slots} index slots"})
        pl = None
        if hasattr(run.pk, "stage_begin"):
            pl = pipelined_learn_leg(run, wl, a, batches, state, n_batches=max(8, a.pl_batches))
        info["seconds"] = time.perf_counter() - t_all
        return {"closed_loop": info, "roofline_closed_loop": roof, "pipelined_learn": pl}
    finally:
        run.close()


def pipelined_learn_leg(run, wl, args, batches, state, n_batches: int = 512, rotate_batches: int = 64):
    """eppk_pick_stage_begin(EPPK_PICK_LEARN) / _end over the two staging sets -- upload of batch k + 1 under the pick and the post-route
    update of batch k -- with the shim's ageing stream-ordered on the device (eppk_index_evict_older_device between two begins: behind the
    picks and updates begun before it, ahead of those begun after; the pipeline is not drained).  Round 5 resubmitted the SAME two batches
    512 times with hashes that lived four batches: a 100 %-returning loop whose index never learned a key (verdict r5, weak #4).  Two legs now,
    both of which bring 16 NEW tail hashes per request in every batch, and both print what the picks found (`returning_fraction`, from the
    probe statistics: hits per request beyond the group's shared blocks):
      * `decisions_per_s` (the library's rate): the two pinned sets hold two different batches, and the epoch ticks + evicts BEHIND EVERY
        batch with keep = 2 epochs -- the tick behind batch k evicts what batch k - 1 learned, i.e. a batch's tail hashes have aged out right
        before it comes round again two batches later (the shared blocks are restamped by every batch and stay), so every batch teaches the
        index 1 Mi new keys and evicts 1 Mi; no host copy in the loop.
      * `rotating` (the caller's rate in THIS harness): a ring of >= 16 distinct batches, batch i copied into its pinned set right before
        its begin, INSIDE the timed loop (a thread pool of numpy slice copies standing in for a dispatcher's request threads:
        `fill_ms_per_batch`), the closed loop's own ageing policy.  Python moves 17 MB in ~1 ms, three times the pipeline's step: this
        figure measures the harness' memcpy, and says so."""
    R = wl.R
    pk = run.pk
    NB = len(batches)
    sb = [pk.stage_buffers(0)[0], pk.stage_buffers(1)[0]]
    shared = wl.meta.get("shared_blocks", 0)
    lw_bytes = 2 if wl.P <= 1024 else 4 if wl.P <= 2048 else 8

    def hits_per_request():
        abytes, lookups, launches = pk.profile_bytes()
        launches = max(int(launches), 1)
        return (max(abytes / launches - (wl.P * 64 + R * (run.stride + 4)) - 8 * lookups / launches, 0.0) / (64 * lw_bytes) if wl.B else 0.0) / R

    def loop(n, every, keep, fill):
        lat, t_begin, fills = [], [0.0, 0.0], []

        def tick():
            state["epoch"] = pk.index_advance_epoch()
            if state["epoch"] > keep:
                pk.index_evict_older_device(state["epoch"] - keep + 1, None)

        def put(which, i):
            if fill is None:
                return
            t0 = time.perf_counter()
            fill(sb[which], batches[i % NB])
            fills.append(time.perf_counter() - t0)
        for i in range(4):                  # warm-up: two batches through each set
            put(i & 1, NB - 4 + i)
            pk.stage_begin(i & 1, R, learn=True)
            pk.stage_end(i & 1)
            if every == 1:
                tick()
        if every != 1:
            tick()
        fills.clear()
        run.torch.cuda.synchronize()
        size0 = int(pk.index_size())
        pk.profile(True)                    # (probe statistics of every pick of the loop: hits per request)
        pk.profile_drain()
        t0 = time.perf_counter()
        put(0, 0)
        t_begin[0] = time.perf_counter()
        pk.stage_begin(0, R, learn=True)
        if every == 1:
            tick()
        for i in range(1, n + 1):
            cur, prev = i & 1, (i - 1) & 1
            if i < n:
                put(cur, i)                 # (set `cur` was handed back by the end of batch i - 2)
                t_begin[cur] = time.perf_counter()
                pk.stage_begin(cur, R, learn=True)
                if i % every == 0:
                    tick()                  # (set `cur` is in flight: the eviction queues behind its pick and update)
            pk.stage_end(prev)
            lat.append(time.perf_counter() - t_begin[prev])
        run.torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        lat = np.asarray(lat) * 1e3
        hpr = hits_per_request()
        pk.profile(False)
        return {"batches": int(n), "decisions_per_s": R * n / t_all, "ms_per_batch": 1e3 * t_all / n,
                "p50_ms": float(np.percentile(lat, 50)), "p99_ms": float(np.percentile(lat, 99)),
                "fill_ms_per_batch": 1e3 * float(np.mean(fills)) if fills else None, "hits_per_request": hpr,
                "returning_fraction": max(0.0, (hpr - shared) / max(1, wl.B - shared)) if wl.B else 0.0,
                "ageing_every_batches": every, "keep_epochs": keep, "index_size_before": size0, "index_size_after": int(pk.index_size())}

    # (1) the library's rate: two resident batches, aged out before they return
    np.copyto(sb[0][:R], batches[0])
    np.copyto(sb[1][:R], batches[1 % NB])
    main = loop(n_batches, 1, 2, None)
    # (2) a ring of distinct batches copied in by the harness, the closed loop's ageing policy
    from concurrent.futures import ThreadPoolExecutor
    n_thr = max(1, min(32, (os.cpu_count() or 1) // 2))
    cuts = [(R * k) // n_thr for k in range(n_thr + 1)]
    with ThreadPoolExecutor(max_workers=n_thr) as pool:
        rot = loop(max(8, min(rotate_batches, n_batches)), args.age_every, args.keep_epochs,
                   lambda dst, src: list(pool.map(lambda k: np.copyto(dst[cuts[k]:cuts[k + 1]], src[cuts[k]:cuts[k + 1]]), range(n_thr))))
    rot.update({"distinct_batches": NB, "fill_threads": n_thr,
                "what": f"a ring of {NB} distinct batches, each copied into its pinned set inside the timed loop by {n_thr} Python threads (numpy slice copies); epoch tick + eviction "
                        f"every {args.age_every} batches, keep {args.keep_epochs}: bounded by the harness' memcpy (fill_ms_per_batch), not by the library"})
    main.update({"pcie_floor_ms": R * run.stride / 55e9 * 1e3, "index_dropped": int(pk.index_dropped()), "launch_status": int(pk.launch_status()), "rotating": rot,
                 "what": "eppk_pick_stage_begin(EPPK_PICK_LEARN) / _end over the two staging sets (two different batches, already in the pinned sets): the post-route index update "
                         "chained on the device behind every pick; epoch tick + eviction behind EVERY batch, keep 2 epochs, stream-ordered on the device between two begins (no drain) "
                         "-- a batch's 1 Mi tail hashes have aged out right before it comes round again, so every batch is a batch of new requests (returning_fraction) that teaches the index "
                         f"1 Mi new keys; {args.cl_slots} index slots; wall time of the whole loop; latency = begin -> end of a batch"})
    return main


def revisit_leg(pkg, torch, args, wl, batches, fractions=(0.25, 0.5, 1.0), launches: int = 20, index_slots: int = 1 << 23):
    """RETURNING requests (the prefix scorer's own use case: docs/proposals/0602-prefix-cache-aware-routing-proposal/README.md:101-112):
    batch 0 is routed and the index learns its picks (eppk_pick_learn_device); then batches in which a fraction f of the rows are rows of
    batch 0 coming back (workload.returning_rows: scattered positions) and the rest are new requests are picked -- kernel time per 64k
    batch from the library's events, picks AND scores of every fraction checked against the oracle at full size on the same evolved
    index, and how much of each batch pick_quad_kernel deferred."""
    orc = graft.load_oracle()
    cores = os.cpu_count() or 1
    R = wl.R
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    earlier, fresh = batches[0], batches[1 % len(batches)]
    t_all = time.perf_counter()
    pk = pkg.BatchedPicker(wl.chain, max_pods=wl.P, max_blocks=wl.B, max_batch=R, index_slots=index_slots, device=int(os.environ.get("LOCAL_RANK", "0")))
    try:
        pk.publish(wl.pods)
        pk.index_insert(wl.index_hashes, wl.index_pods)
        st = torch.cuda.Stream(device=dev)
        d_pick = torch.empty(R, dtype=torch.int32, device=dev)
        d_score = torch.empty(R, dtype=torch.float64, device=dev)
        d_rows = torch.from_numpy(earlier.view(np.int64)).to(dev)
        pk.pick_learn_device(d_rows.data_ptr(), R, None, d_pick.data_ptr(), d_score.data_ptr(), st.cuda_stream)
        torch.cuda.synchronize()
        oix = orc.OracleIndex()
        oix.insert(wl.index_hashes, wl.index_pods)
        op0, _, _ = orc.pick_batch(wl.chain, wl.pods, oix, earlier, wl.B, threads=cores)
        first_ok = bool(np.array_equal(d_pick.cpu().numpy(), op0))
        oix.insert_picks(earlier, wl.B, op0)
        size_ok = int(pk.index_size()) == int(oix.size())
        by_f = {}
        for f in (0.0,) + tuple(fractions):
            rows = pkg.workload.returning_rows(fresh, earlier, f, 0x5EED0000 + args.config)
            d_rows = torch.from_numpy(rows.view(np.int64)).to(dev)
            for _ in range(3):
                pk.pick_device(d_rows.data_ptr(), R, None, d_pick.data_ptr(), d_score.data_ptr(), st.cuda_stream)
            torch.cuda.synchronize()
            l0, q0 = pk.quad_stats()
            pk.profile(True)
            pk.profile_drain()
            t0 = time.perf_counter()
            for _ in range(launches):
                pk.pick_device(d_rows.data_ptr(), R, None, d_pick.data_ptr(), d_score.data_ptr(), st.cuda_stream)
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
            k_ms = np.asarray(pk.profile_drain(), dtype=np.float64)
            pk.profile(False)
            l1, q1 = pk.quad_stats()
            op, osc, _ = orc.pick_batch(wl.chain, wl.pods, oix, rows, wl.B, threads=cores)
            ok = bool(np.array_equal(d_pick.cpu().numpy(), op)) and bool(np.array_equal(d_score.cpu().numpy().view(np.uint64), osc.view(np.uint64)))
            by_f[str(f)] = {"returning_requests": int(np.count_nonzero((rows[:, 1:] == earlier[:, 1:]).all(axis=1))) if f > 0 else 0,
                            "kernel_us_per_batch": float(k_ms.mean()) * 1e3 if k_ms.size else None,
                            "value": R / (float(k_ms.mean()) * 1e-3) if k_ms.size else None,
                            "wall_us_per_batch": 1e6 * wall / launches,
                            "quad_route_launches": int(l1 - l0), "deferred_per_launch": (q1 - q0) / max(1, l1 - l0),
                            "picks_and_scores_equal_oracle": ok}
        return {"by_fraction": by_f, "learned_batch_picks_equal_oracle": first_ok, "index_size_equal_oracle": size_ok, "index_slots": index_slots,
                "launches_timed": launches, "seconds": time.perf_counter() - t_all,
                "what": "64k x 4096 batches in which the given fraction of the requests are requests of an earlier batch coming back after the index learned where they were "
                        "routed (their 16 tail blocks on ONE pod, their 16 shared blocks on the group's pods); `value` = requests / mean kernel time of one batch alone on the "
                        "GPU (events on the launch's dispatch packets); one batch in flight"}
    finally:
        pk.close()


def closed_loop_verify(run, wl, args):
    """The first generations of the closed loop, at full size, against the oracle: pick -> index[hash] U= {pick} -> next batch
    (-> ageing), picks AND scores bit-exact on the EVOLVED index each generation.  Returns a summary for the JSON line."""
    orc = graft.load_oracle()
    cores = os.cpu_count() or 1
    oix = orc.OracleIndex()
    oix.insert(wl.index_hashes, wl.index_pods)
    ok_p = ok_s = True
    epoch = 1
    t0 = time.perf_counter()
    for g in range(args.cl_verify):
        b = run.batch_of(run.step_no)
        run.step()
        run.torch.cuda.synchronize()
        picks, scores = run.last_outputs()
        op, osc, _ = orc.pick_batch(wl.chain, wl.pods, oix, run.h_batches[b], wl.B, threads=cores)
        ok_p &= bool(np.array_equal(picks, op))
        ok_s &= bool(np.array_equal(scores.view(np.uint64), osc.view(np.uint64)))
        oix.insert_picks(run.h_batches[b], wl.B, op)
        if (g + 1) % args.age_every == 0:
            epoch = run.pk.index_advance_epoch()
            oe = oix.advance_epoch()
            assert oe == epoch
            if epoch > args.keep_epochs:
                run.torch.cuda.synchronize()       # (the synchronous form runs on the context's own stream)
                n_gpu = run.pk.index_evict_older(epoch - args.keep_epochs + 1)
                n_orc = oix.evict_older(epoch - args.keep_epochs + 1)
                ok_p &= n_gpu == n_orc
    size_gpu, size_orc = run.pk.index_size(), oix.size()
    return {"generations_verified": args.cl_verify, "picks_equal_oracle": ok_p, "scores_bitwise_equal_oracle": ok_s,
            "index_size": size_gpu, "index_size_oracle": size_orc, "index_dropped": run.pk.index_dropped(),
            "age_every": args.age_every, "keep_epochs": args.keep_epochs, "index_slots": args.cl_slots, "epoch": epoch,
            "verify_seconds": time.perf_counter() - t0}


# measured ceilings of random 64-byte-line traffic on this GPU (scripts/micro/linermw.hip -> profiles/r02_micro_linermw.txt): what a
# post-route index update is made of -- reads 46 G lines/s, plain stores 28 G, atomics 20 G (whatever line they hit)
RANDOM_LINE_READS, RANDOM_LINE_STORES, RANDOM_LINE_ATOMICS = 46e9, 28e9, 20e9


def closed_loop_roofline(run, wl, args, state, ms_per_step, steps: int = 24):
    """What a closed-loop step is made of, measured behind the timed region (same loop, every kernel group bracketed by events on
    its stream and synchronised): pick, index update (insert + re-sort of the touched lists), ageing (amortised per step); the
    HBM lines the index layout moves per step, their rate against the 8 TB/s peak and -- what actually bounds random 64-byte
    lines -- against the measured random-line ceilings."""
    torch = run.torch
    st_t = run.computes[0]
    ev = lambda: torch.cuda.Event(enable_timing=True)
    t_pick, t_ins, t_evict, new_keys, victims = [], [], [], [], []
    run.pk.profile(True if run.learn_api else False)      # (fused call: the pick kernel's own duration from the events on its dispatch packet)
    if run.learn_api:
        run.pk.profile_drain()
    for i in range(steps):
        slot = run.ring.next_slot()
        b = run.batch_of(run.step_no)
        st = run.streams[slot % len(run.streams)]
        size0 = run.pk.index_size()                                   # (synchronises: the step below runs alone)
        e0, e1, e2, e3 = ev(), ev(), ev(), ev()
        e0.record(st_t)
        if run.learn_api:      # the step's own call: pick + update behind one another on the stream
            run.pk.pick_learn_device(run.p_batches[b] + run.lo * run.stride, run.n_mine, None, run.p_picks[slot], run.p_scores[slot], st)
            e1.record(st_t)
        else:
            run.pk.pick_device(run.p_batches[b] + run.lo * run.stride, run.n_mine, None, run.p_picks[slot], run.p_scores[slot], st)
            e1.record(st_t)
            run.pk.index_insert_picks_device(run.p_batches[b] + run.lo * run.stride, run.p_picks[slot], run.n_mine, st)
        e2.record(st_t)
        run.ring.after_batch()
        run.step_no += 1
        torch.cuda.synchronize()
        size1 = run.pk.index_size()
        if run.learn_api:
            k_ms = run.pk.profile_drain()
            pick_ms_i = float(k_ms[-1]) if len(k_ms) else 0.0
            t_pick.append(pick_ms_i)
            t_ins.append(e0.elapsed_time(e2) - pick_ms_i)        # (pick + update, events around the call) - the pick kernel
        else:
            t_pick.append(e0.elapsed_time(e1)); t_ins.append(e1.elapsed_time(e2))
        new_keys.append(size1 - size0)
        if (i + 1) % args.age_every == 0:
            state["epoch"] = run.pk.index_advance_epoch()
            if state["epoch"] > args.keep_epochs:
                e2.record(st_t)
                run.pk.index_evict_older_device(state["epoch"] - args.keep_epochs + 1, st)
                e3.record(st_t)
                torch.cuda.synchronize()
                t_evict.append(e2.elapsed_time(e3)); victims.append(size1 - run.pk.index_size())
    pick_ms, ins_ms = float(np.mean(t_pick)), float(np.mean(t_ins))
    evict_ms = float(np.mean(t_evict)) / args.age_every if t_evict else 0.0
    n_new = float(np.mean(new_keys)); n_vic = (float(np.mean(victims)) / args.age_every) if victims else 0.0
    pairs = float(run.n_mine * wl.B)
    # MODEL of the HBM lines per step under the "lists first" layout (64-byte lines; an atomic or a partial store reads and writes its line):
    #   new key     bucket line read + written back (the CAS and the tag byte land in the line the look-up read) + list line read + written
    #               (a 16-byte store into a 64-byte line): 4 line transfers.  (Round 5 counted a stamp line of its own on top -- 6 transfers --
    #               that has not existed since the stamps became header tags in round 4.)
    #   known pair  bucket + list line read (the 4 096 hot keys of this workload stay in L2: not counted)
    #   victim      key word + tag written into the line the scan just read: 1 line written back
    #   scan        key words (with their header tags) of every slot, once per ageing pass
    scan_bytes = (args.cl_slots * 8.0) / args.age_every
    bytes_step = n_new * 64.0 * 4.0 + n_vic * 64.0 * 1.0 + scan_bytes + run.n_mine * (run.stride + 12.0)
    floor_ins = n_new * (1.0 / RANDOM_LINE_READS + 1.0 / RANDOM_LINE_ATOMICS + 2.0 / RANDOM_LINE_STORES)
    floor_evict = n_vic / RANDOM_LINE_STORES + scan_bytes / (HBM_PEAK_GBS * 1e9)
    step_s = ms_per_step * 1e-3
    traffic = traffic_src = None
    tj = stamped_json("pmc_traffic_closed_loop.json", kernel_source_hash())
    if tj and tj.get("age_every") == args.age_every and tj.get("requests") == run.n_mine:
        traffic, traffic_src = tj.get("hbm_bytes_per_step"), tj
    # `achieved` / `frac`: the COUNTER traffic over the step when a stamped PMC pass of this build exists (the honest figure: 0.33 in round 5),
    # else the line model above; the model always stands beside it as model_*
    ach_bytes = traffic if traffic else bytes_step
    return {"bound": "hbm-random-lines", "achieved": ach_bytes / step_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach_bytes / step_s / 1e9 / HBM_PEAK_GBS,
            "achieved_source": "rocprofv3 counters (profiles/pmc_traffic_closed_loop.json)" if traffic else "line model (no stamped counter pass for this build)",
            "model_bytes_per_step": bytes_step, "model_frac": bytes_step / step_s / 1e9 / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_frac": (traffic / step_s / 1e9 / HBM_PEAK_GBS) if traffic else None,
            "traffic_by_kernel": ({k: traffic_src[k] for k in ("pick", "index_update", "ageing_per_step") if k in traffic_src} if traffic_src else None),
            "traffic_source": ("profiles/pmc_traffic_closed_loop.json: rocprofv3 --pmc passes of `bench.py --closed-loop` with THIS kernel build (stamped with its source hash): "
                               "64 * RDREQ_64B + 128 * RDREQ_128B + WRITE_SIZE, mean per dispatch of the LEARN pick, the update kernels and (divided by age_every) the eviction"
                               if traffic else None),
            "bytes_per_step": bytes_step, "ms_per_step_timed": ms_per_step,
            "step_parts_ms": {"pick": pick_ms, "index_update": ins_ms, "ageing_per_step": evict_ms, "sum": pick_ms + ins_ms + evict_ms,
                              "note": f"each part alone on the GPU, events around it, {steps} steps behind the timed region"},
            "per_step": {"pairs": pairs, "new_keys": n_new, "victims": n_vic, "line_transfers_per_new_key": 4, "line_transfers_per_victim": 1},
            "random_line_floor_ms": {"index_update": floor_ins * 1e3, "ageing_per_step": floor_evict * 1e3,
                                     "definition": "new keys x (1 random line read / 46 G/s + 1 atomic / 20 G/s + 2 stores / 28 G/s); victims x 1 store + the streaming scan at 8 TB/s "
                                                   "(measured ceilings: profiles/r02_micro_linermw.txt)"},
            "frac_of_random_line_floor": {"index_update": floor_ins * 1e3 / ins_ms if ins_ms else None,
                                          "ageing": floor_evict * 1e3 / evict_ms if evict_ms else None},
            "kernels": "pick_quad_kernel + work-list pass | index_insert_picks_kernel + index_lists_sort_kernel | index_evict_kernel"}


def cold_reference(pkg, torch, args, khash, steps: int = 60, warmup: int = 10):
    """The pick kernel where HBM IS the bound: 262 144 prefix groups, uniform -> 4.2 M distinct hashes in 268 MB of key buckets (16.8 M API
    slots = 4.2 M buckets of 64 bytes), far beyond the 32 MB of L2 and the 256 MB Infinity Cache; every request gathers 17 random 64-byte
    buckets and ONE line of the 16 MiB set table (protocol v5; round 5: 20 buckets + 16 random 64-byte lists).  One batch at a time (the
    kernel has the GPU to itself: its duration is the launch duration)."""
    import copy
    a = copy.copy(args)
    a.groups, a.zipf, a.inflight, a.profile_every = 262144, 0.0, 1, 1
    t0 = time.perf_counter()
    wl = pkg.workload.make_workload(a.config, n_groups=a.groups, zipf_s=a.zipf, pods_per_group=4)
    batches = make_batches(pkg, wl, a, 4)        # (each batch draws 64k of the groups at random: every batch touches different lists)
    gen_s = time.perf_counter() - t0
    run = Runner(pkg, torch, None, wl, batches, a, 0, 1, int(os.environ.get("LOCAL_RANK", "0")))
    run.setup("single", 1)
    elapsed, kern_ms, stats = run.timed(steps, warmup)
    run.pk.profile(False)
    quad = run.pk.quad_stats()[0] > 0
    bm = byte_models(wl, wl.R, stats, khash, quad=quad)
    run.close()
    avg_ms = float(kern_ms.mean())
    ach = bm["cold_hbm"] / (avg_ms * 1e-3) / 1e9
    roof = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
            "workload": f"{wl.name}, cold index: {a.groups} prefix groups, uniform ({int(np.unique(wl.index_hashes).size)} distinct hashes, {wl.index_slots} slots)",
            "value": wl.R * steps / elapsed, "ms_per_step": 1e3 * elapsed / steps, "kernel_avg_ms": avg_ms, "kernel_p99_ms": float(np.percentile(kern_ms, 99)),
            "bytes_per_launch": bm["cold_hbm"],
            "frac_strict": bm["cold_strict"] / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "bytes_strict_per_launch": bm["cold_strict"],
            "frac_hbm_granular": (2.0 * bm["cold_strict"] - wl.R * (8 + 8 * wl.B + 12)) / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "strict_definition": "SURVEY 8(d) to the letter: probes = matched + 1 key buckets per request (what the sequential walk needs, device-counted; a hit's set id comes "
                                 "with its 64-byte bucket line), ONE 64-byte pod-set line per request, request rows in, picks / scores out -- without the buckets the kernel "
                                 "gathers ahead of knowing where the walk ends.  Every 64-byte line costs HBM a 128-byte request: `frac_hbm_granular` prices the same lines at 128 bytes",
            "kernel": "pick_quad_kernel" if quad else "pick_fast_kernel",
            "bytes_definition": ("what the layout reads from HBM per launch: request rows + outputs + one 64-byte key bucket per gathered hash (" +
                                 ("17 per request: pick_quad_kernel gathers the first 17 ahead, the rest only behind 17 hits" if quad else "32 per request") +
                                 ") + " + ("one 64-byte set line per request" if quad else "one 64-byte pod list per hit") + " (adapter tables stay in L2)"),
            "launches_in_flight": len(run.streams), "steps": steps, "generate_seconds": gen_s, "kernel_src_sha16": khash}
    tj = stamped_json("pmc_traffic_cold.json", khash)
    if tj:
        roof["traffic"] = tj.get("hbm_bytes_per_launch")
        roof["traffic_frac"] = roof["traffic"] / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        roof["traffic_source"] = ("profiles/pmc_traffic_cold.json: rocprofv3 TCC_EA0_RDREQ 64/128-byte request counters + WRITE_SIZE of this kernel build; about twice "
                                  "the algorithmic bytes because HBM is read in 128-byte requests and every gathered bucket / list is a 64-byte line")
    # the ceiling of this access pattern, measured: random 64-byte line gathers reach 3.3 TB/s on this GPU whatever the depth
    # (scripts/micro/linegather.hip -> profiles/r02_micro_linegather_hbm_ceiling.txt); 512-byte rows 6.2-6.9 TB/s (rowgather.hip)
    roof["gather_ceiling_GBps"] = 3300.0
    roof["frac_of_gather_ceiling"] = ach / 3300.0
    return roof


if __name__ == "__main__":
    main()
