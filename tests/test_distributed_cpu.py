"""CPU, world_size 2 over gloo: the N>1 path (request sharding + all-gather of picks) is correct by
construction — sharded picks equal the unsharded oracle picks, including ragged shard sizes."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_reqs, masked, outdir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import __graft_entry__ as g
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg, orc = g.load_package(), g.load_oracle()
    wl = pkg.workload.make_workload(3, R=n_reqs, P=300, masked=masked)
    oix = orc.OracleIndex()
    oix.insert(wl.index_hashes, wl.index_pods)

    def pick_fn(reqs, mask):   # stand-in for the HIP picker on a CPU-only box (tests may use the oracle)
        return orc.pick_batch(wl.chain, wl.pods, oix, reqs, wl.B, mask)[0]

    allp = pkg.distributed.sharded_pick(wl.reqs, wl.mask, rank, world, pick_fn)
    full = pick_fn(wl.reqs, wl.mask)
    np.save(os.path.join(outdir, f"r{rank}.npy"), np.stack([allp, full]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_reqs,masked", [(128, False), (101, True), (1, False)])
def test_sharded_pick_equals_unsharded(tmp_path, n_reqs, masked):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), n_reqs, masked, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        allp, full = np.load(tmp_path / f"r{r}.npy")
        assert allp.shape == (n_reqs,) and np.array_equal(allp, full)


def test_shard_bounds_cover_exactly(pkg):
    d = pkg.distributed
    for n in (0, 1, 7, 64, 65, 65536):
        for w in (1, 2, 3, 8):
            spans = [d.shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            assert all(hi - lo <= d.shard_size(n, w) for lo, hi in spans)


@pytest.mark.parametrize("nbuf,every", [(8, 4), (8, 1), (8, 8), (8, 3), (6, 2), (1, 4)])
def test_gather_ring_bookkeeping(pkg, nbuf, every):
    """bench.py's N>1 path: every batch's slot is gathered exactly once, in order, by contiguous non-wrapping ranges of at most
    `gather_every` slots; a slot is never rewritten before the collective that reads it was issued; flushes may come anywhere."""
    import random
    rnd = random.Random(nbuf * 100 + every)
    ring = pkg.distributed.GatherRing(nbuf=nbuf, gather_every=every)
    assert ring.nbuf % ring.gather_every == 0 and 1 <= ring.gather_every <= max(every, 1)
    written = {}            # slot -> step currently held, not yet gathered
    gathered = []           # steps in gather order
    trip_closed_at = -1     # number of steps issued when the last trip-closing collective was issued

    def take(due):
        if due is None:
            return
        first, n, closes = due
        assert 1 <= n <= ring.gather_every and first + n <= ring.nbuf
        assert ring.bucket_of(first) == ring.bucket_of(first + n - 1) < ring.n_buckets
        for s in range(first, first + n):
            gathered.append(written.pop(s))
        nonlocal trip_closed_at
        if closes:
            assert first + n == ring.nbuf
            trip_closed_at = ring.steps

    for step in range(200):
        if ring.begins_trip():
            assert trip_closed_at == ring.steps and not written      # the whole ring was gathered before it is reused
        slot = ring.next_slot()
        assert slot not in written
        written[slot] = step
        take(ring.after_batch())
        if rnd.random() < 0.15:
            take(ring.flush())
            assert ring.flush() is None
    take(ring.flush())
    assert gathered == list(range(200)) and not written


def _ring_worker(rank, world, port, n_steps, every, outdir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import __graft_entry__ as g
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = g.load_package()
    R = 37
    ring = pkg.distributed.GatherRing(nbuf=8, gather_every=every)
    picks_all = torch.full((ring.nbuf * R,), -1, dtype=torch.int32)
    outs = [torch.empty(world * ring.gather_every * R, dtype=torch.int32) for _ in range(ring.n_buckets)]
    seen = []

    def gather(due):                      # the same calls as bench.py's gather(), on CPU tensors over gloo
        if due is None:
            return
        b0, n, _ = due
        out = outs[ring.bucket_of(b0)][: world * n * R]
        dist.all_gather_into_tensor(out, picks_all[b0 * R:(b0 + n) * R])
        seen.append(out.clone().reshape(world, n, R))

    for step in range(n_steps):
        b = ring.next_slot()
        picks_all[b * R:(b + 1) * R] = step * 1000 + rank * 100 + torch.arange(R, dtype=torch.int32)   # "the kernel"
        gather(ring.after_batch())
        if step == 4:
            gather(ring.flush())          # a fence in the middle of a bucket
    gather(ring.flush())
    got = torch.cat([s for s in seen], dim=1)                         # [world][n_steps][R]
    np.save(os.path.join(outdir, f"ring{rank}.npy"), got.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_steps,every", [(19, 4), (8, 1)])
def test_bucketed_all_gather_world2(tmp_path, n_steps, every):
    """World size 2 over gloo: the bucketed all-gather hands every rank every step's picks of every rank, in step order."""
    world = 2
    mp.spawn(_ring_worker, args=(world, _free_port(), n_steps, every, str(tmp_path)), nprocs=world, join=True)
    R = 37
    want = np.stack([np.stack([s * 1000 + r * 100 + np.arange(R, dtype=np.int32) for s in range(n_steps)]) for r in range(world)])
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"ring{r}.npy"), want)
