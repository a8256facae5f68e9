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
