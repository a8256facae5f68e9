"""Request sharding across GPUs + the one collective of this path: an all-gather of per-shard picks.

SURVEY.md §8(e): under a frozen snapshot every pick depends only on its own request row and read-only
replicated state (pod table, prefix index), so the batch shards by request with NO data-path
collective; the only exchange is the all-gather that hands every rank (and the host) all picks —
which is also what lets each rank apply the same deterministic post-pick index update locally
(SEMANTICS.md §6).  One process per GPU, torch.distributed ("nccl" = RCCL over xGMI on ROCm; "gloo"
for the CPU tests).
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import numpy as np


def shard_bounds(n_reqs: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous request range [lo, hi) of `rank`: ceil-sized shards, the last ones may be short/empty."""
    per = (n_reqs + world - 1) // world
    lo = min(rank * per, n_reqs)
    return lo, min(lo + per, n_reqs)


def shard_size(n_reqs: int, world: int) -> int:
    return (n_reqs + world - 1) // world


def all_gather_picks(local_picks, n_reqs: int, world: int, group=None):
    """All-gather equal-size (padded) shards of int32 picks and return the first n_reqs entries.

    `local_picks` is a torch int32 tensor of exactly shard_size(n_reqs, world) elements (pad with -1);
    on a GPU it stays on the device and the collective runs on the current stream."""
    import torch
    import torch.distributed as dist

    per = shard_size(n_reqs, world)
    assert local_picks.dtype == torch.int32 and local_picks.numel() == per
    out = torch.empty(per * world, dtype=torch.int32, device=local_picks.device)
    if world == 1:
        out.copy_(local_picks)
    else:
        try:
            dist.all_gather_into_tensor(out, local_picks.contiguous(), group=group)
        except (RuntimeError, NotImplementedError):   # backends without the fused form
            parts = [torch.empty_like(local_picks) for _ in range(world)]
            dist.all_gather(parts, local_picks.contiguous(), group=group)
            out = torch.cat(parts)
    return out[:n_reqs]


def sharded_pick(reqs: np.ndarray, mask: Optional[np.ndarray], rank: int, world: int,
                 pick_fn: Callable[[np.ndarray, Optional[np.ndarray]], np.ndarray], group=None) -> np.ndarray:
    """Pick a whole batch cooperatively: this rank scores rows [lo, hi) with `pick_fn` (the HIP picker in
    production; tests inject the CPU oracle), then all ranks exchange picks. Returns all n_reqs picks."""
    import torch

    n = reqs.shape[0]
    lo, hi = shard_bounds(n, rank, world)
    per = shard_size(n, world)
    local = np.full(per, -1, dtype=np.int32)
    if hi > lo:
        local[: hi - lo] = pick_fn(reqs[lo:hi], None if mask is None else mask[lo:hi])
    return all_gather_picks(torch.from_numpy(local), n, world, group).numpy()
