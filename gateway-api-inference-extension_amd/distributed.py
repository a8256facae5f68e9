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


class GatherRing:
    """Bookkeeping of a ring of `nbuf` per-batch pick buffers whose contents are all-gathered in buckets.

    A stream of batches writes slot `step % nbuf`; after every `gather_every` batches (and whenever `flush()` is called)
    the slots written since the last collective -- always one contiguous range of at most `gather_every` slots that
    does not wrap -- are gathered with ONE collective ("fewer, larger collectives": a per-batch message of a few hundred
    KiB is pure latency on xGMI).  The caller must not let a batch overwrite a slot before the collective that reads
    it has finished: `begins_trip()` tells when a new trip around the ring starts, which is where bench.py makes
    the compute streams wait for the last collective of the previous trip.  Pure bookkeeping: no torch in here.
    """

    def __init__(self, nbuf: int = 8, gather_every: int = 4) -> None:
        if nbuf < 1:
            raise ValueError("nbuf must be positive")
        g = max(1, min(int(gather_every), nbuf))
        while nbuf % g:          # buckets tile the ring, so a bucket never wraps
            g -= 1
        self.nbuf, self.gather_every = nbuf, g
        self.steps = 0           # batches issued so far
        self._first = 0          # first slot written but not yet gathered
        self._count = 0          # number of such slots

    @property
    def n_buckets(self) -> int:
        return self.nbuf // self.gather_every

    def begins_trip(self) -> bool:
        """True iff the NEXT batch starts a new trip around the ring (and it is not the very first batch)."""
        return self.steps > 0 and self.steps % self.nbuf == 0

    def next_slot(self) -> int:
        """Slot the next batch writes; call once per batch, before `after_batch()`."""
        return self.steps % self.nbuf

    def after_batch(self) -> Optional[Tuple[int, int, bool]]:
        """Account for one issued batch.  Returns (first_slot, n_slots, closes_trip) when a bucket is due, else None;
        `closes_trip` marks the collective after which the whole ring may be reused."""
        slot = self.steps % self.nbuf
        self.steps += 1
        self._count += 1
        if (slot + 1) % self.gather_every == 0:
            return self._take(slot == self.nbuf - 1)
        return None

    def flush(self) -> Optional[Tuple[int, int, bool]]:
        """The slots of a partly filled bucket (None if nothing is pending)."""
        return self._take(False) if self._count else None

    def bucket_of(self, first_slot: int) -> int:
        return first_slot // self.gather_every

    def _take(self, closes_trip: bool) -> Tuple[int, int, bool]:
        first, n = self._first, self._count
        self._first, self._count = (first + n) % self.nbuf, 0
        return first, n, closes_trip
