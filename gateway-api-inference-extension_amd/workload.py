"""Synthetic pod-metric / request tables of SURVEY.md §8(d) (BASELINE.json `configs`).

Everything derives from splitmix64 with ``seed = 0x5EED0000 + config#`` so CPU oracle, GPU kernel and
bench see identical inputs.  Block hashes are real: prompts are generated as bytes and chain-hashed
with XXH64 through the library's host function (SEMANTICS.md §4) — no GPU needed to build a workload.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np

from . import _lib
from .picker import POD_DTYPE, ScorerKind, make_req_rows

_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
BLOCK_CHARS = 64  # 16-token blocks x ~4 chars


def splitmix64(seed: int, n: int) -> np.ndarray:
    """First n outputs of splitmix64 seeded with `seed` (vectorised)."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + np.arange(1, n + 1, dtype=np.uint64) * _GOLD
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def _sub(seed: int, tag: int) -> int:
    return int(splitmix64(seed ^ (tag * 0xD1B54A32D192ED03 & 0xFFFFFFFFFFFFFFFF), 1)[0])


@dataclass
class Workload:
    name: str
    R: int
    P: int
    A: int
    B: int
    chain: List[Tuple[int, int]]
    pods: np.ndarray                      # [P] POD_DTYPE
    reqs: np.ndarray                      # [R, 1+B] u64 rows
    index_hashes: np.ndarray              # [E] u64
    index_pods: np.ndarray                # [E] u32
    index_slots: int
    mask: Optional[np.ndarray] = None     # [R, ceil(P/64)] u64
    meta: dict = field(default_factory=dict)

    @property
    def adapter(self) -> np.ndarray:
        return (self.reqs[:, 0] & np.uint64(0xFFFFFFFF)).astype(np.uint32).view(np.int32)

    @property
    def n_blocks(self) -> np.ndarray:
        return (self.reqs[:, 0] >> np.uint64(32)).astype(np.uint32)


FULL_CHAIN = [(ScorerKind.QUEUE, 2), (ScorerKind.KV, 2), (ScorerKind.LORA, 1), (ScorerKind.PREFIX, 3)]

# BASELINE.json configs (index = config number)
CONFIGS = {
    1: dict(name="C1 128x16 queue+kv", R=128, P=16, A=0, B=0, chain=[(ScorerKind.QUEUE, 1), (ScorerKind.KV, 1)]),
    2: dict(name="C2 4kx256 queue+kv+lora", R=4096, P=256, A=128, B=0,
            chain=[(ScorerKind.QUEUE, 2), (ScorerKind.KV, 2), (ScorerKind.LORA, 1)]),
    3: dict(name="C3 8kx1024 prefix B=32", R=8192, P=1024, A=128, B=32, chain=FULL_CHAIN),
    4: dict(name="C4 16kx2048x128 lora+queue", R=16384, P=2048, A=128, B=0,
            chain=[(ScorerKind.QUEUE, 2), (ScorerKind.LORA, 1)]),
    5: dict(name="C5 64kx4096 full chain + prefix B=32", R=65536, P=4096, A=128, B=32, chain=FULL_CHAIN),
}


def make_pods(seed: int, P: int, A: int) -> np.ndarray:
    pods = np.zeros(P, dtype=POD_DTYPE)
    if P == 0:
        return pods
    pods["queue"] = (splitmix64(_sub(seed, 1), P) % np.uint64(64)).astype(np.uint32)       # many ties by design
    pods["running"] = (splitmix64(_sub(seed, 2), P) % np.uint64(256)).astype(np.uint32)
    pods["kv_util"] = (splitmix64(_sub(seed, 3), P) % np.uint64(1025)).astype(np.float64) / 1024.0
    pods["max_lora"] = (4 + 4 * (splitmix64(_sub(seed, 4), P) & np.uint64(1))).astype(np.uint32)
    if A > 0:
        nact = (splitmix64(_sub(seed, 5), P) % (pods["max_lora"].astype(np.uint64) + np.uint64(1))).astype(np.int64)
        cand = (splitmix64(_sub(seed, 6), P * 8) % np.uint64(A)).astype(np.int64).reshape(P, 8)
        nwait = (splitmix64(_sub(seed, 7), P) % np.uint64(3)).astype(np.int64)
        wcand = (splitmix64(_sub(seed, 8), P * 2) % np.uint64(A)).astype(np.int64).reshape(P, 2)
        act = np.zeros((P, 2), dtype=np.uint64)
        wai = np.zeros((P, 2), dtype=np.uint64)
        for i in range(8):
            sel = nact > i
            a = cand[:, i]
            np.bitwise_or.at(act, (np.nonzero(sel)[0], (a[sel] >> 6)), np.uint64(1) << (a[sel] & 63).astype(np.uint64))
        for i in range(2):
            sel = nwait > i
            a = wcand[:, i]
            np.bitwise_or.at(wai, (np.nonzero(sel)[0], (a[sel] >> 6)), np.uint64(1) << (a[sel] & 63).astype(np.uint64))
        pods["active"] = act
        pods["waiting"] = wai
    return pods


def _model_name(adapter: int) -> bytes:
    return b"base-model" if adapter < 0 else b"adapter-%d" % adapter


def make_workload(config: int, R: Optional[int] = None, P: Optional[int] = None, masked: bool = False,
                  n_groups: int = 256, pods_per_group: int = 8, seed: Optional[int] = None,
                  req_seed: Optional[int] = None, zipf_s: float = 1.0, B: Optional[int] = None) -> Workload:
    """Build config `config` of BASELINE.json (optionally with R / P overridden for small parity cases)."""
    c = dict(CONFIGS[config])
    if R is not None:
        c["R"] = R
    if P is not None:
        c["P"] = P
    if B is not None:
        c["B"] = B
    R, P, A, B = c["R"], c["P"], c["A"], c["B"]
    seed = (0x5EED0000 + config) if seed is None else seed
    rseed = seed if req_seed is None else req_seed   # request streams only (pods / groups / index stay on `seed`)
    lib = _lib.load_library()

    pods = make_pods(seed, P, A)

    # requests: Zipf(s=1) over shared "system prompt" groups; a group carries its tenant's adapter
    gw = 1.0 / np.arange(1, n_groups + 1, dtype=np.float64) ** zipf_s   # zipf_s = 0 -> uniform (cold-cache variant)
    cdf = np.cumsum(gw) / gw.sum()
    u = (splitmix64(_sub(rseed, 10), R) >> np.uint64(11)).astype(np.float64) / float(1 << 53)
    group = np.minimum(np.searchsorted(cdf, u, side="right"), n_groups - 1).astype(np.int64)
    ga_r = splitmix64(_sub(seed, 11), n_groups)
    if A > 0:
        group_adapter = np.where((ga_r & np.uint64(3)) == 0, -1, ((ga_r >> np.uint64(8)) % np.uint64(A)).astype(np.int64)).astype(np.int32)
    else:
        group_adapter = np.full(n_groups, -1, dtype=np.int32)
    adapter = group_adapter[group]

    Bs = B // 2          # shared prefix blocks
    Bu = B - Bs          # unique tail blocks
    hashes = np.zeros((R, max(B, 1)), dtype=np.uint64)
    idx_h: List[np.ndarray] = []
    idx_p: List[np.ndarray] = []
    if B > 0:
        gbytes = splitmix64(_sub(seed, 12), n_groups * Bs * (BLOCK_CHARS // 8)).reshape(n_groups, -1)
        tbytes = splitmix64(_sub(rseed, 13), R * Bu * (BLOCK_CHARS // 8)).reshape(R, -1)
        out = np.zeros(B, dtype=np.uint64)
        for r in range(R):
            g = int(group[r])
            prompt = gbytes[g].tobytes() + tbytes[r].tobytes()
            model = _model_name(int(adapter[r]))
            n = lib.eppk_hash_prompt(model, len(model), prompt, len(prompt), BLOCK_CHARS, out.ctypes.data, B)
            assert n == B
            hashes[r, :B] = out
        # index pre-population: each group's shared blocks cached on `pods_per_group` pods
        gp = (splitmix64(_sub(seed, 14), n_groups * pods_per_group) % np.uint64(max(P, 1))).astype(np.uint32).reshape(n_groups, pods_per_group)
        gout = np.zeros((n_groups, Bs), dtype=np.uint64)
        models = {}
        stride = Bs * 8
        base = gout.ctypes.data
        gb = np.ascontiguousarray(gbytes)
        plen = gb.shape[1] * 8
        for g in range(n_groups):
            a = int(group_adapter[g])
            model = models.get(a)
            if model is None:
                model = models[a] = _model_name(a)
            n = lib.eppk_hash_prompt(model, len(model), gb.ctypes.data + g * plen, plen, BLOCK_CHARS, base + g * stride, Bs)
            assert n == Bs
        # pair order: group-major, then block, then pod (what the per-group repeat / tile of the first version produced)
        idx_h.append(np.repeat(gout.reshape(-1), pods_per_group))
        idx_p.append(np.tile(gp[:, None, :], (1, Bs, 1)).reshape(-1))
    index_hashes = np.concatenate(idx_h) if idx_h else np.zeros(0, dtype=np.uint64)
    index_pods = np.concatenate(idx_p) if idx_p else np.zeros(0, dtype=np.uint32)
    n_keys = n_groups * Bs if B > 0 else 0
    slots = 64
    while slots < 4 * n_keys:   # load factor <= 0.25: the sizing libeppk recommends (bucket overflows become negligible)
        slots *= 2
    index_slots = slots if B > 0 else 0

    n_blocks = np.full(R, B, dtype=np.uint32)
    reqs = make_req_rows(adapter, n_blocks, hashes[:, :B] if B else None, B)

    mask = None
    if masked:
        W = (P + 63) // 64
        mask = splitmix64(_sub(rseed, 15), R * max(W, 1)).reshape(R, max(W, 1))[:, :W].copy()  # ~50 % of pods
        if P % 64 and W:
            mask[:, -1] &= np.uint64((1 << (P % 64)) - 1)

    return Workload(name=c["name"], R=R, P=P, A=A, B=B, chain=[(int(k), int(w)) for k, w in c["chain"]], pods=pods,
                    reqs=reqs, index_hashes=index_hashes, index_pods=index_pods.astype(np.uint32), index_slots=index_slots,
                    mask=mask, meta=dict(config=config, seed=seed, n_groups=n_groups, pods_per_group=pods_per_group, zipf_s=zipf_s,
                                         shared_blocks=Bs, unique_blocks=Bu, group_adapter=group_adapter,
                                         group_bytes=gbytes if B > 0 else None))


def returning_rows(fresh: np.ndarray, earlier: np.ndarray, revisit_frac: float, seed: int) -> np.ndarray:
    """A batch in which a fraction `revisit_frac` of the rows are RETURNING requests: row r of `earlier` -- a batch that has been routed
    and whose picks the index has learned -- takes the place of row r of `fresh` wherever a splitmix64 draw falls below the fraction
    (positions scattered over the batch, so that returning and new requests share wavefronts).  What the prefix scorer exists for
    (docs/proposals/0602-prefix-cache-aware-routing-proposal/README.md:101-112): such a request finds its whole prompt in the index --
    the group's shared blocks on the group's pods, its own tail blocks on the ONE pod it was routed to."""
    assert fresh.shape == earlier.shape
    if revisit_frac <= 0.0:
        return fresh.copy()
    R = fresh.shape[0]
    u = (splitmix64(_sub(seed, 16), R) >> np.uint64(11)).astype(np.float64) / float(1 << 53)
    back = u < revisit_frac
    out = fresh.copy()
    out[back] = earlier[back]
    return out


def make_requests(wl: Workload, req_seed: int, revisit_of: Optional[np.ndarray] = None, revisit_frac: float = 0.0) -> np.ndarray:
    """Another batch of request rows against the SAME snapshot, prefix groups and index as `wl` (bench.py rotates through several
    distinct batches): exactly the rows make_workload(..., req_seed=req_seed) would produce, without rebuilding pods and index.
    `revisit_of` / `revisit_frac`: that fraction of the rows are returning requests out of the earlier batch `revisit_of`
    (returning_rows above)."""
    if revisit_of is not None and revisit_frac > 0.0:
        return returning_rows(make_requests(wl, req_seed), revisit_of, revisit_frac, req_seed)
    m = wl.meta
    R, B, n_groups = wl.R, wl.B, m["n_groups"]
    lib = _lib.load_library()
    gw = 1.0 / np.arange(1, n_groups + 1, dtype=np.float64) ** m["zipf_s"]
    cdf = np.cumsum(gw) / gw.sum()
    u = (splitmix64(_sub(req_seed, 10), R) >> np.uint64(11)).astype(np.float64) / float(1 << 53)
    group = np.minimum(np.searchsorted(cdf, u, side="right"), n_groups - 1).astype(np.int64)
    adapter = m["group_adapter"][group]
    hashes = np.zeros((R, max(B, 1)), dtype=np.uint64)
    if B > 0:
        Bu = m["unique_blocks"]
        gbytes = m["group_bytes"]
        tbytes = splitmix64(_sub(req_seed, 13), R * Bu * (BLOCK_CHARS // 8)).reshape(R, -1)
        out = np.zeros(B, dtype=np.uint64)
        for r in range(R):
            prompt = gbytes[int(group[r])].tobytes() + tbytes[r].tobytes()
            model = _model_name(int(adapter[r]))
            n = lib.eppk_hash_prompt(model, len(model), prompt, len(prompt), BLOCK_CHARS, out.ctypes.data, B)
            assert n == B
            hashes[r, :B] = out
    return make_req_rows(adapter, np.full(R, B, dtype=np.uint32), hashes[:, :B] if B else None, B)
