"""Host-side mirror of the reference picker seam, on top of the C ABI (include/eppk.h).

Reference shapes mirrored (same names, argument meaning and error behaviour):

* ``EndpointPicker.Pick(ctx, *PickRequest, []*Endpoint) (*PickResult, error)``
  — pkg/lwepp/handlers/server.go:79-82; ``BatchedPicker.pick_endpoints`` is the batched form.
* ``RoundRobinPicker`` — server.go:84-101 (pre-increment atomic counter, ``Unavailable`` on empty).
* subset filter of ``handleRequestHeaders`` — request.go:104-133 → ``subset_mask``.
* ``WeightedScorer{Scorer, weight int}`` / ``SchedulerProfile`` —
  docs/proposals/0845-scheduler-architecture-proposal/interfaces/interface.go:70-79, :132-135
  → the ``chain`` argument.

Everything that scores or picks runs in libeppk's HIP kernels; this file only marshals buffers.
"""
from __future__ import annotations

import ctypes as C
import enum
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import EppkError

POD_DTYPE = np.dtype([("queue", "<u4"), ("running", "<u4"), ("kv_util", "<f8"), ("max_lora", "<u4"),
                      ("flags", "<u4"), ("active", "<u8", (2,)), ("waiting", "<u8", (2,)),
                      ("reserved", "<u8")])
assert POD_DTYPE.itemsize == 64


class ScorerKind(enum.IntEnum):
    QUEUE = 1
    KV = 2
    LORA = 3
    PREFIX = 4


class Unavailable(RuntimeError):
    """codes.Unavailable: no endpoints available (server.go:91-93, request.go:100-102)."""


@dataclass
class Endpoint:
    """Identity of a candidate — pkg/lwepp/datastore/datastore.go:40-46."""
    address: str
    port: str
    pod_name: str = ""


@dataclass
class PickResult:
    """server.go:72-77."""
    endpoint: str
    fallbacks: List[str] = field(default_factory=list)


def join_host_port(host: str, port: str) -> str:
    """net.JoinHostPort as used at server.go:99 (IPv6 literals are bracketed)."""
    return f"[{host}]:{port}" if (":" in host or "%" in host) else f"{host}:{port}"


def make_req_rows(adapter: np.ndarray, n_blocks: np.ndarray, hashes: Optional[np.ndarray], max_blocks: int) -> np.ndarray:
    """Pack request rows: [R, 1+max_blocks] u64; word 0 = adapter (i32, low half) | n_blocks << 32."""
    adapter = np.asarray(adapter, dtype=np.int32)
    n_blocks = np.asarray(n_blocks, dtype=np.uint32)
    R = adapter.shape[0]
    rows = np.zeros((R, 1 + max_blocks), dtype=np.uint64)
    rows[:, 0] = adapter.view(np.uint32).astype(np.uint64) | (n_blocks.astype(np.uint64) << np.uint64(32))
    if hashes is not None and max_blocks:
        h = np.asarray(hashes, dtype=np.uint64)
        rows[:, 1:1 + h.shape[1]] = h
    return rows


def subset_mask(endpoints: Sequence[Endpoint], filter_value: Optional[str]) -> Tuple[np.ndarray, int]:
    """Candidate bitmask of one request — request.go:104-133. Returns (mask words, n_candidates)."""
    lib = _lib.load_library()
    n = len(endpoints)
    addrs = (C.c_char_p * max(n, 1))(*[e.address.encode() for e in endpoints])
    ports = (C.c_char_p * max(n, 1))(*[e.port.encode() for e in endpoints])
    mask = np.zeros(max((n + 63) // 64, 1), dtype=np.uint64)
    rc = lib.eppk_subset_mask(addrs, ports, n, None if filter_value is None else filter_value.encode(),
                              mask.ctypes.data)
    if rc < 0:
        raise EppkError(rc, "eppk_subset_mask")
    return mask[: (n + 63) // 64], rc


def subset_entries(filter_value: Optional[str]) -> np.ndarray:
    """Entries of one request's subset filter as 128-bit fingerprints, [n, 2] u64 (include/eppk.h eppk_subset_entries):
    None (no filter) -> the single "every pod" entry (0, 0); "" -> no entries (fail closed)."""
    lib = _lib.load_library()
    raw = None if filter_value is None else filter_value.encode()
    cap = 8
    while True:
        out = np.zeros((cap, 2), dtype=np.uint64)
        n = lib.eppk_subset_entries(raw, out.ctypes.data, cap)
        if n < 0:
            raise EppkError(n, "eppk_subset_entries")
        if n <= cap:
            return out[:n]
        cap = n


def subset_entries_csr(filters: Sequence[Optional[str]]) -> Tuple[np.ndarray, np.ndarray]:
    """Entry lists of a batch in the CSR form eppk_subset_masks / eppk_pick_batch_subset take: (keys [total, 2] u64, off [R + 1] u32)."""
    per = [subset_entries(f) for f in filters]
    off = np.zeros(len(per) + 1, dtype=np.uint32)
    off[1:] = np.cumsum([e.shape[0] for e in per], dtype=np.uint64)
    keys = np.concatenate(per) if per else np.zeros((0, 2), dtype=np.uint64)
    return np.ascontiguousarray(keys, dtype=np.uint64).reshape(-1, 2), off


TEST_ENDPOINT_SELECTION_HEADER = "test-epp-endpoint-selection"          # request.go:84-97
SUBSET_FILTER_NAMESPACE = "envoy.lb.subset_hint"                        # pkg/lwepp/metadata/consts.go:21
SUBSET_FILTER_KEY = "x-gateway-destination-endpoint-subset"             # pkg/lwepp/metadata/consts.go:24
_GO_SPACE = (" \t\n\v\f\r\x85\xa0\u1680" + "".join(chr(c) for c in range(0x2000, 0x200B)) + "\u2028\u2029\u202f\u205f\u3000")   # unicode.IsSpace (strings.TrimSpace)


def resolve_subset_filter(headers: Sequence[Tuple[str, str]], request_metadata: Optional[dict]) -> Optional[str]:
    """Which subset filter governs a request -- handleRequestHeaders, request.go:44-97, up to the PodList call.

    `headers`: (key, value) pairs in wire order; `request_metadata`: the extracted filter metadata
    ({namespace: {key: str | list}}).  Returns None when no filter applies (every pod is a candidate, request.go:136-137)
    or the comma-joined entries for `subset_mask` ("" = a subset filter that is present but empty: zero candidates,
    fail closed, request.go:128-131).  The test header wins over metadata (request.go:84-97)."""
    metadata_endpoints: List[str] = []
    has_subset = False
    ns = (request_metadata or {}).get(SUBSET_FILTER_NAMESPACE)
    if isinstance(ns, dict) and SUBSET_FILTER_KEY in ns:
        has_subset = True
        val = ns[SUBSET_FILTER_KEY]
        parts = [val] if isinstance(val, str) else [v for v in val if isinstance(v, str)] if isinstance(val, (list, tuple)) else []
        for part in parts:
            for ep in part.split(","):
                t = ep.strip(_GO_SPACE)
                if t:
                    metadata_endpoints.append(t)
    filter_endpoints: List[str] = []
    for key, value in headers:
        if key == TEST_ENDPOINT_SELECTION_HEADER:
            if value != "":
                filter_endpoints = value.split(",")
            break
    if not filter_endpoints and metadata_endpoints:
        filter_endpoints = metadata_endpoints
    if has_subset or filter_endpoints:
        return ",".join(filter_endpoints)
    return None


def handle_request_headers(endpoints: Sequence[Endpoint], headers: Sequence[Tuple[str, str]],
                           request_metadata: Optional[dict]) -> List[Endpoint]:
    """reqCtx.Candidates of handleRequestHeaders (request.go:34-139) for one request; Unavailable when the datastore is empty."""
    if len(endpoints) == 0:
        raise Unavailable("no pods available")                          # request.go:100-102
    mask, _ = subset_mask(endpoints, resolve_subset_filter(headers, request_metadata))
    return [e for i, e in enumerate(endpoints) if (int(mask[i >> 6]) >> (i & 63)) & 1]


def hash_prompt(model: bytes, prompt: bytes, block_chars: int, max_blocks: int) -> np.ndarray:
    """Chained XXH64 block hashes of one prompt (SEMANTICS.md §4)."""
    lib = _lib.load_library()
    out = np.zeros(max_blocks, dtype=np.uint64)
    n = lib.eppk_hash_prompt(model, len(model), prompt, len(prompt), block_chars, out.ctypes.data, max_blocks)
    if n < 0:
        raise EppkError(n, "eppk_hash_prompt")
    return out[:n]


class RoundRobinPicker:
    """server.go:84-101 — the reference's picker; the shim's fail-open fallback."""

    def __init__(self) -> None:
        self._ctr = C.c_uint64(0)
        self._lib = _lib.load_library()

    def pick_index(self, n_candidates: int) -> int:
        idx = self._lib.eppk_round_robin(C.byref(self._ctr), n_candidates)
        if idx < 0:
            raise Unavailable("no endpoints available")
        return idx

    def Pick(self, req, endpoints: Sequence[Endpoint]) -> PickResult:  # noqa: N802 (reference name)
        e = endpoints[self.pick_index(len(endpoints))] if len(endpoints) else None
        if e is None:
            raise Unavailable("no endpoints available")
        return PickResult(endpoint=join_host_port(e.address, e.port))


class BatchedPicker:
    """One SchedulerProfile (weighted scorer chain + best-score picker) bound to one GPU."""

    def __init__(self, chain: Sequence[Tuple[int, int]], max_pods: int, max_blocks: int = 0, max_batch: int = 65536,
                 index_slots: int = 0, device: int = 0) -> None:
        self._lib = _lib.load_library()
        cfg = _lib.Cfg()
        cfg.struct_size = C.sizeof(_lib.Cfg)
        cfg.device = device
        cfg.max_pods = max_pods
        cfg.max_blocks = max_blocks
        cfg.max_batch = max_batch
        cfg.index_slots = index_slots
        cfg.n_scorers = len(chain)
        if len(chain) > _lib.EPPK_MAX_SCORERS:
            raise EppkError(-2, "more than 8 scorers")
        for i, (kind, weight) in enumerate(chain):
            cfg.chain[i].kind = int(kind)
            cfg.chain[i].weight = int(weight)
        self._ctx = C.c_void_p()
        rc = self._lib.eppk_create(C.byref(cfg), C.byref(self._ctx))
        if rc != 0:
            raise EppkError(rc, (self._lib.eppk_last_error(None) or b"").decode())
        self.chain = [(int(k), int(w)) for k, w in chain]
        self.max_pods, self.max_blocks, self.max_batch, self.index_slots = max_pods, max_blocks, max_batch, index_slots
        self.device = device
        self.n_pods = 0
        self.row_words = 1 + max_blocks

    # -- plumbing ---------------------------------------------------------------------------
    def _check(self, rc: int, what: str) -> None:
        if rc != 0:
            raise EppkError(rc, f"{what}: {(self._lib.eppk_last_error(self._ctx) or b'').decode()}")

    def close(self) -> None:
        if getattr(self, "_ctx", None) is not None and self._ctx:
            self._lib.eppk_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- snapshot / index -------------------------------------------------------------------
    def publish(self, pods: np.ndarray, epoch: int = 0) -> None:
        pods = np.ascontiguousarray(pods, dtype=POD_DTYPE)
        self._check(self._lib.eppk_snapshot_publish(self._ctx, pods.ctypes.data, pods.shape[0], epoch), "snapshot_publish")
        self.n_pods = int(pods.shape[0])

    def index_clear(self) -> None:
        self._check(self._lib.eppk_index_clear(self._ctx), "index_clear")

    def index_insert(self, hashes: np.ndarray, pods: np.ndarray) -> None:
        h = np.ascontiguousarray(hashes, dtype=np.uint64).ravel()
        p = np.ascontiguousarray(pods, dtype=np.uint32).ravel()
        assert h.shape == p.shape
        self._check(self._lib.eppk_index_insert(self._ctx, h.ctypes.data, p.ctypes.data, h.shape[0]), "index_insert")

    def index_remove_pod(self, pod: int) -> None:
        self._check(self._lib.eppk_index_remove_pod(self._ctx, pod), "index_remove_pod")

    def index_dropped(self) -> int:
        """Inserts dropped so far because the table was at capacity (include/eppk.h eppk_index_dropped)."""
        n = C.c_uint64(0)
        self._check(self._lib.eppk_index_dropped(self._ctx, C.byref(n)), "index_dropped")
        return n.value

    def index_selfcheck(self) -> int:
        """Diagnostic: index rows violating an internal invariant (0 on a healthy index; include/eppk.h eppk_index_selfcheck)."""
        n = C.c_uint64(0)
        self._check(self._lib.eppk_index_selfcheck(self._ctx, C.byref(n)), "index_selfcheck")
        return n.value

    def index_advance_epoch(self) -> int:
        """Tick the index epoch that stamps every later insert (ageing, include/eppk.h)."""
        e = C.c_uint32(0)
        self._check(self._lib.eppk_index_advance_epoch(self._ctx, C.byref(e)), "index_advance_epoch")
        return e.value

    def index_evict_older(self, min_epoch: int) -> int:
        """Drop every hash last inserted before `min_epoch`; returns how many were dropped."""
        n = C.c_uint32(0)
        self._check(self._lib.eppk_index_evict_older(self._ctx, min_epoch, C.byref(n)), "index_evict_older")
        return n.value

    def index_evict_older_device(self, min_epoch: int, stream: int = 0) -> None:
        """The same eviction, asynchronous on `stream` (hipStream_t as int), no count (include/eppk.h)."""
        self._check(self._lib.eppk_index_evict_older_device(self._ctx, min_epoch, stream or None), "index_evict_older_device")

    def index_trim_pods(self, cap_per_pod: int) -> int:
        """Per-pod capacity (include/eppk.h eppk_index_trim_pods; SEMANTICS.md §6c); returns the (hash, pod) pairs removed."""
        n = C.c_uint64(0)
        self._check(self._lib.eppk_index_trim_pods(self._ctx, int(cap_per_pod), C.byref(n)), "index_trim_pods")
        return n.value

    def index_size(self) -> int:
        n = C.c_uint32(0)
        self._check(self._lib.eppk_index_size(self._ctx, C.byref(n)), "index_size")
        return n.value

    # -- the hot path -----------------------------------------------------------------------
    def pick(self, reqs: np.ndarray, mask: Optional[np.ndarray] = None) -> Tuple[np.ndarray, np.ndarray]:
        """Host-buffer entry point. reqs: [R, 1+max_blocks] u64 rows; mask: [R, ceil(P/64)] u64 or None."""
        reqs = np.ascontiguousarray(reqs, dtype=np.uint64)
        R = reqs.shape[0]
        assert reqs.ndim == 2 and reqs.shape[1] == self.row_words, "request row stride mismatch"
        picks = np.empty(R, dtype=np.int32)
        scores = np.empty(R, dtype=np.float64)
        mptr = None
        if mask is not None:
            mask = np.ascontiguousarray(mask, dtype=np.uint64)
            assert mask.shape == (R, (self.n_pods + 63) // 64), "mask shape mismatch"
            mptr = mask.ctypes.data
        self._check(self._lib.eppk_pick_batch(self._ctx, reqs.ctypes.data, R, mptr, picks.ctypes.data, scores.ctypes.data), "pick_batch")
        return picks, scores

    def staging(self, with_mask: bool = False) -> Tuple[np.ndarray, Optional[np.ndarray]]:
        """The context's pinned staging buffers as numpy arrays: [max_batch, row_words] u64 request rows and a FLAT u64 array for the
        mask rows (pack them [R, ceil(n_pods / 64)] at its start): fill the first R rows in place, then `pick_staged(R)`
        (include/eppk.h eppk_host_staging)."""
        rp, mp = C.c_void_p(0), C.c_void_p(0)
        self._check(self._lib.eppk_host_staging(self._ctx, C.byref(rp), C.byref(mp) if with_mask else None), "host_staging")
        reqs = np.frombuffer((C.c_uint64 * (self.max_batch * self.row_words)).from_address(rp.value), dtype=np.uint64).reshape(self.max_batch, self.row_words)
        mask = None
        if with_mask:
            jmax = (self.max_pods + 63) // 64
            mask = np.frombuffer((C.c_uint64 * (self.max_batch * jmax)).from_address(mp.value), dtype=np.uint64)
        return reqs, mask

    def pick_staged(self, R: int, use_mask: bool = False) -> Tuple[np.ndarray, np.ndarray]:
        """`pick` over the first R rows of the staging buffers (mask rows: [R, ceil(n_pods/64)] packed at the START of the mask buffer)."""
        picks = np.empty(R, dtype=np.int32)
        scores = np.empty(R, dtype=np.float64)
        self._check(self._lib.eppk_pick_batch_staged(self._ctx, R, 1 if use_mask else 0, picks.ctypes.data, scores.ctypes.data), "pick_batch_staged")
        return picks, scores

    def pick_staged_into(self, R: int, picks_ptr: int, scores_ptr: int, use_mask: bool = False) -> None:
        """`pick_staged` into caller-owned arrays given by ADDRESS (`arr.ctypes.data`, taken once): the call a latency measurement
        times -- without the two allocations and pointer conversions `pick_staged` does per call (3-4 us of Python, none of which a
        cgo caller has)."""
        rc = self._lib.eppk_pick_batch_staged(self._ctx, R, 1 if use_mask else 0, picks_ptr, scores_ptr)
        if rc:
            self._check(rc, "pick_batch_staged")

    # -- the pipelined host path (include/eppk.h eppk_pick_stage_*): two staging sets, upload of one batch under the kernel of the other
    def stage_buffers(self, which: int, with_mask: bool = False) -> Tuple[np.ndarray, Optional[np.ndarray]]:
        """Pinned request rows [max_batch, row_words] u64 (and the flat mask buffer) of staging set `which` (0 / 1)."""
        rp, mp = C.c_void_p(0), C.c_void_p(0)
        self._check(self._lib.eppk_pick_stage_buffers(self._ctx, which, C.byref(rp), C.byref(mp) if with_mask else None), "pick_stage_buffers")
        reqs = np.frombuffer((C.c_uint64 * (self.max_batch * self.row_words)).from_address(rp.value), dtype=np.uint64).reshape(self.max_batch, self.row_words)
        mask = None
        if with_mask:
            jmax = (self.max_pods + 63) // 64
            mask = np.frombuffer((C.c_uint64 * (self.max_batch * jmax)).from_address(mp.value), dtype=np.uint64)
        return reqs, mask

    def stage_begin(self, which: int, R: int, use_mask: bool = False, learn: bool = False) -> None:
        """Enqueue upload + pick (+ the post-route index update on the device when `learn`) + download of set `which`; returns at once."""
        self._check(self._lib.eppk_pick_stage_begin(self._ctx, which, R, 1 if use_mask else 0, 1 if learn else 0), "pick_stage_begin")
        self._stage_n = getattr(self, "_stage_n", {})
        self._stage_n[which] = R

    def stage_end(self, which: int) -> Tuple[np.ndarray, np.ndarray]:
        """Wait for the picks of set `which` and return (picks, scores)."""
        R = getattr(self, "_stage_n", {}).get(which, 0)
        picks = np.empty(R, dtype=np.int32)
        scores = np.empty(R, dtype=np.float64)
        self._check(self._lib.eppk_pick_stage_end(self._ctx, which, picks.ctypes.data, scores.ctypes.data), "pick_stage_end")
        return picks, scores

    # -- subset filter resolved on the device (include/eppk.h "the subset filter for a whole batch") ------------------
    def set_addresses(self, endpoints: Sequence[Optional[Endpoint]]) -> None:
        """Address / port of every slot of the CURRENT snapshot (None = a hole)."""
        n = len(endpoints)
        addrs = (C.c_char_p * max(n, 1))(*[None if e is None else e.address.encode() for e in endpoints])
        ports = (C.c_char_p * max(n, 1))(*[None if e is None else e.port.encode() for e in endpoints])
        self._check(self._lib.eppk_snapshot_set_addresses(self._ctx, addrs, ports, n), "snapshot_set_addresses")

    def subset_masks(self, filters: Sequence[Optional[str]]) -> np.ndarray:
        """Mask rows [R, ceil(P/64)] of a batch of subset filters (None = no filter), built by the device."""
        keys, off = subset_entries_csr(filters)
        R = len(filters)
        out = np.zeros((R, (self.n_pods + 63) // 64), dtype=np.uint64)
        self._check(self._lib.eppk_subset_masks(self._ctx, keys.ctypes.data, off.ctypes.data, R, out.ctypes.data), "subset_masks")
        return out

    def pick_subset(self, reqs: np.ndarray, filters: Sequence[Optional[str]]) -> Tuple[np.ndarray, np.ndarray]:
        """eppk_pick_batch_subset: the pick of a batch whose candidates are given as subset filters (request.go:104-133)."""
        reqs = np.ascontiguousarray(reqs, dtype=np.uint64)
        R = reqs.shape[0]
        assert reqs.ndim == 2 and reqs.shape[1] == self.row_words and len(filters) == R, "request rows / filters mismatch"
        keys, off = subset_entries_csr(filters)
        picks = np.empty(R, dtype=np.int32)
        scores = np.empty(R, dtype=np.float64)
        self._check(self._lib.eppk_pick_batch_subset(self._ctx, reqs.ctypes.data, R, keys.ctypes.data, off.ctypes.data, picks.ctypes.data,
                                                     scores.ctypes.data), "pick_batch_subset")
        return picks, scores

    def pick_topk(self, reqs: np.ndarray, k: int, mask: Optional[np.ndarray] = None) -> Tuple[np.ndarray, np.ndarray]:
        """Ordered fallbacks: ([R, k] candidate indices, [R, k] totals); column 0 is the pick (include/eppk.h eppk_pick_topk)."""
        if not 1 <= int(k) <= 8:
            raise EppkError(-1, "pick_topk: k out of range (1..8)")
        reqs = np.ascontiguousarray(reqs, dtype=np.uint64)
        assert reqs.ndim == 2 and reqs.shape[1] == self.row_words, "request row stride mismatch"
        R = reqs.shape[0]
        if R > self.max_batch:
            raise EppkError(-2, "pick_topk: more requests than max_batch")
        picks = np.full((R, k), -1, dtype=np.int32)
        scores = np.zeros((R, k), dtype=np.float64)
        m = None
        if mask is not None:
            m = np.ascontiguousarray(mask, dtype=np.uint64)
            assert m.shape == (R, (self.n_pods + 63) // 64), "mask shape mismatch"
        self._check(self._lib.eppk_pick_topk(self._ctx, reqs.ctypes.data, R, m.ctypes.data if m is not None else None, k,
                                             picks.ctypes.data, scores.ctypes.data), "pick_topk")
        return picks, scores

    def pick_random_topk(self, reqs: np.ndarray, k: int, seed: int, mask: Optional[np.ndarray] = None) -> Tuple[np.ndarray, np.ndarray]:
        """Picker "random-top-k" (examples/example.yaml:25; SEMANTICS.md §3b): seeded choice among each request's k best candidates."""
        reqs = np.ascontiguousarray(reqs, dtype=np.uint64)
        assert reqs.ndim == 2 and reqs.shape[1] == self.row_words, "request row stride mismatch"
        R = reqs.shape[0]
        picks = np.empty(R, dtype=np.int32)
        scores = np.empty(R, dtype=np.float64)
        mptr = None
        if mask is not None:
            mask = np.ascontiguousarray(mask, dtype=np.uint64)
            assert mask.shape == (R, (self.n_pods + 63) // 64), "mask shape mismatch"
            mptr = mask.ctypes.data
        self._check(self._lib.eppk_pick_random_topk(self._ctx, reqs.ctypes.data, R, mptr, k, seed & 0xFFFFFFFFFFFFFFFF, picks.ctypes.data,
                                                    scores.ctypes.data), "pick_random_topk")
        return picks, scores

    def set_assumed_load(self, epochs: int) -> None:
        """Assumed load in `epochs` sub-batches per batch (SEMANTICS.md §2b); 0 = off."""
        self._check(self._lib.eppk_set_assumed_load(self._ctx, int(epochs)), "set_assumed_load")

    def pick_device(self, d_reqs: int, n_reqs: int, d_mask: Optional[int], d_pick: int, d_score: Optional[int], stream: int = 0) -> None:
        """Device-pointer entry point (asynchronous on `stream`, a hipStream_t as int; 0 = the context's stream)."""
        self._check(self._lib.eppk_pick_batch_device(self._ctx, d_reqs, n_reqs, d_mask, d_pick, d_score, stream or None), "pick_batch_device")

    def pick_learn_device(self, d_reqs: int, n_reqs: int, d_mask: Optional[int], d_pick: int, d_score: Optional[int], stream: int = 0) -> None:
        """The pick and its post-route index update in one call (include/eppk.h eppk_pick_learn_device): pick_device followed by
        index_insert_picks_device on `stream`, the pick kernel telling the update which pairs it has already seen in the index."""
        self._check(self._lib.eppk_pick_learn_device(self._ctx, d_reqs, n_reqs, d_mask, d_pick, d_score, stream or None), "pick_learn_device")

    def pick_candidates_device(self, d_reqs: int, n_reqs: int, d_mask: int, k: int, d_pick: int, d_score: Optional[int], stream: int = 0) -> None:
        """Candidate-major kernel for masked batches with few candidates (include/eppk.h eppk_pick_batch_candidates_device):
        k = 1 the pick, k > 1 ordered fallbacks ([n_reqs, k] entries)."""
        self._check(self._lib.eppk_pick_batch_candidates_device(self._ctx, d_reqs, n_reqs, d_mask, k, d_pick, d_score, stream or None),
                    "pick_batch_candidates_device")

    def pick_candidates(self, reqs: np.ndarray, mask: np.ndarray, k: int = 1) -> Tuple[np.ndarray, np.ndarray]:
        """Host-array convenience around pick_candidates_device (uses torch for the device buffers): ([R, k] picks, [R, k] scores)."""
        import torch
        reqs = np.ascontiguousarray(reqs, dtype=np.uint64)
        mask = np.ascontiguousarray(mask, dtype=np.uint64)
        R = reqs.shape[0]
        assert reqs.ndim == 2 and reqs.shape[1] == self.row_words and mask.shape == (R, (self.n_pods + 63) // 64)
        dev = torch.device("cuda", self.device)
        d_reqs = torch.from_numpy(reqs.view(np.int64)).to(dev)
        d_mask = torch.from_numpy(mask.view(np.int64)).to(dev)
        d_pick = torch.empty((R, k), dtype=torch.int32, device=dev)
        d_score = torch.empty((R, k), dtype=torch.float64, device=dev)
        torch.cuda.synchronize(dev)
        self.pick_candidates_device(d_reqs.data_ptr(), R, d_mask.data_ptr(), k, d_pick.data_ptr(), d_score.data_ptr())
        torch.cuda.synchronize(dev)      # (the launch went to the context's stream: device-wide sync)
        return d_pick.cpu().numpy(), d_score.cpu().numpy()

    def launch_status(self) -> int:
        """Sticky flags of the *_device launches since the last call (include/eppk.h: EPPK_LAUNCH_BAD_REQUEST_ROW = 1,
        EPPK_LAUNCH_BAD_PICK = 2); synchronises the device."""
        f = C.c_uint32(0)
        self._check(self._lib.eppk_launch_status(self._ctx, C.byref(f)), "launch_status")
        return f.value

    def stream_wait_pick(self, waiting_stream: int) -> None:
        """Make `waiting_stream` (hipStream_t as int) wait for the most recent pick launch (include/eppk.h eppk_stream_wait_pick)."""
        self._check(self._lib.eppk_stream_wait_pick(self._ctx, waiting_stream), "stream_wait_pick")

    def index_insert_picks_device(self, d_reqs: int, d_picks: int, n_reqs: int, stream: int = 0) -> None:
        self._check(self._lib.eppk_index_insert_picks_device(self._ctx, d_reqs, d_picks, n_reqs, stream or None), "index_insert_picks_device")

    def hash_prompts_device(self, d_prompts: int, prompt_stride: int, d_prompt_len: int, d_seed: int, d_adapter: int, n_reqs: int,
                            block_chars: int, d_reqs_out: int, stream: int = 0) -> None:
        """Chain-hash a batch of prompts on the device into request rows (device pointers as ints)."""
        self._check(self._lib.eppk_hash_prompts_device(self._ctx, d_prompts, prompt_stride, d_prompt_len, d_seed, d_adapter, n_reqs,
                                                       block_chars, d_reqs_out, stream or None), "hash_prompts_device")

    def pick_endpoints(self, endpoints: Sequence[Endpoint], reqs: np.ndarray, mask: Optional[np.ndarray] = None,
                       fallbacks: int = 0) -> List[PickResult]:
        """Batched EndpointPicker.Pick: one PickResult per request; Unavailable if any request has no candidate.
        `fallbacks` > 0 also fills PickResult.Fallbacks (server.go:74) with the next-best endpoints, in order."""
        if len(endpoints) != self.n_pods:
            raise EppkError(-1, "endpoints must be the published snapshot's candidate slice")
        if fallbacks > 0:
            picks, _ = self.pick_topk(reqs, 1 + fallbacks, mask)
        else:
            picks = self.pick(reqs, mask)[0].reshape(-1, 1)
        out = []
        for row in picks:
            if row[0] < 0:
                raise Unavailable("no endpoints available")
            eps = [endpoints[int(p)] for p in row if p >= 0]
            out.append(PickResult(endpoint=join_host_port(eps[0].address, eps[0].port),
                                  fallbacks=[join_host_port(e.address, e.port) for e in eps[1:]]))
        return out

    def chain_is_fused(self) -> int:
        """0 = generic per-pair kernel, 1 = fused sparse kernel, 2 = fused with an interpreted tail (include/eppk.h)."""
        return int(self._lib.eppk_chain_is_fused(self._ctx))

    def resident_stats(self) -> Tuple[bool, int, int]:
        """(switch on?, small batches answered by the resident workgroup, times it was started) -- include/eppk.h EPPK_RESIDENT."""
        b, st = C.c_uint64(0), C.c_uint64(0)
        on = self._lib.eppk_resident_stats(self._ctx, C.byref(b), C.byref(st))
        return bool(on > 0), int(b.value), int(st.value)

    def quad_stats(self) -> Tuple[int, int]:
        """(pick launches that went through the four-requests-per-wavefront kernel, requests those launches deferred to the
        general kernel); synchronises the device (include/eppk.h eppk_quad_stats)."""
        a, b = C.c_uint64(0), C.c_uint64(0)
        self._check(self._lib.eppk_quad_stats(self._ctx, C.byref(a), C.byref(b)), "quad_stats")
        return int(a.value), int(b.value)

    # -- measurement ------------------------------------------------------------------------
    def profile(self, on) -> None:
        """False / 0: off; True / 1: every pick launch carries events and probe counts; N > 1: every Nth launch (sampled)."""
        self._check(self._lib.eppk_profile_enable(self._ctx, int(on)), "profile_enable")

    def profile_drain(self, cap: int = 65536) -> np.ndarray:
        ms = np.zeros(cap, dtype=np.float32)
        n = C.c_uint32(0)
        self._check(self._lib.eppk_profile_drain(self._ctx, ms.ctypes.data, cap, C.byref(n)), "profile_drain")
        return ms[: n.value].copy()

    def profile_bytes(self) -> Tuple[int, int, int]:
        """(algorithmic bytes, index look-ups, launches) accumulated since profile(True)."""
        b, p, n = C.c_uint64(0), C.c_uint64(0), C.c_uint32(0)
        self._check(self._lib.eppk_profile_bytes(self._ctx, C.byref(b), C.byref(p), C.byref(n)), "profile_bytes")
        return b.value, p.value, n.value


GATHER_PEER, GATHER_RCCL, GATHER_HOST = 0, 1, 2
GROUP_LEARN, GROUP_GATHER = 1, 2


class DeviceGroup:
    """One picker over several GPUs behind the C ABI (include/eppk.h "device groups"): replicated snapshot + prefix index,
    batches sharded by request, picks all-gathered on the devices only for the post-route index update (`learn`)."""

    def __init__(self, chain: Sequence[Tuple[int, int]], devices: Sequence[int], max_pods: int, max_blocks: int = 0, max_batch: int = 65536,
                 index_slots: int = 0, gather: int = GATHER_PEER, min_shard: Optional[int] = None) -> None:
        self._lib = _lib.load_library()
        cfg = _lib.Cfg()
        cfg.struct_size = C.sizeof(_lib.Cfg)
        cfg.device = 0
        cfg.max_pods, cfg.max_blocks, cfg.max_batch, cfg.index_slots = max_pods, max_blocks, max_batch, index_slots
        cfg.n_scorers = len(chain)
        for i, (kind, weight) in enumerate(chain):
            cfg.chain[i].kind = int(kind)
            cfg.chain[i].weight = int(weight)
        devs = (C.c_int32 * len(devices))(*[int(d) for d in devices])
        self._g = C.c_void_p()
        rc = self._lib.eppk_group_create(C.byref(cfg), devs, len(devices), gather, C.byref(self._g))
        if rc != 0:
            raise EppkError(rc, (self._lib.eppk_group_last_error(None) or b"").decode())
        self.n_pods, self.row_words, self.max_batch, self.max_pods = 0, 1 + max_blocks, max_batch, max_pods
        if min_shard is not None:
            self._check(self._lib.eppk_group_set_min_shard(self._g, int(min_shard)), "set_min_shard")

    def _check(self, rc: int, what: str) -> None:
        if rc != 0:
            raise EppkError(rc, f"{what}: {(self._lib.eppk_group_last_error(self._g) or b'').decode()}")

    def close(self) -> None:
        if getattr(self, "_g", None) is not None and self._g:
            self._lib.eppk_group_destroy(self._g)
            self._g = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def size(self) -> int:
        return int(self._lib.eppk_group_size(self._g))

    @property
    def ranks_seen(self) -> int:
        return int(self._lib.eppk_group_ranks_seen(self._g))

    def publish(self, pods: np.ndarray, epoch: int = 0) -> None:
        pods = np.ascontiguousarray(pods, dtype=POD_DTYPE)
        self._check(self._lib.eppk_group_snapshot_publish(self._g, pods.ctypes.data, pods.shape[0], epoch), "group_snapshot_publish")
        self.n_pods = int(pods.shape[0])

    def index_insert(self, hashes: np.ndarray, pods: np.ndarray) -> None:
        h = np.ascontiguousarray(hashes, dtype=np.uint64).ravel()
        p = np.ascontiguousarray(pods, dtype=np.uint32).ravel()
        self._check(self._lib.eppk_group_index_insert(self._g, h.ctypes.data, p.ctypes.data, h.shape[0]), "group_index_insert")

    def index_clear(self) -> None:
        self._check(self._lib.eppk_group_index_clear(self._g), "group_index_clear")

    def index_remove_pod(self, pod: int) -> None:
        self._check(self._lib.eppk_group_index_remove_pod(self._g, pod), "group_index_remove_pod")

    def index_advance_epoch(self) -> int:
        e = C.c_uint32(0)
        self._check(self._lib.eppk_group_index_advance_epoch(self._g, C.byref(e)), "group_index_advance_epoch")
        return e.value

    def index_evict_older(self, min_epoch: int) -> int:
        n = C.c_uint32(0)
        self._check(self._lib.eppk_group_index_evict_older(self._g, min_epoch, C.byref(n)), "group_index_evict_older")
        return n.value

    def pick(self, reqs: np.ndarray, mask: Optional[np.ndarray] = None, learn: bool = False, gather: bool = False) -> Tuple[np.ndarray, np.ndarray]:
        reqs = np.ascontiguousarray(reqs, dtype=np.uint64)
        R = reqs.shape[0]
        assert reqs.ndim == 2 and reqs.shape[1] == self.row_words, "request row stride mismatch"
        picks = np.empty(R, dtype=np.int32)
        scores = np.empty(R, dtype=np.float64)
        mptr = None
        if mask is not None:
            mask = np.ascontiguousarray(mask, dtype=np.uint64)
            assert mask.shape == (R, (self.n_pods + 63) // 64), "mask shape mismatch"
            mptr = mask.ctypes.data
        flags = (GROUP_LEARN if learn else 0) | (GROUP_GATHER if gather else 0)
        self._check(self._lib.eppk_group_pick_batch(self._g, reqs.ctypes.data, R, mptr, picks.ctypes.data, scores.ctypes.data, flags), "group_pick_batch")
        return picks, scores

    def index_evict_older_device(self, min_epoch: int) -> None:
        """eppk_group_index_evict_older_device: asynchronous on every member, valid between two stage_begin calls."""
        self._check(self._lib.eppk_group_index_evict_older_device(self._g, min_epoch), "group_index_evict_older_device")

    def index_trim_pods(self, cap_per_pod: int) -> int:
        n = C.c_uint64(0)
        self._check(self._lib.eppk_group_index_trim_pods(self._g, cap_per_pod, C.byref(n)), "group_index_trim_pods")
        return n.value

    def _rows_and_mask(self, reqs, mask):
        reqs = np.ascontiguousarray(reqs, dtype=np.uint64)
        assert reqs.ndim == 2 and reqs.shape[1] == self.row_words, "request row stride mismatch"
        mptr = None
        if mask is not None:
            mask = np.ascontiguousarray(mask, dtype=np.uint64)
            assert mask.shape == (reqs.shape[0], (self.n_pods + 63) // 64), "mask shape mismatch"
            mptr = mask.ctypes.data
        return reqs, mask, mptr

    def pick_topk(self, reqs: np.ndarray, k: int, mask: Optional[np.ndarray] = None) -> Tuple[np.ndarray, np.ndarray]:
        """eppk_group_pick_topk: ([R, k] candidate indices, [R, k] totals), column 0 = the pick."""
        reqs, mask, mptr = self._rows_and_mask(reqs, mask)
        R = reqs.shape[0]
        picks = np.full((R, max(int(k), 1)), -1, dtype=np.int32)
        scores = np.zeros((R, max(int(k), 1)), dtype=np.float64)
        self._check(self._lib.eppk_group_pick_topk(self._g, reqs.ctypes.data, R, mptr, k, picks.ctypes.data, scores.ctypes.data), "group_pick_topk")
        return picks, scores

    def pick_random_topk(self, reqs: np.ndarray, k: int, seed: int, mask: Optional[np.ndarray] = None) -> Tuple[np.ndarray, np.ndarray]:
        reqs, mask, mptr = self._rows_and_mask(reqs, mask)
        R = reqs.shape[0]
        picks = np.empty(R, dtype=np.int32)
        scores = np.empty(R, dtype=np.float64)
        self._check(self._lib.eppk_group_pick_random_topk(self._g, reqs.ctypes.data, R, mptr, k, seed & 0xFFFFFFFFFFFFFFFF, picks.ctypes.data,
                                                          scores.ctypes.data), "group_pick_random_topk")
        return picks, scores

    # -- the pipelined host path over the group (eppk_group_pick_stage_*) ---------------------------------------------
    def stage_buffers(self, which: int, with_mask: bool = False) -> Tuple[np.ndarray, Optional[np.ndarray]]:
        rp, mp = C.c_void_p(0), C.c_void_p(0)
        self._check(self._lib.eppk_group_pick_stage_buffers(self._g, which, C.byref(rp), C.byref(mp) if with_mask else None), "group_pick_stage_buffers")
        reqs = np.frombuffer((C.c_uint64 * (self.max_batch * self.row_words)).from_address(rp.value), dtype=np.uint64).reshape(self.max_batch, self.row_words)
        mask = None
        if with_mask:
            jmax = (self.max_pods + 63) // 64
            mask = np.frombuffer((C.c_uint64 * (self.max_batch * jmax)).from_address(mp.value), dtype=np.uint64)
        return reqs, mask

    def stage_begin(self, which: int, R: int, use_mask: bool = False, learn: bool = False) -> None:
        self._check(self._lib.eppk_group_pick_stage_begin(self._g, which, R, 1 if use_mask else 0, 1 if learn else 0), "group_pick_stage_begin")
        self._stage_n = getattr(self, "_stage_n", {})
        self._stage_n[which] = R

    def stage_end(self, which: int) -> Tuple[np.ndarray, np.ndarray]:
        R = getattr(self, "_stage_n", {}).get(which, 0)
        picks = np.empty(R, dtype=np.int32)
        scores = np.empty(R, dtype=np.float64)
        self._check(self._lib.eppk_group_pick_stage_end(self._g, which, picks.ctypes.data, scores.ctypes.data), "group_pick_stage_end")
        return picks, scores

    def member_launch_status(self, i: int) -> int:
        f = C.c_uint32(0)
        rc = self._lib.eppk_launch_status(self._lib.eppk_group_ctx(self._g, i), C.byref(f))
        if rc != 0:
            raise EppkError(rc, "launch_status")
        return f.value

    def member_index_size(self, i: int) -> int:
        n = C.c_uint32(0)
        rc = self._lib.eppk_index_size(self._lib.eppk_group_ctx(self._g, i), C.byref(n))
        if rc != 0:
            raise EppkError(rc, "index_size")
        return n.value

    def member_selfcheck(self, i: int) -> int:
        n = C.c_uint64(0)
        rc = self._lib.eppk_index_selfcheck(self._lib.eppk_group_ctx(self._g, i), C.byref(n))
        if rc != 0:
            raise EppkError(rc, "index_selfcheck")
        return n.value

    def member_pick(self, i: int, reqs: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        """eppk_pick_batch on member i alone (its replica of snapshot + index): diagnostics / tests."""
        reqs = np.ascontiguousarray(reqs, dtype=np.uint64)
        R = reqs.shape[0]
        picks = np.empty(R, dtype=np.int32)
        scores = np.empty(R, dtype=np.float64)
        rc = self._lib.eppk_pick_batch(self._lib.eppk_group_ctx(self._g, i), reqs.ctypes.data, R, None, picks.ctypes.data, scores.ctypes.data)
        if rc != 0:
            raise EppkError(rc, "member pick_batch")
        return picks, scores

    def device_picks_ptr(self, i: int) -> int:
        return int(self._lib.eppk_group_device_picks(self._g, i) or 0)

    def pick_device(self, d_reqs: Sequence[int], n_rows: Sequence[int], d_picks: Sequence[int], d_scores: Optional[Sequence[int]] = None,
                    d_gathered: Optional[Sequence[int]] = None) -> None:
        """eppk_group_pick_device: member i scores the n_rows[i] rows at device pointer d_reqs[i] (its own memory) into d_picks[i]
        (/ d_scores[i]); with `d_gathered` every member also receives all picks, member-major.  Asynchronous: `sync()` waits."""
        n = len(d_reqs)
        arr = lambda xs: (C.c_void_p * n)(*[C.c_void_p(int(x)) for x in xs])
        self._check(self._lib.eppk_group_pick_device(self._g, arr(d_reqs), (C.c_uint32 * n)(*[int(x) for x in n_rows]), arr(d_picks),
                                                     arr(d_scores) if d_scores is not None else None,
                                                     arr(d_gathered) if d_gathered is not None else None, 2 if d_gathered is not None else 0), "group_pick_device")

    def sync(self) -> None:
        self._check(self._lib.eppk_group_sync(self._g), "group_sync")

    def member_stream(self, i: int) -> int:
        return int(self._lib.eppk_group_stream(self._g, i) or 0)
