"""ctypes binding of oracle/liboracle.so — TEST INFRASTRUCTURE ONLY.

May be imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the
product package.  PARITY UNPINNED (see oracle.h): restates SEMANTICS.md.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build() -> str:
    subprocess.run(["make", "-s", "-C", _HERE], check=True)
    return os.path.join(_HERE, "liboracle.so")


def load() -> C.CDLL:
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.path.join(_HERE, "liboracle.so")
    if not os.path.exists(path):
        build()
    lib = C.CDLL(path)
    vp, u32, u64 = C.c_void_p, C.c_uint32, C.c_uint64
    lib.orc_index_new.restype = vp
    lib.orc_index_free.argtypes = [vp]
    lib.orc_index_free.restype = None
    lib.orc_index_clear.argtypes = [vp]
    lib.orc_index_clear.restype = None
    lib.orc_index_insert.argtypes = [vp, u64, u32]
    lib.orc_index_insert.restype = None
    lib.orc_index_remove_pod.argtypes = [vp, u32]
    lib.orc_index_remove_pod.restype = None
    lib.orc_index_scrub_inactive.argtypes = [vp, vp, u32]
    lib.orc_index_scrub_inactive.restype = None
    lib.orc_index_advance_epoch.argtypes = [vp]
    lib.orc_index_advance_epoch.restype = u32
    lib.orc_index_evict_older.argtypes = [vp, u32]
    lib.orc_index_evict_older.restype = u32
    lib.orc_index_trim_pods.argtypes = [vp, u32, u32]
    lib.orc_index_trim_pods.restype = u64
    lib.orc_index_size.argtypes = [vp]
    lib.orc_index_size.restype = u64
    lib.orc_index_lookup.argtypes = [vp, u64, vp, u32]
    lib.orc_index_lookup.restype = u32
    lib.orc_pick_batch.argtypes = [vp, u32, vp, u32, vp, vp, u32, u32, vp, vp, vp, vp]
    lib.orc_pick_batch_mt.argtypes = [vp, u32, vp, u32, vp, vp, u32, u32, vp, vp, vp, C.c_int]
    lib.orc_score_row.argtypes = [vp, u32, vp, u32, vp, vp, vp, vp]
    lib.orc_pick_batch_assumed.argtypes = [vp, u32, vp, u32, vp, vp, u32, u32, vp, u32, vp, vp]
    lib.orc_pick_random_topk.argtypes = [vp, u32, vp, u32, vp, vp, u32, u32, vp, u32, u64, vp, vp]
    lib.orc_pick_topk.argtypes = [vp, u32, vp, u32, vp, vp, u32, u32, vp, u32, C.c_int, vp, vp]
    lib.orc_pick_topk.restype = C.c_int
    lib.orc_index_insert_picks.argtypes = [vp, vp, u32, u32, vp]
    lib.orc_tables_new.argtypes = [vp, u32, vp, u32]
    lib.orc_tables_new.restype = vp
    lib.orc_tables_free.argtypes = [vp]
    lib.orc_tables_free.restype = None
    lib.orc_pick_batch_sparse.argtypes = [vp, vp, vp, u32, u32, vp, vp, C.c_int]
    lib.orc_index_insert_picks.restype = None
    lib.orc_xxh64.argtypes = [vp, C.c_size_t, u64]
    lib.orc_xxh64.restype = u64
    lib.orc_hash_prompt.argtypes = [vp, C.c_size_t, vp, C.c_size_t, u32, vp, u32]
    lib.orc_round_robin.argtypes = [C.POINTER(u64), u32]
    lib.orc_round_robin.restype = C.c_int32
    lib.orc_subset_mask.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), u32, C.c_char_p, vp]
    _LIB = lib
    return lib


def _chain_array(chain: Sequence[Tuple[int, int]]) -> np.ndarray:
    arr = np.zeros(len(chain), dtype=[("kind", "<u4"), ("weight", "<i4")])
    for i, (k, w) in enumerate(chain):
        arr[i] = (int(k), int(w))
    return arr


class OracleIndex:
    def __init__(self) -> None:
        self.lib = load()
        self.h = C.c_void_p(self.lib.orc_index_new())

    def insert(self, hashes, pods, snapshot=None) -> None:
        """`snapshot` (pod rows): the published snapshot -- pairs that name one of its holes are ignored (SEMANTICS.md 6b)."""
        hole = None
        if snapshot is not None:
            hole = (np.ascontiguousarray(snapshot)["flags"] & 1).astype(bool)
        for h, p in zip(np.asarray(hashes, dtype=np.uint64).ravel().tolist(), np.asarray(pods, dtype=np.uint32).ravel().tolist()):
            if hole is not None and p < hole.shape[0] and hole[p]:
                continue
            self.lib.orc_index_insert(self.h, h, p)

    def scrub_inactive(self, snapshot) -> None:
        """The index side of publishing `snapshot`: every slot that is a hole in it is forgotten (SEMANTICS.md 6b)."""
        s = np.ascontiguousarray(snapshot)
        self.lib.orc_index_scrub_inactive(self.h, s.ctypes.data, s.shape[0])

    def remove_pod(self, pod: int) -> None:
        self.lib.orc_index_remove_pod(self.h, pod)

    def clear(self) -> None:
        self.lib.orc_index_clear(self.h)

    def advance_epoch(self) -> int:
        return int(self.lib.orc_index_advance_epoch(self.h))

    def evict_older(self, min_epoch: int) -> int:
        return int(self.lib.orc_index_evict_older(self.h, min_epoch))

    def trim_pods(self, n_pods_max: int, cap: int) -> int:
        """Per-pod capacity (SEMANTICS.md 6c); returns the (hash, pod) pairs removed."""
        return int(self.lib.orc_index_trim_pods(self.h, n_pods_max, cap))

    def size(self) -> int:
        return int(self.lib.orc_index_size(self.h))

    def lookup(self, h: int, cap: int = 4096) -> np.ndarray:
        out = np.zeros(cap, dtype=np.uint32)
        n = self.lib.orc_index_lookup(self.h, h, out.ctypes.data, cap)
        return out[:n]

    def insert_picks(self, reqs: np.ndarray, max_blocks: int, picks: np.ndarray) -> None:
        reqs = np.ascontiguousarray(reqs, dtype=np.uint64)
        picks = np.ascontiguousarray(picks, dtype=np.int32)
        self.lib.orc_index_insert_picks(self.h, reqs.ctypes.data, max_blocks, reqs.shape[0], picks.ctypes.data)

    def __del__(self):
        try:
            self.lib.orc_index_free(self.h)
        except Exception:
            pass


def pick_batch(chain, pods: np.ndarray, index: Optional[OracleIndex], reqs: np.ndarray, max_blocks: int,
               mask: Optional[np.ndarray] = None, threads: int = 0):
    """Sequential per-request Schedule() on the CPU. Returns (picks i32, scores f64, probes u32)."""
    lib = load()
    ch = _chain_array(chain)
    pods = np.ascontiguousarray(pods)
    assert pods.dtype.itemsize == 64
    reqs = np.ascontiguousarray(reqs, dtype=np.uint64)
    R = reqs.shape[0]
    assert reqs.shape[1] == 1 + max_blocks
    picks = np.empty(R, dtype=np.int32)
    scores = np.empty(R, dtype=np.float64)
    probes = np.zeros(R, dtype=np.uint32)
    mptr = None
    if mask is not None:
        mask = np.ascontiguousarray(mask, dtype=np.uint64)
        mptr = mask.ctypes.data
    ih = index.h if index is not None else None
    if threads and threads > 1:
        rc = lib.orc_pick_batch_mt(ch.ctypes.data, len(chain), pods.ctypes.data, pods.shape[0], ih, reqs.ctypes.data,
                                   max_blocks, R, mptr, picks.ctypes.data, scores.ctypes.data, threads)
    else:
        rc = lib.orc_pick_batch(ch.ctypes.data, len(chain), pods.ctypes.data, pods.shape[0], ih, reqs.ctypes.data,
                                max_blocks, R, mptr, picks.ctypes.data, scores.ctypes.data, probes.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"oracle rc={rc}")
    return picks, scores, probes


class OracleTables:
    """Per (chain, snapshot) tables of the second CPU algorithm (oracle.c: orc_pick_batch_sparse); keeps `pods` alive."""

    def __init__(self, chain, pods: np.ndarray) -> None:
        self._lib = load()
        self._chain = _chain_array(chain)
        self._pods = np.ascontiguousarray(pods)
        assert self._pods.dtype.itemsize == 64
        self.h = self._lib.orc_tables_new(self._chain.ctypes.data, len(chain), self._pods.ctypes.data, self._pods.shape[0])
        if not self.h:
            raise RuntimeError("orc_tables_new failed")

    def pick_batch(self, index: Optional[OracleIndex], reqs: np.ndarray, max_blocks: int, threads: int = 1):
        """Unmasked batch -> (picks i32, scores f64); bit-identical to pick_batch() by construction, held so by the tests."""
        reqs = np.ascontiguousarray(reqs, dtype=np.uint64)
        R = reqs.shape[0]
        assert reqs.shape[1] == 1 + max_blocks
        picks = np.empty(R, dtype=np.int32)
        scores = np.empty(R, dtype=np.float64)
        rc = self._lib.orc_pick_batch_sparse(self.h, index.h if index is not None else None, reqs.ctypes.data, max_blocks, R,
                                             picks.ctypes.data, scores.ctypes.data, int(threads))
        if rc != 0:
            raise RuntimeError(f"oracle rc={rc}")
        return picks, scores

    def __del__(self):
        try:
            if self.h:
                self._lib.orc_tables_free(self.h)
                self.h = None
        except Exception:
            pass


def score_row(chain, pods: np.ndarray, index: Optional[OracleIndex], req_row: np.ndarray, mask_row: Optional[np.ndarray] = None) -> np.ndarray:
    lib = load()
    ch = _chain_array(chain)
    pods = np.ascontiguousarray(pods)
    req_row = np.ascontiguousarray(req_row, dtype=np.uint64)
    out = np.empty(pods.shape[0], dtype=np.float64)
    mptr = None
    if mask_row is not None:
        mask_row = np.ascontiguousarray(mask_row, dtype=np.uint64)
        mptr = mask_row.ctypes.data
    rc = lib.orc_score_row(ch.ctypes.data, len(chain), pods.ctypes.data, pods.shape[0], index.h if index else None,
                           req_row.ctypes.data, mptr, out.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"oracle rc={rc}")
    return out


def pick_batch_assumed(chain, pods: np.ndarray, index: Optional[OracleIndex], reqs: np.ndarray, max_blocks: int, epochs: int,
                       mask: Optional[np.ndarray] = None):
    """SEMANTICS.md 2b: `pods` is MUTATED (queue += assumed load) -- pass the same array to the next batch of the snapshot."""
    lib = load()
    ch = _chain_array(chain)
    assert pods.flags["C_CONTIGUOUS"] and pods.dtype.itemsize == 64
    reqs = np.ascontiguousarray(reqs, dtype=np.uint64)
    R = reqs.shape[0]
    picks = np.empty(R, dtype=np.int32)
    scores = np.empty(R, dtype=np.float64)
    mptr = None
    if mask is not None:
        mask = np.ascontiguousarray(mask, dtype=np.uint64)
        mptr = mask.ctypes.data
    rc = lib.orc_pick_batch_assumed(ch.ctypes.data, len(chain), pods.ctypes.data, pods.shape[0], index.h if index is not None else None,
                                    reqs.ctypes.data, max_blocks, R, mptr, epochs, picks.ctypes.data, scores.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"oracle rc={rc}")
    return picks, scores


def pick_random_topk(chain, pods: np.ndarray, index: Optional[OracleIndex], reqs: np.ndarray, max_blocks: int, k: int, seed: int,
                     mask: Optional[np.ndarray] = None):
    """SEMANTICS.md 3b (picker random-top-k)."""
    lib = load()
    ch = _chain_array(chain)
    pods = np.ascontiguousarray(pods)
    reqs = np.ascontiguousarray(reqs, dtype=np.uint64)
    R = reqs.shape[0]
    picks = np.empty(R, dtype=np.int32)
    scores = np.empty(R, dtype=np.float64)
    mptr = None
    if mask is not None:
        mask = np.ascontiguousarray(mask, dtype=np.uint64)
        mptr = mask.ctypes.data
    rc = lib.orc_pick_random_topk(ch.ctypes.data, len(chain), pods.ctypes.data, pods.shape[0], index.h if index is not None else None,
                                  reqs.ctypes.data, max_blocks, R, mptr, k, seed & 0xFFFFFFFFFFFFFFFF, picks.ctypes.data, scores.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"oracle rc={rc}")
    return picks, scores


def pick_topk_batch(chain, pods: np.ndarray, index: Optional[OracleIndex], reqs: np.ndarray, max_blocks: int, k: int,
                    mask: Optional[np.ndarray] = None, threads: int = 1):
    """Ordered fallbacks of a whole batch in C (orc_pick_topk: k selection passes per request, `threads` host threads): what checks
    fallback lists at full batch sizes.  tests/test_oracle_golden.py holds it equal to `pick_topk` below (an independent formulation)."""
    lib = load()
    ch = _chain_array(chain)
    pods = np.ascontiguousarray(pods)
    reqs = np.ascontiguousarray(reqs, dtype=np.uint64)
    R = reqs.shape[0]
    picks = np.empty((R, k), dtype=np.int32)
    scores = np.empty((R, k), dtype=np.float64)
    mptr = None
    if mask is not None:
        mask = np.ascontiguousarray(mask, dtype=np.uint64)
        mptr = mask.ctypes.data
    rc = lib.orc_pick_topk(ch.ctypes.data, len(chain), pods.ctypes.data, pods.shape[0], index.h if index is not None else None,
                           reqs.ctypes.data, max_blocks, R, mptr, k, threads, picks.ctypes.data, scores.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"oracle rc={rc}")
    return picks, scores


def pick_topk(chain, pods: np.ndarray, index: Optional[OracleIndex], reqs: np.ndarray, k: int,
              mask: Optional[np.ndarray] = None):
    """Ordered fallbacks (SEMANTICS.md §2a): the k best candidates of every request under (total desc, index asc), from the
    oracle's own per-request totals (orc_score_row); padded with -1 / 0.0."""
    reqs = np.ascontiguousarray(reqs, dtype=np.uint64)
    R = reqs.shape[0]
    picks = np.full((R, k), -1, dtype=np.int32)
    scores = np.zeros((R, k), dtype=np.float64)
    for r in range(R):
        tot = score_row(chain, pods, index, reqs[r], None if mask is None else mask[r])
        cand = np.nonzero(~np.isnan(tot))[0]
        order = cand[np.lexsort((cand, -tot[cand]))][:k]      # primary: total descending; ties: lowest index
        picks[r, :order.size] = order
        scores[r, :order.size] = tot[order]
    return picks, scores


def xxh64(data: bytes, seed: int = 0) -> int:
    return int(load().orc_xxh64(data, len(data), seed))


def hash_prompt(model: bytes, prompt: bytes, block_chars: int, max_blocks: int) -> np.ndarray:
    out = np.zeros(max(max_blocks, 1), dtype=np.uint64)
    n = load().orc_hash_prompt(model, len(model), prompt, len(prompt), block_chars, out.ctypes.data, max_blocks)
    assert n >= 0
    return out[:n]


def round_robin(counter: C.c_uint64, n: int) -> int:
    return int(load().orc_round_robin(C.byref(counter), n))


def subset_mask(addrs: Sequence[str], ports: Sequence[str], filt: Optional[str]):
    n = len(addrs)
    a = (C.c_char_p * max(n, 1))(*[s.encode() for s in addrs])
    p = (C.c_char_p * max(n, 1))(*[s.encode() for s in ports])
    mask = np.zeros(max((n + 63) // 64, 1), dtype=np.uint64)
    rc = load().orc_subset_mask(a, p, n, None if filt is None else filt.encode(), mask.ctypes.data)
    return mask[: (n + 63) // 64], rc
