#!/usr/bin/env python3
"""Generate tests/golden/cases.npz — golden input/output vectors for the pick path.

The reference snapshot holds NO golden vectors for the scorer chain (parity
unpinned), and its Go source cannot be imported here.  These vectors therefore come from a SECOND,
independent restatement of SEMANTICS.md written in numpy matrix form (whole [R x P] score matrices per
scorer, dict-of-sets prefix index, np.argmax first-maximum), structurally unlike oracle/oracle.c
(per-request loops) and unlike the HIP kernel (lane-transposed bit sets).  Agreement of all three on
these vectors is the strongest pin available until llm-d-router's source can be diffed.

Hash known-answers are produced with python-xxhash (an independent XXH64 implementation).

Run:  python tests/golden/gen_golden.py     (writes cases.npz next to this file; deterministic)
"""
import os

import numpy as np

Q, KV, L, PF = 1, 2, 3, 4
POD_DTYPE = np.dtype([("queue", "<u4"), ("running", "<u4"), ("kv_util", "<f8"), ("max_lora", "<u4"),
                      ("flags", "<u4"), ("active", "<u8", (2,)), ("waiting", "<u8", (2,)), ("reserved", "<u8")])


def clamp01(s):
    out = np.where(s > 1.0, 1.0, s)
    return np.where(out >= 0.0, out, 0.0)   # NaN and negatives -> 0


def bits128(words, a):
    """words [P,2] u64, a scalar adapter id -> bool [P]"""
    return ((words[:, a >> 6] >> np.uint64(a & 63)) & np.uint64(1)).astype(bool)


def popcount128(words):
    b = np.unpackbits(words.view(np.uint8).reshape(words.shape[0], -1), axis=1)
    return b.sum(axis=1).astype(np.int64)


def numpy_pick(chain, pods, index, adapter, n_blocks, hashes, mask):
    """SEMANTICS.md §2-3 in matrix form. index: dict hash -> set(pod). mask: [R, W] u64 or None."""
    total, cand = numpy_totals(chain, pods, index, adapter, n_blocks, hashes, mask)
    R, P = total.shape
    masked_total = np.where(cand, total, -np.inf)
    pick = np.argmax(masked_total, axis=1).astype(np.int32) if P else np.zeros(R, np.int32)
    has = cand.any(axis=1) if P else np.zeros(R, bool)
    score = np.where(has, total[np.arange(R), pick] if P else 0.0, 0.0)
    pick = np.where(has, pick, -1).astype(np.int32)
    return pick, score.astype(np.float64)


def numpy_topk(chain, pods, index, adapter, n_blocks, hashes, mask, k):
    """Ordered fallbacks (SEMANTICS.md §3a): the k best candidates under (total descending, index ascending), -1 / 0.0 padded."""
    total, cand = numpy_totals(chain, pods, index, adapter, n_blocks, hashes, mask)
    R, P = total.shape
    picks = np.full((R, k), -1, dtype=np.int32)
    scores = np.zeros((R, k), dtype=np.float64)
    for r in range(R):
        c = np.nonzero(cand[r])[0]
        order = c[np.argsort(-total[r, c], kind="stable")][:k]   # stable: equal totals stay in index order
        picks[r, :order.size] = order
        scores[r, :order.size] = total[r, order]
    return picks, scores


def numpy_totals(chain, pods, index, adapter, n_blocks, hashes, mask):
    """Weighted totals [R, P] (binary64, scorers added in chain order) and the candidate matrix [R, P]."""
    R, P = adapter.shape[0], pods.shape[0]
    if mask is None:
        cand = np.ones((R, P), dtype=bool)
    else:
        bits = np.unpackbits(mask.view(np.uint8).reshape(R, -1), axis=1, bitorder="little")
        cand = bits[:, :P].astype(bool)
    cand = cand & ((pods["flags"] & 1) == 0)[None, :]          # SEMANTICS.md §6b: a hole of the snapshot is never a candidate
    total = np.zeros((R, P), dtype=np.float64)
    q = pods["queue"].astype(np.int64)
    for kind, w in chain:
        if kind == Q:
            big = np.int64(1) << 40
            mn = np.where(cand, q[None, :], big).min(axis=1, initial=big)
            mx = np.where(cand, q[None, :], -1).max(axis=1, initial=-1)
            den = (mx - mn).astype(np.float64)
            with np.errstate(divide="ignore", invalid="ignore"):
                s = (mx[:, None] - q[None, :]).astype(np.float64) / den[:, None]
            s = np.where((mx == mn)[:, None], 1.0, s)
        elif kind == KV:
            s = np.broadcast_to(1.0 - pods["kv_util"][None, :], (R, P))
        elif kind == L:
            loaded = popcount128(pods["active"]) + popcount128(pods["waiting"])
            free = loaded < pods["max_lora"].astype(np.int64)
            s = np.zeros((R, P))
            for r in range(R):
                a = int(adapter[r])
                act = bits128(pods["active"], a) if a >= 0 else np.zeros(P, bool)
                wai = bits128(pods["waiting"], a) if a >= 0 else np.zeros(P, bool)
                s[r] = np.where(act, 1.0, np.where(free, 0.8, np.where(wai, 0.6, 0.0)))
        elif kind == PF:
            matched = np.zeros((R, P), dtype=np.int64)
            for r in range(R):
                for i in range(int(n_blocks[r])):
                    servers = index.get(int(hashes[r, i]), set())
                    if not servers:
                        break
                    for p in servers:
                        if p < P:
                            matched[r, p] += 1
            nb = n_blocks.astype(np.float64)[:, None]
            with np.errstate(divide="ignore", invalid="ignore"):
                s = matched.astype(np.float64) / nb
            s = np.where(nb == 0, 0.0, s)
        else:
            raise ValueError(kind)
        total = total + clamp01(s) * float(w)
    return total, cand


def rand_pods(rng, P, A=128, tie_heavy=False):
    pods = np.zeros(P, dtype=POD_DTYPE)
    pods["queue"] = rng.integers(0, 4 if tie_heavy else 64, P)
    pods["running"] = rng.integers(0, 256, P)
    pods["kv_util"] = rng.integers(0, 5 if tie_heavy else 1025, P) / (4.0 if tie_heavy else 1024.0)
    pods["max_lora"] = rng.choice([4, 8], P)
    for p in range(P):
        for a in rng.integers(0, A, rng.integers(0, int(pods["max_lora"][p]) + 1)):
            pods["active"][p, a >> 6] |= np.uint64(1) << np.uint64(a & 63)
        for a in rng.integers(0, A, rng.integers(0, 3)):
            pods["waiting"][p, a >> 6] |= np.uint64(1) << np.uint64(a & 63)
    return pods


def rand_case(rng, R, P, B, chain, masked, n_groups=6, tie_heavy=False, removed=(), holes=()):
    pods = rand_pods(rng, P, tie_heavy=tie_heavy)
    for p in holes:                              # SEMANTICS.md §6b: holes are out of every candidate set AND out of the index
        pods["flags"][p] |= 1
    removed = tuple(removed) + tuple(holes)
    group_hash = rng.integers(1, 2**63, (n_groups, max(B, 1)), dtype=np.uint64)
    adapter = rng.integers(-1, 128, R).astype(np.int32)
    n_blocks = rng.integers(0, B + 1, R).astype(np.uint32) if B else np.zeros(R, np.uint32)
    hashes = rng.integers(1, 2**63, (R, max(B, 1)), dtype=np.uint64)
    for r in range(R):
        g = rng.integers(0, n_groups)
        share = rng.integers(0, B + 1) if B else 0
        hashes[r, :share] = group_hash[g, :share]
    # index: each group's chain cached on a few pods, some only partially (interior gaps included)
    ih, ip = [], []
    for g in range(n_groups):
        for p in rng.choice(P, size=min(P, 5), replace=False):
            depth = rng.integers(1, B + 1) if B else 0
            for i in range(depth):
                if rng.random() < 0.9:           # 10 % interior holes
                    ih.append(group_hash[g, i]); ip.append(p)
    ih = np.array(ih, dtype=np.uint64); ip = np.array(ip, dtype=np.uint32)
    mask = None
    if masked:
        W = (P + 63) // 64
        mask = rng.integers(0, 2**63, (R, W), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, (R, W), dtype=np.uint64)
        if P % 64:
            mask[:, -1] &= np.uint64((1 << (P % 64)) - 1)
        mask[0, :] = 0                           # a request with no candidates
    index = {}
    for h, p in zip(ih.tolist(), ip.tolist()):
        index.setdefault(h, set()).add(p)
    for p in removed:                            # SEMANTICS.md §6: removal leaves empty sets behind
        for s in index.values():
            s.discard(p)
    pick, score = numpy_pick(chain, pods, index, adapter, n_blocks, hashes, mask)
    return dict(chain=np.array(chain, dtype=np.int64).reshape(-1, 2), pods=pods, adapter=adapter, n_blocks=n_blocks,
                hashes=hashes, mask=mask if mask is not None else np.zeros((0, 0), np.uint64), index_hashes=ih, index_pods=ip,
                removed=np.array(list(removed), dtype=np.uint32), B=np.int64(B), pick=pick, score=score)


def main():
    rng = np.random.Generator(np.random.PCG64(20260821))
    FULL = [(Q, 2), (KV, 2), (L, 1), (PF, 3)]
    cases = {
        "full_small": rand_case(rng, 64, 100, 8, FULL, False),
        "full_masked": rand_case(rng, 64, 100, 8, FULL, True),
        "ties": rand_case(rng, 48, 70, 4, FULL, True, tie_heavy=True),
        "noncanonical_negw": rand_case(rng, 48, 130, 6, [(L, -1), (Q, 3), (PF, 4), (KV, -2), (PF, 1)], True),
        "queue_kv_only": rand_case(rng, 128, 16, 0, [(Q, 1), (KV, 1)], False),
        "removed_pods": rand_case(rng, 64, 40, 8, [(PF, 5), (KV, 1)], False, removed=tuple(range(0, 40, 2))),
        "one_pod": rand_case(rng, 8, 1, 3, FULL, False),
        "p64_exact": rand_case(rng, 32, 64, 5, FULL, True),
        "long_chains": rand_case(rng, 24, 90, 70, FULL, False, n_groups=3),   # > 63 blocks: wide counters
    }
    # out-of-range gauges: clamp01 and NaN handling
    c = rand_case(rng, 16, 12, 2, [(KV, 3), (Q, 1)], False)
    c["pods"]["kv_util"][:4] = [1.5, -0.25, np.nan, 0.0]
    idx = {}
    c["pick"], c["score"] = numpy_pick([(KV, 3), (Q, 1)], c["pods"], idx, c["adapter"], np.zeros(16, np.uint32), c["hashes"], None)
    c["n_blocks"] = np.zeros(16, np.uint32)
    c["index_hashes"] = np.zeros(0, np.uint64); c["index_pods"] = np.zeros(0, np.uint32)
    cases["clamp_nan"] = c

    flat = {}
    for name, case in cases.items():
        for k, v in case.items():
            flat[f"{name}/{k}"] = v

    # XXH64 known answers from python-xxhash (independent implementation of the published algorithm)
    import xxhash
    msgs = [b"", b"a", b"abc", b"0123456789abcdef0123456789abcde", b"0123456789abcdef0123456789abcdef",
            bytes(range(256)) * 3 + b"tail-7!"]
    flat["xxh64/lens"] = np.array([len(m) for m in msgs], dtype=np.int64)
    flat["xxh64/data"] = np.frombuffer(b"".join(msgs), dtype=np.uint8)
    flat["xxh64/seed0"] = np.array([xxhash.xxh64(m, seed=0).intdigest() for m in msgs], dtype=np.uint64)
    flat["xxh64/seed_2a"] = np.array([xxhash.xxh64(m, seed=0x2A).intdigest() for m in msgs], dtype=np.uint64)
    # chain-hash KAT (SEMANTICS.md §4): model "m", prompt 200 bytes, 64-byte blocks -> 3 hashes
    prompt = bytes((i * 7 + 3) & 0xFF for i in range(200))
    prev = xxhash.xxh64(b"m", seed=0).intdigest()
    chain = []
    for i in range(3):
        prev = xxhash.xxh64(prompt[i * 64:(i + 1) * 64] + prev.to_bytes(8, "little"), seed=0).intdigest()
        chain.append(prev)
    flat["chain/prompt"] = np.frombuffer(prompt, dtype=np.uint8)
    flat["chain/expected"] = np.array(chain, dtype=np.uint64)

    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cases.npz")
    np.savez_compressed(out, **flat)
    print("wrote", out, os.path.getsize(out), "bytes;", len(cases), "pick cases")


if __name__ == "__main__":
    main()
