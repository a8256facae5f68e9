"""Parity tests proper: HIP path (through the C ABI) vs oracle/ on the same seeded inputs.

Bar: picks identical, scores BITWISE identical (binary64), for every BASELINE.json config shape,
with and without candidate masks, canonical and non-canonical chain orders.
"""
import os
import dataclasses

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

Q, KV, L, PF = 1, 2, 3, 4


def run_both(pkg, orc, wl, chain=None, mask=None, max_pods=None):
    chain = chain or wl.chain
    with pkg.BatchedPicker(chain, max_pods=max_pods or max(wl.P, 1), max_blocks=wl.B, max_batch=max(wl.R, 1),
                           index_slots=wl.index_slots) as pk:
        pk.publish(wl.pods)
        if wl.index_slots:
            pk.index_insert(wl.index_hashes, wl.index_pods)
        picks, scores = pk.pick(wl.reqs, mask)
    oix = orc.OracleIndex()
    oix.insert(wl.index_hashes, wl.index_pods)
    opicks, oscores, _ = orc.pick_batch(chain, wl.pods, oix, wl.reqs, wl.B, mask)
    return picks, scores, opicks, oscores


def assert_same(picks, scores, opicks, oscores):
    bad = np.nonzero(picks != opicks)[0]
    assert bad.size == 0, f"{bad.size} picks differ, first at {bad[:5]}: gpu {picks[bad[:5]]} oracle {opicks[bad[:5]]}"
    sb = np.nonzero(scores.view(np.uint64) != oscores.view(np.uint64))[0]
    assert sb.size == 0, f"{sb.size} scores differ bitwise, first {sb[:5]}: {scores[sb[:5]]} vs {oscores[sb[:5]]}"


@pytest.mark.parametrize("config,R,P", [(1, 128, 16), (2, 4096, 256), (3, 2048, 1024), (4, 2048, 2048), (5, 1024, 4096)])
def test_config_shapes_unmasked(pkg, orc, config, R, P):
    wl = pkg.workload.make_workload(config, R=R, P=P)
    assert_same(*run_both(pkg, orc, wl))


@pytest.mark.parametrize("config,R,P", [(1, 128, 16), (2, 1024, 256), (3, 1024, 1000), (4, 512, 2048), (5, 512, 4096)])
def test_config_shapes_masked(pkg, orc, config, R, P):
    wl = pkg.workload.make_workload(config, R=R, P=P, masked=True)
    assert_same(*run_both(pkg, orc, wl, mask=wl.mask))


@pytest.mark.parametrize("P", [1, 63, 64, 65, 1000, 1024, 1025, 2047, 2048, 2049, 4095, 4096])
def test_ragged_pod_counts(pkg, orc, P):
    wl = pkg.workload.make_workload(5, R=192, P=P)
    assert_same(*run_both(pkg, orc, wl))


@pytest.mark.parametrize("chain", [
    [(PF, 3), (L, 1)],                    # canonical, prefix first
    [(L, 1), (PF, 3)],
    [(PF, 3)],
    [(L, 2)],
    [(KV, 5)],
    [(L, 1), (Q, 2), (PF, 3), (KV, 2)],   # pod-only scorers between / behind LORA and PREFIX -> interpreted tail
    [(PF, 3), (KV, 5)],                   # the reference example's decode profile (0845-…/examples/example.yaml:21-25)
    [(PF, 3), (Q, 1), (KV, 1)],
    [(KV, 2), (PF, 4), (Q, -3), (L, 2)],
    [(L, 1), (Q, 2), (KV, 2), (Q, 1), (PF, 3)],   # three trailing pod-only scorers -> generic kernel
    [(PF, 3), (Q, 1), (PF, 3)],           # duplicates
    [(Q, -2), (KV, 3), (L, -1), (PF, 4)], # negative weights
    [],                                   # no scorers: every total is +0.0, pick = first candidate
])
def test_chain_orders(pkg, orc, chain):
    wl = pkg.workload.make_workload(3, R=512, P=777)
    assert_same(*run_both(pkg, orc, wl, chain=chain, max_pods=1024))
    wlm = pkg.workload.make_workload(3, R=256, P=777, masked=True)
    assert_same(*run_both(pkg, orc, wlm, chain=chain, mask=wlm.mask, max_pods=1024))


def test_empty_and_single_candidate_masks(pkg, orc):
    wl = pkg.workload.make_workload(5, R=64, P=300, masked=True)
    wl.mask[0, :] = 0                       # no candidate -> NO_PICK, score 0
    wl.mask[1, :] = 0
    wl.mask[1, 3] = np.uint64(1) << np.uint64(17)   # exactly pod 3*64+17
    picks, scores, opicks, oscores = run_both(pkg, orc, wl, mask=wl.mask, max_pods=4096)
    assert picks[0] == -1 and scores[0] == 0.0
    assert picks[1] == 3 * 64 + 17
    assert_same(picks, scores, opicks, oscores)


@pytest.mark.parametrize("cover", ["all", "top64", "half"])
def test_dense_fallback_when_prefix_is_cached_everywhere(pkg, orc, cover):
    """All 64 per-adapter table entries inside M -> the kernel must fall back to the dense scan."""
    wl = pkg.workload.make_workload(5, R=256, P=2500)
    rng = np.random.default_rng(7)
    # cache the first 3 blocks of EVERY request's chain on many pods
    if cover == "all":
        pods = np.arange(wl.P, dtype=np.uint32)
    elif cover == "half":
        pods = np.arange(0, wl.P, 2, dtype=np.uint32)
    else:
        pods = rng.choice(wl.P, 700, replace=False).astype(np.uint32)
    hs = np.unique(wl.reqs[:, 1:4].ravel())
    extra_h = np.repeat(hs, pods.size)
    extra_p = np.tile(pods, hs.size)
    wl.index_hashes = np.concatenate([wl.index_hashes, extra_h])
    wl.index_pods = np.concatenate([wl.index_pods, extra_p])
    wl.index_slots = 1 << 14
    assert_same(*run_both(pkg, orc, wl, max_pods=4096))


def test_small_pod_counts_table_holds_every_pod(pkg, orc):
    for P in (1, 2, 17, 64):
        wl = pkg.workload.make_workload(3, R=128, P=P, pods_per_group=64)   # every pod is in M for most requests
        assert_same(*run_both(pkg, orc, wl, max_pods=1024))


def test_index_remove_pod_and_reserved_hashes(pkg, orc):
    """Tombstones (rows emptied by remove_pod behave as absent) and the reserved hashes 0 / ~0."""
    Q, KV, L, PF = 1, 2, 3, 4
    P, B = 130, 6
    pods = pkg.workload.make_pods(99, P, 128)
    rng = np.random.default_rng(3)
    # chains built from reserved and ordinary hashes
    H0, HF = np.uint64(0), np.uint64(0xFFFFFFFFFFFFFFFF)
    chains = np.array([[H0, 5, 6, 7, 8, 9], [HF, H0, 11, 12, 13, 14], [21, 22, HF, 24, 25, 26], [31, H0, HF, 34, 35, 36]], dtype=np.uint64)
    hashes = chains[rng.integers(0, 4, 200)]
    reqs = pkg.picker.make_req_rows(rng.integers(-1, 128, 200), np.full(200, B), hashes, B)
    ih = np.concatenate([np.repeat(chains[g], 3) for g in range(4)])
    ip = np.concatenate([np.tile(np.array([g, g + 64, g + 100], dtype=np.uint32), B) for g in range(4)])
    chain = [(KV, 1), (PF, 5)]
    with pkg.BatchedPicker(chain, max_pods=1024, max_blocks=B, max_batch=256, index_slots=256) as pk:
        pk.publish(pods)
        pk.index_insert(ih, ip)
        oix = orc.OracleIndex()
        oix.insert(ih, ip)
        for step in range(4):
            picks, scores = pk.pick(reqs)
            op, osc, _ = orc.pick_batch(chain, pods, oix, reqs, B)
            assert_same(picks, scores, op, osc)
            # remove the pods of group `step` one after another: rows become empty -> keys tombstoned
            for pod in (step, step + 64, step + 100):
                pk.index_remove_pod(pod)
                oix.remove_pod(pod)
        picks, scores = pk.pick(reqs)
        op, osc, _ = orc.pick_batch(chain, pods, oix, reqs, B)
        assert_same(picks, scores, op, osc)
        # re-insert after tombstoning: the chain must be found again
        pk.index_insert(ih[:18], ip[:18])
        oix.insert(ih[:18], ip[:18])
        picks, scores = pk.pick(reqs)
        op, osc, _ = orc.pick_batch(chain, pods, oix, reqs, B)
        assert_same(picks, scores, op, osc)


def test_insert_picks_post_pass_matches_oracle(pkg, orc):
    """SEMANTICS.md §6: after a batch, index[hash[r][i]] ∪= {pick[r]} — device post-pass vs oracle, then re-pick."""
    import torch
    wl = pkg.workload.make_workload(3, R=512, P=600)
    with pkg.BatchedPicker(wl.chain, max_pods=1024, max_blocks=wl.B, max_batch=wl.R, index_slots=1 << 16) as pk:
        pk.publish(wl.pods)
        pk.index_insert(wl.index_hashes, wl.index_pods)
        oix = orc.OracleIndex()
        oix.insert(wl.index_hashes, wl.index_pods)
        d_reqs = torch.from_numpy(wl.reqs.view(np.int64)).cuda()
        d_pick = torch.empty(wl.R, dtype=torch.int32, device="cuda")
        d_score = torch.empty(wl.R, dtype=torch.float64, device="cuda")
        for _ in range(3):
            pk.pick_device(d_reqs.data_ptr(), wl.R, None, d_pick.data_ptr(), d_score.data_ptr())
            pk.index_insert_picks_device(d_reqs.data_ptr(), d_pick.data_ptr(), wl.R)
            torch.cuda.synchronize()
            op, osc, _ = orc.pick_batch(wl.chain, wl.pods, oix, wl.reqs, wl.B)
            assert_same(d_pick.cpu().numpy(), d_score.cpu().numpy(), op, osc)
            oix.insert_picks(wl.reqs, wl.B, op)
        assert pk.index_size() == oix.size()


def test_device_prompt_hashing_matches_host(pkg):
    """eppk_hash_prompts_device writes the same request rows as the host chain (SEMANTICS.md §4), bit for bit."""
    import torch
    rng = np.random.default_rng(11)
    R, B, BC, stride = 300, 12, 64, 1024
    prompts = rng.integers(0, 256, (R, stride), dtype=np.uint8)
    lens = rng.integers(0, stride + 1, R).astype(np.uint32)
    lens[:4] = [0, 63, 64, stride]                      # no block, just short of one, exactly one, more than max_blocks*64
    adapters = rng.integers(-1, 128, R).astype(np.int32)
    lib = pkg.load_library()
    seeds = np.array([lib.eppk_xxh64(b"adapter-%d" % a, len(b"adapter-%d" % a), 0) if a >= 0 else lib.eppk_xxh64(b"base", 4, 0) for a in adapters], dtype=np.uint64)
    want = np.zeros((R, 1 + B), dtype=np.uint64)
    for r in range(R):
        model = (b"adapter-%d" % adapters[r]) if adapters[r] >= 0 else b"base"
        h = pkg.picker.hash_prompt(model, prompts[r, :lens[r]].tobytes(), BC, B)
        want[r, 1:1 + h.size] = h
        want[r, 0] = np.uint64(np.uint32(adapters[r])) | (np.uint64(h.size) << np.uint64(32))
    with pkg.BatchedPicker([(2, 1)], max_pods=64, max_blocks=B, max_batch=R) as pk:
        d_p = torch.from_numpy(prompts).cuda()
        d_l = torch.from_numpy(lens.view(np.int32)).cuda()
        d_s = torch.from_numpy(seeds.view(np.int64)).cuda()
        d_a = torch.from_numpy(adapters).cuda()
        d_rows = torch.full((R, 1 + B), -1, dtype=torch.int64, device="cuda")
        pk.hash_prompts_device(d_p.data_ptr(), stride, d_l.data_ptr(), d_s.data_ptr(), d_a.data_ptr(), R, BC, d_rows.data_ptr())
        torch.cuda.synchronize()
        got = d_rows.cpu().numpy().view(np.uint64)
        assert np.array_equal(got, want)
        # misaligned block size is rejected, not silently mis-hashed
        with pytest.raises(pkg.EppkError):
            pk.hash_prompts_device(d_p.data_ptr(), stride, d_l.data_ptr(), d_s.data_ptr(), d_a.data_ptr(), R, 60, d_rows.data_ptr())


def test_masked_fast_path_corner_cases(pkg, orc):
    """Masked batches on the sparse kernel: (a) masks that keep the global queue extremes (tables apply),
    (b) masks that remove every min-queue / max-queue pod (exact per-candidate evaluation, many candidates),
    (c) masks that remove the whole top of the ranking (table exhausted -> scan of the remaining candidates)."""
    wl = pkg.workload.make_workload(5, R=384, P=3000, masked=True)
    P, W = wl.P, (wl.P + 63) // 64
    q = wl.pods["queue"]
    bits = np.zeros((wl.R, W * 64), dtype=bool)
    rng = np.random.default_rng(5)
    bits[:, :P] = rng.random((wl.R, P)) < 0.6
    # (b) rows 0..127: drop all pods at the global minimum queue; rows 128..255: drop the global maximum
    bits[:128, :P] &= (q != q.min())[None, :]
    bits[128:256, :P] &= (q != q.max())[None, :]
    # (c) rows 256..: drop the 300 best pods by (queue asc, kv asc) so every adapter's top-64 table is masked out
    order = np.lexsort((wl.pods["kv_util"], q))
    bits[256:, order[:1500]] = False
    bits[300:310, :] = False
    bits[300:310, 7] = True                       # single candidate
    mask = np.packbits(bits.reshape(wl.R, W, 64), axis=2, bitorder="little").view(np.uint64).reshape(wl.R, W)
    assert_same(*run_both(pkg, orc, wl, mask=mask, max_pods=4096))
    # chains without a QUEUE scorer never need the exact path
    assert_same(*run_both(pkg, orc, wl, chain=[(2, 3), (3, 2), (4, 5)], mask=mask, max_pods=4096))
    assert_same(*run_both(pkg, orc, wl, chain=[(4, 5), (3, 2)], mask=mask, max_pods=4096))


@pytest.mark.parametrize("P,max_pods", [(3000, 4096), (1500, 2048), (700, 1024), (50, 64)])
@pytest.mark.parametrize("chain", [None, [(KV, 1), (Q, 3), (PF, 2), (Q, 1)], [(PF, 4), (L, 2), (Q, 2)]], ids=["c5", "queue_twice", "prefix_first"])
def test_masked_requests_with_their_own_queue_normalisers_from_the_lists(pkg, orc, P, max_pods, chain):
    """Single picks whose candidates miss a snapshot-wide QUEUE extreme are evaluated candidate by candidate with the request's own
    normalisers (request.go:104-133 + the queue scorer's min / max over the CANDIDATES).  The fast kernel's list route does that from
    the hits' pod lists (matched[] in a byte histogram, masked_exact_hist) in every lane-word width: sparse and dense masks, hits
    whose lists DIFFER (a second, third ... pod learned for some blocks only), listed pods inside and outside the candidates."""
    wl = pkg.workload.make_workload(5, R=256, P=P, masked=True)
    W = (P + 63) // 64
    q = wl.pods["queue"]
    rng = np.random.default_rng(P)
    bits = np.zeros((wl.R, W * 64), dtype=bool)
    dens = np.where(np.arange(wl.R) % 3 == 0, 0.6, np.where(np.arange(wl.R) % 3 == 1, 0.125, 0.03))
    bits[:, :P] = rng.random((wl.R, P)) < dens[:, None]
    bits[: wl.R // 2, :P] &= (q != q.min())[None, :]
    bits[wl.R // 2 :, :P] &= (q != q.max())[None, :]
    bits[::16, :P] &= ((q != q.max()) & (q != q.min()))[None, :]
    # more pods for SOME blocks of the indexed prefixes: the lists of a request's hits differ, matched[] varies per pod; every second
    # request is sure to have one of its listed pods among its candidates
    ih, ip = wl.index_hashes, wl.index_pods
    sel = rng.random(ih.size) < 0.3
    ih2 = np.concatenate([ih, ih[sel], ih[sel][::2]])
    ip2 = np.concatenate([ip, (ip[sel].astype(np.int64) * 7 + 3) % P, (ip[sel][::2].astype(np.int64) * 11 + 5) % P]).astype(ip.dtype)
    first = {int(h): int(p_) for h, p_ in zip(ih[::-1], ip[::-1])}
    for r in range(0, wl.R, 2):
        h0 = int(wl.reqs[r, 1])
        if h0 in first:
            bits[r, first[h0]] = True
    mask = np.packbits(bits.reshape(wl.R, W, 64), axis=2, bitorder="little").view(np.uint64).reshape(wl.R, W)
    wl2 = dataclasses.replace(wl, index_hashes=ih2, index_pods=ip2)
    assert_same(*run_both(pkg, orc, wl2, chain=chain, mask=mask, max_pods=max_pods))


def _home_bucket(h: np.ndarray, n_buckets: int) -> np.ndarray:
    """Home bucket of a block hash, as libeppk places it (eppk_kernels.hip.h home_bucket): top log2(buckets) bits of
    (lo ^ hi) * 0x9E3779B1 mod 2^32.  Test-side restatement used only to BUILD colliding keys."""
    lg = int(n_buckets).bit_length() - 1
    f = ((h & np.uint64(0xFFFFFFFF)) ^ (h >> np.uint64(32))).astype(np.uint64)
    return ((f * np.uint64(0x9E3779B1)) & np.uint64(0xFFFFFFFF)) >> np.uint64(32 - lg)


@pytest.mark.parametrize("chain", [[(KV, 1), (PF, 5)], [(PF, 5), (Q, 1), (KV, 1)]], ids=["fast", "generic"])
def test_overflowed_buckets(pkg, orc, chain):
    """More keys than a 7-key bucket holds hash to the same home bucket: the surplus lives in the following buckets and
    look-ups (hits and misses) must walk there.  64 slots = 8 buckets, 30 keys, 24 of them with home bucket 1."""
    P, B, slots = 200, 8, 64
    rng = np.random.default_rng(11)
    cand = rng.integers(1, 2**63, 200000, dtype=np.uint64)
    hb = _home_bucket(cand, slots // 8)
    hot = cand[hb == 1][:24]
    cold = cand[hb == 3][:6]
    absent = cand[hb == 1][24:40]           # never inserted; their home bucket is full and overflowed
    assert hot.size == 24 and cold.size == 6 and absent.size == 16
    keys = np.concatenate([hot, cold])
    ih = np.repeat(keys, 3)
    ip = (np.arange(ih.size, dtype=np.uint32) * 7) % P
    pods = pkg.workload.make_pods(5, P, 128)
    # chains: runs of present keys, broken at various depths by an absent key
    R = 300
    hashes = np.zeros((R, B), dtype=np.uint64)
    for r in range(R):
        depth = rng.integers(0, B + 1)
        hashes[r, :depth] = rng.choice(keys, depth, replace=False) if depth else []
        hashes[r, depth:] = rng.choice(absent, B - depth)
    reqs = pkg.picker.make_req_rows(rng.integers(-1, 128, R), np.full(R, B), hashes, B)
    with pkg.BatchedPicker(chain, max_pods=1024, max_blocks=B, max_batch=R, index_slots=slots) as pk:
        pk.publish(pods)
        pk.index_insert(ih, ip)
        assert pk.index_size() == keys.size
        picks, scores = pk.pick(reqs)
    oix = orc.OracleIndex()
    oix.insert(ih, ip)
    op, osc, _ = orc.pick_batch(chain, pods, oix, reqs, B)
    assert_same(picks, scores, op, osc)


def test_index_full_is_reported(pkg):
    """The table refuses keys beyond its load limit (slots / 2) with EPPK_ERR_INDEX_FULL instead of degrading."""
    with pkg.BatchedPicker([(PF, 1)], max_pods=64, max_blocks=4, max_batch=8, index_slots=64) as pk:
        pk.publish(pkg.workload.make_pods(1, 64, 128))
        h = np.arange(1, 33, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
        pk.index_insert(h, np.zeros(32, dtype=np.uint32))          # exactly at the limit
        with pytest.raises(pkg.EppkError) as ei:
            pk.index_insert(np.array([12345], dtype=np.uint64), np.zeros(1, dtype=np.uint32))
        assert ei.value.code == -5


@pytest.mark.parametrize("B,R", [(40, 256), (100, 128), (256, 64)])
def test_long_chains_beyond_the_pipelined_gather(pkg, orc, B, R):
    """More than 32 blocks per request: the first 32 hashes go through the pipelined gather, the rest through the
    synchronous chunk loop (9 counter planes when B >= 64)."""
    wl = pkg.workload.make_workload(3, R=R, P=700, B=B)
    assert_same(*run_both(pkg, orc, wl, max_pods=1024))


@pytest.mark.parametrize("R", [1, 2, 3, 7, 4099])
def test_tiny_and_odd_batches(pkg, orc, R):
    """Batches smaller than the software pipeline's depth (prologue/epilogue with clamped row prefetches) and batches that
    do not divide evenly over the persistent wavefronts."""
    wl = pkg.workload.make_workload(5, R=R, P=4096)
    assert_same(*run_both(pkg, orc, wl))
    wm = pkg.workload.make_workload(3, R=R, P=1000, masked=True)
    assert_same(*run_both(pkg, orc, wm, mask=wm.mask, max_pods=1024))


@pytest.mark.parametrize("config,R,P,k,masked", [(5, 96, 4096, 4, False), (3, 128, 1000, 8, True), (2, 256, 256, 3, False),
                                                  (4, 128, 2048, 2, True), (1, 128, 16, 8, True)])
def test_topk_fallbacks(pkg, orc, config, R, P, k, masked):
    """Ordered fallback lists (PickResult.Fallbacks): column 0 is the pick, the rest the next best candidates under
    (total desc, index asc) -- bitwise against the oracle's totals; requests with fewer than k candidates are padded."""
    wl = pkg.workload.make_workload(config, R=R, P=P, masked=masked)
    mask = wl.mask
    if masked:
        mask = mask.copy()
        mask[0, :] = 0                                     # no candidate at all
        mask[1, :] = 0; mask[1, 0] = np.uint64(0b101)      # two candidates (< k)
    with pkg.BatchedPicker(wl.chain, max_pods=max(P, 1), max_blocks=wl.B, max_batch=R, index_slots=wl.index_slots) as pk:
        pk.publish(wl.pods)
        if wl.index_slots:
            pk.index_insert(wl.index_hashes, wl.index_pods)
        picks, scores = pk.pick_topk(wl.reqs, k, mask)
        p1, s1 = pk.pick(wl.reqs, mask)
    oix = orc.OracleIndex()
    oix.insert(wl.index_hashes, wl.index_pods)
    op, osc = orc.pick_topk(wl.chain, wl.pods, oix, wl.reqs, k, mask)
    assert np.array_equal(picks, op), f"fallback lists differ at rows {np.nonzero((picks != op).any(axis=1))[0][:5]}"
    assert np.array_equal(scores.view(np.uint64), osc.view(np.uint64))
    assert np.array_equal(picks[:, 0], p1) and np.array_equal(scores[:, 0].view(np.uint64), s1.view(np.uint64))
    if masked:
        assert (picks[0] == -1).all() and (picks[1, 2:] == -1).all() and set(picks[1, :2]) == {0, 2}


def test_pick_endpoints_fills_fallbacks(pkg):
    wl = pkg.workload.make_workload(2, R=16, P=64)
    eps = [pkg.Endpoint(address=f"10.0.{i // 256}.{i % 256}", port="8000") for i in range(64)]
    with pkg.BatchedPicker(wl.chain, max_pods=64, max_blocks=wl.B, max_batch=16) as pk:
        pk.publish(wl.pods)
        res = pk.pick_endpoints(eps, wl.reqs, fallbacks=2)
        plain = pk.pick_endpoints(eps, wl.reqs)
    assert all(len(r.fallbacks) == 2 and r.endpoint not in r.fallbacks for r in res)
    assert [r.endpoint for r in res] == [r.endpoint for r in plain]


def test_index_ageing_and_word_reuse(pkg, orc):
    """SEMANTICS.md §6a: inserts stamp hashes with the index epoch, evict_older drops the stale ones, their table words are
    reused by later inserts -- picks, scores and live-key counts stay equal to the oracle through several generations, in a
    table small enough (512 slots, at most 256 non-empty words) that the later generations only fit because evicted words are
    recycled (4 x 96 new hashes + re-inserted ones)."""
    P, B, slots = 300, 8, 512
    pods = pkg.workload.make_pods(7, P, 128)
    rng = np.random.default_rng(21)
    chain = [(Q, 1), (KV, 2), (L, 1), (PF, 4)]

    def generation(seed, n_chains=12):
        g = np.random.default_rng(seed)
        chains = g.integers(1, 2**63, (n_chains, B), dtype=np.uint64)
        ih = np.repeat(chains.ravel(), 2)
        ip = g.integers(0, P, ih.size).astype(np.uint32)
        return chains, ih, ip

    def requests(chains_list, R=256):
        allc = np.concatenate(chains_list)
        pick = rng.integers(0, allc.shape[0], R)
        cut = rng.integers(0, B + 1, R)
        hs = allc[pick].copy()
        for r in range(R):                        # break the chain at a random depth with an unknown hash
            hs[r, cut[r]:] = rng.integers(1, 2**63, B - cut[r], dtype=np.uint64)
        return pkg.picker.make_req_rows(rng.integers(-1, 128, R), np.full(R, B), hs, B)

    with pkg.BatchedPicker(chain, max_pods=1024, max_blocks=B, max_batch=256, index_slots=slots) as pk:
        pk.publish(pods)
        oix = orc.OracleIndex()
        gens = []
        for gen in range(4):
            chains, ih, ip = generation(100 + gen)
            gens.append(chains)
            pk.index_insert(ih, ip); oix.insert(ih, ip)                       # 96 new hashes, stamped with the current epoch
            if gen >= 1:                                                      # keep a few old hashes alive by re-inserting them
                keep_h = gens[gen - 1][:3].ravel(); keep_p = np.full(keep_h.size, gen, dtype=np.uint32)
                pk.index_insert(keep_h, keep_p); oix.insert(keep_h, keep_p)
            reqs = requests(gens)
            picks, scores = pk.pick(reqs)
            op, osc, _ = orc.pick_batch(chain, pods, oix, reqs, B)
            assert_same(picks, scores, op, osc)
            assert pk.index_size() == oix.size()
            e_dev, e_orc = pk.index_advance_epoch(), oix.advance_epoch()
            assert e_dev == e_orc == gen + 2
            n_dev, n_orc = pk.index_evict_older(e_dev - 1), oix.evict_older(e_orc - 1)   # drop everything not touched this epoch
            assert n_dev == n_orc and pk.index_size() == oix.size()
            picks, scores = pk.pick(reqs)
            op, osc, _ = orc.pick_batch(chain, pods, oix, reqs, B)
            assert_same(picks, scores, op, osc)


@pytest.mark.parametrize("slots", [1 << 23, 1 << 24])
@pytest.mark.parametrize("lists", ["1", "0"])
def test_index_of_4gib_and_more(pkg, orc, monkeypatch, slots, lists):
    """An index whose rows + keys reach 4 GiB (2^23 slots x 512 B at P = 4096) and beyond (2^24 slots: 8.6 GB, half of the rows past
    the 4 GiB offset) is served by the BIG instantiation of the fast kernel (rows through wave-uniform 64-bit bases) -- same picks,
    same scores, with the pod lists and with the dense rows alone (EPPK_LISTS=0: every hit reads its row)."""
    monkeypatch.setenv("EPPK_LISTS", lists)
    wl = pkg.workload.make_workload(5, R=2048, P=4096)
    wl.index_slots = slots
    assert_same(*run_both(pkg, orc, wl))
    wm = pkg.workload.make_workload(5, R=256, P=4096, masked=True)
    wm.index_slots = slots
    assert_same(*run_both(pkg, orc, wm, mask=wm.mask))


def test_full_size_headline_batch_and_batch_properties(pkg, orc):
    """BASELINE.json's headline size (64k requests x 4096 pods, full chain + prefix index) against the oracle on all host cores,
    plus two size-independent properties of the batched pick: determinism (same batch twice -> identical bits) and
    equivariance under a permutation of the requests (a pick depends on its own request row only)."""
    import os
    wl = pkg.workload.make_workload(5)
    assert wl.R == 65536 and wl.P == 4096
    with pkg.BatchedPicker(wl.chain, max_pods=wl.P, max_blocks=wl.B, max_batch=wl.R, index_slots=wl.index_slots) as pk:
        pk.publish(wl.pods)
        pk.index_insert(wl.index_hashes, wl.index_pods)
        picks, scores = pk.pick(wl.reqs)
        picks2, scores2 = pk.pick(wl.reqs)
        perm = np.random.default_rng(5).permutation(wl.R)
        pp, sp = pk.pick(np.ascontiguousarray(wl.reqs[perm]))
    assert np.array_equal(picks, picks2) and np.array_equal(scores.view(np.uint64), scores2.view(np.uint64))
    assert np.array_equal(pp, picks[perm]) and np.array_equal(sp.view(np.uint64), scores[perm].view(np.uint64))
    oix = orc.OracleIndex()
    oix.insert(wl.index_hashes, wl.index_pods)
    op, osc, _ = orc.pick_batch(wl.chain, wl.pods, oix, wl.reqs, wl.B, threads=os.cpu_count() or 1)
    assert_same(picks, scores, op, osc)


@pytest.mark.parametrize("chain,kind", [
    ([(Q, 2), (KV, 2), (L, 1), (PF, 3)], 1),
    ([(PF, 3), (L, 1)], 1),
    ([(PF, 3), (KV, 5)], 2),                         # reference example (0845-…/examples/example.yaml:21-25)
    ([(L, 1), (Q, 2), (PF, 3), (KV, 2)], 2),
    ([(L, 1), (Q, 2), (KV, 2), (Q, 1), (PF, 3)], 0),
    ([(PF, 3), (Q, 1), (PF, 3)], 0),
])
def test_which_kernel_serves_a_chain(pkg, chain, kind):
    with pkg.BatchedPicker(chain, max_pods=64, max_blocks=4, max_batch=8, index_slots=64) as pk:
        assert pk.chain_is_fused() == kind


def test_interpreted_tail_at_headline_shape(pkg, orc):
    """The reference example's `[prefix-cache: 3, kv-cache-util: 5]` and a chain with scorers on both sides of LORA / PREFIX at
    the headline pod count (P = 4096, u64 lane words), masked and unmasked."""
    for chain in ([(PF, 3), (KV, 5)], [(Q, 1), (L, 2), (KV, 2), (PF, 3), (Q, 2)]):
        wl = pkg.workload.make_workload(5, R=768, P=4096)
        assert_same(*run_both(pkg, orc, wl, chain=chain))
        wm = pkg.workload.make_workload(5, R=384, P=4096, masked=True)
        assert_same(*run_both(pkg, orc, wm, chain=chain, mask=wm.mask))


def test_library_before_torch_shares_one_hip_runtime():
    """A process that loads libeppk first and PyTorch second must still see the GPU from both (one HIP runtime per process:
    _lib.load_library imports torch ahead of the dlopen, see the comment there)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import numpy as np\n"
        "import __graft_entry__ as g\n"
        "pkg = g.load_package()\n"
        "wl = pkg.workload.make_workload(3, R=64, P=100)\n"
        "pk = pkg.BatchedPicker(wl.chain, max_pods=128, max_blocks=wl.B, max_batch=wl.R, index_slots=wl.index_slots)\n"
        "pk.publish(wl.pods); pk.index_insert(wl.index_hashes, wl.index_pods)\n"
        "p0, s0 = pk.pick(wl.reqs)\n"
        "import torch\n"
        "assert torch.cuda.is_available()\n"
        "d = torch.from_numpy(wl.reqs.view(np.int64)).cuda()\n"
        "dp = torch.empty(wl.R, dtype=torch.int32, device='cuda'); ds = torch.empty(wl.R, dtype=torch.float64, device='cuda')\n"
        "pk.pick_device(d.data_ptr(), wl.R, None, dp.data_ptr(), ds.data_ptr()); torch.cuda.synchronize()\n"
        "assert np.array_equal(dp.cpu().numpy(), p0)\n"
        "pk.close(); print('ok')\n" % root)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]


@pytest.mark.parametrize("P", [700, 2000, 4096])
def test_short_lists_follow_the_dense_rows(pkg, orc, P):
    """Every slot keeps its pod set twice: dense row + short list of pod ids (<= 24 members, csrc/eppk_kernels.hip.h).  Grow pod
    sets across the list capacity, shrink them again with pod removals, evict, re-insert: the self-check finds list and row
    in step after every operation and picks stay identical to the oracle's."""
    rng = np.random.default_rng(77 + P)
    B = 8
    chain = [(Q, 2), (KV, 2), (L, 1), (PF, 3)]
    pods = pkg.workload.make_pods(4242, P, 128)
    chains = rng.integers(1, 2**63, (12, B), dtype=np.uint64)
    R = 128

    def batch():
        hs = chains[rng.integers(0, chains.shape[0], R)].copy()
        for r in range(R):
            if rng.random() < 0.4:
                cut = int(rng.integers(0, B))
                hs[r, cut:] = rng.integers(1, 2**63, B - cut, dtype=np.uint64)
        return pkg.picker.make_req_rows(rng.integers(-1, 128, R), np.full(R, B), hs, B)

    with pkg.BatchedPicker(chain, max_pods=P, max_blocks=B, max_batch=R, index_slots=4096) as pk:
        pk.publish(pods)
        oix = orc.OracleIndex()

        def check(what):
            assert pk.index_selfcheck() == 0, what
            assert pk.index_size() == oix.size(), what
            reqs = batch()
            picks, scores = pk.pick(reqs)
            op, osc, _ = orc.pick_batch(chain, pods, oix, reqs, B)
            assert_same(picks, scores, op, osc)

        members = {c: [] for c in range(chains.shape[0])}
        # grow: chain c gets (c + 1) * 4 pods in three rounds -> sets of 4 .. 48 members, the larger ones cross the capacity
        for rnd in range(3):
            ih, ip = [], []
            for c in range(chains.shape[0]):
                want = (c + 1) * 4 * (rnd + 1) // 3
                new = rng.choice(P, size=min(want, P), replace=False)[: max(want - len(members[c]), 0)]
                members[c] += [int(x) for x in new]
                for p_ in new:
                    depth = int(rng.integers(1, B + 1))
                    ih.append(chains[c, :depth]); ip.append(np.full(depth, p_, dtype=np.uint32))
            ih = np.concatenate(ih); ip = np.concatenate(ip)
            pk.index_insert(ih, ip); oix.insert(ih, ip)
            check(f"grow round {rnd}")
        # shrink: remove pods that many sets contain, until the big sets fit a list again
        for step in range(30):
            c = int(rng.integers(6, chains.shape[0]))
            if not members[c]:
                continue
            pod = members[c].pop()
            pk.index_remove_pod(pod); oix.remove_pod(pod)
            for m in members.values():
                if pod in m:
                    m.remove(pod)
            if step % 5 == 4:
                check(f"shrink step {step}")
        check("after shrinking")
        # age everything out, then start over
        e = pk.index_advance_epoch(); oix.advance_epoch()
        assert pk.index_evict_older(e) == oix.evict_older(e)
        check("after evicting everything")
        ih = chains[:, :4].reshape(-1); ip = rng.integers(0, P, ih.size).astype(np.uint32)
        pk.index_insert(ih, ip); oix.insert(ih, ip)
        check("re-inserted")
        pk.index_clear(); oix = orc.OracleIndex()
        check("cleared")


@pytest.mark.parametrize("profiling", [False, True])
def test_stream_wait_pick_orders_a_second_stream(pkg, orc, profiling):
    """eppk_stream_wait_pick: work enqueued on another stream after it sees the picks of the launch it waited for
    (with the kernel's own completion event while profiling, with an event recorded behind the launch otherwise)."""
    import torch
    wl = pkg.workload.make_workload(3, R=4096, P=1000)
    with pkg.BatchedPicker(wl.chain, max_pods=1024, max_blocks=wl.B, max_batch=wl.R, index_slots=wl.index_slots) as pk:
        pk.publish(wl.pods)
        pk.index_insert(wl.index_hashes, wl.index_pods)
        oix = orc.OracleIndex(); oix.insert(wl.index_hashes, wl.index_pods)
        op, _, _ = orc.pick_batch(wl.chain, wl.pods, oix, wl.reqs, wl.B)
        d_reqs = torch.from_numpy(wl.reqs.view(np.int64)).cuda()
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        pk.profile(profiling)
        for _ in range(5):
            d_pick = torch.full((wl.R,), -7, dtype=torch.int32, device="cuda")
            d_copy = torch.full((wl.R,), -9, dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()
            pk.pick_device(d_reqs.data_ptr(), wl.R, None, d_pick.data_ptr(), None, s1.cuda_stream)
            pk.stream_wait_pick(s2.cuda_stream)
            with torch.cuda.stream(s2):
                d_copy.copy_(d_pick, non_blocking=True)
            s2.synchronize()
            assert np.array_equal(d_copy.cpu().numpy(), op)
        if profiling:
            pk.profile_drain()
            pk.profile(False)
