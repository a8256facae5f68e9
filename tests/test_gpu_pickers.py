"""GPU: the second picker of the reference example (`selection: random-top-3`, 0845-…/examples/example.yaml:25; SEMANTICS.md §3b)
and assumed load (006-scheduler/README.md:154-156; SEMANTICS.md §2b), both against the oracle, bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

Q, KV, L, PF = 1, 2, 3, 4


def _same(picks, scores, op, osc, what=""):
    bad = np.nonzero(picks != op)[0]
    assert bad.size == 0, f"{what}: {bad.size} picks differ, first {bad[:5]}: gpu {picks[bad[:5]]} oracle {op[bad[:5]]}"
    assert np.array_equal(scores.view(np.uint64), osc.view(np.uint64)), what


@pytest.mark.parametrize("chain", [[(PF, 3), (KV, 5)],                          # the example's decode profile
                                   [(Q, 2), (KV, 2), (L, 1), (PF, 3)],
                                   [(L, 1), (Q, 2), (KV, 2), (Q, 1), (PF, 3)]])  # generic kernel
@pytest.mark.parametrize("k,masked", [(3, False), (3, True), (1, False), (8, True)])
def test_random_top_k_matches_the_oracle(pkg, orc, chain, k, masked):
    wl = pkg.workload.make_workload(3, R=900, P=777, masked=masked)
    if masked:
        wl.mask[5, :] = 0                                  # no candidate
        wl.mask[6, :] = 0
        wl.mask[6, 2] = np.uint64(0b101)                   # two candidates: fewer than k
    with pkg.BatchedPicker(chain, max_pods=1024, max_blocks=wl.B, max_batch=wl.R, index_slots=wl.index_slots) as pk:
        pk.publish(wl.pods)
        pk.index_insert(wl.index_hashes, wl.index_pods)
        oix = orc.OracleIndex()
        oix.insert(wl.index_hashes, wl.index_pods)
        for seed in (0, 1, 0xDEADBEEFCAFEF00D):
            picks, scores = pk.pick_random_topk(wl.reqs, k, seed, wl.mask)
            op, osc = orc.pick_random_topk(chain, wl.pods, oix, wl.reqs, wl.B, k, seed, wl.mask)
            _same(picks, scores, op, osc, f"k {k} seed {seed}")
        if masked:
            assert picks[5] == -1 and scores[5] == 0.0
        # the choice is among the request's fallback list, and with k > 1 it is not always its head
        tp, _ = pk.pick_topk(wl.reqs, k, wl.mask)
        assert all(picks[r] in tp[r] for r in range(wl.R))
        if k > 1:
            assert np.any(picks != tp[:, 0])
        if k == 1:
            assert np.array_equal(picks, tp[:, 0])


@pytest.mark.parametrize("chain", [[(Q, 2), (KV, 2), (L, 1), (PF, 3)], [(PF, 3), (Q, 5)], [(Q, 1)], [(L, 1), (Q, 2), (KV, 2), (Q, 1), (PF, 3)]])
@pytest.mark.parametrize("epochs", [1, 3, 16])
def test_assumed_load_epochs_match_the_oracle(pkg, orc, chain, epochs):
    wl = pkg.workload.make_workload(3, R=1000, P=300)
    batches = [wl.reqs, pkg.workload.make_requests(wl, 99)]
    with pkg.BatchedPicker(chain, max_pods=512, max_blocks=wl.B, max_batch=wl.R, index_slots=wl.index_slots) as pk:
        pk.publish(wl.pods)
        pk.index_insert(wl.index_hashes, wl.index_pods)
        oix = orc.OracleIndex()
        oix.insert(wl.index_hashes, wl.index_pods)
        # off: today's behaviour, twice the same picks
        p0, s0 = pk.pick(wl.reqs)
        _same(p0, s0, *orc.pick_batch(chain, wl.pods, oix, wl.reqs, wl.B)[:2], "off")
        pk.set_assumed_load(epochs)
        opods = wl.pods.copy()
        for b in (0, 1, 0):                                # the bumped gauges persist from batch to batch
            picks, scores = pk.pick(batches[b][: 1000 - 7 * b])
            op, osc = orc.pick_batch_assumed(chain, opods, oix, batches[b][: 1000 - 7 * b], wl.B, epochs)
            _same(picks, scores, op, osc, f"epochs {epochs} batch {b}")
        # a fresh publish forgets the assumed load
        pk.publish(wl.pods)
        opods = wl.pods.copy()
        picks, scores = pk.pick(wl.reqs)
        _same(picks, scores, *orc.pick_batch_assumed(chain, opods, oix, wl.reqs, wl.B, epochs), "after publish")
        pk.set_assumed_load(0)
        pk.publish(wl.pods)
        _same(*pk.pick(wl.reqs), p0, s0, "off again")


def test_assumed_load_spreads_a_herd(pkg, orc):
    """What the epochs are for: with a frozen snapshot every request of a group lands on the same pod; with assumed load the
    queue term pushes later epochs elsewhere.  (Queue-only chain on tie-heavy gauges: the effect is maximal.)"""
    wl = pkg.workload.make_workload(2, R=4096, P=256)
    chain = [(Q, 1)]
    with pkg.BatchedPicker(chain, max_pods=256, max_blocks=0, max_batch=wl.R) as pk:
        pk.publish(wl.pods)
        frozen, _ = pk.pick(wl.reqs)
        assert np.unique(frozen).size == 1                 # everybody picks the first pod with the shortest queue
        shares = {}
        for e in (1, 8, 64, 4096):
            pk.publish(wl.pods)
            pk.set_assumed_load(e)
            picks, scores = pk.pick(wl.reqs)
            opods = wl.pods.copy()
            op, osc = orc.pick_batch_assumed(chain, opods, None, wl.reqs, 0, e)
            _same(picks, scores, op, osc, f"epochs {e}")
            shares[e] = np.bincount(picks, minlength=wl.P).max() / wl.R
        assert shares[1] == 1.0 and shares[8] <= 0.126 and shares[64] < shares[8] and shares[4096] < shares[64]


def test_random_top_k_with_assumed_load_and_device_entry(pkg, orc):
    import torch
    chain = [(PF, 3), (KV, 5), (Q, 1)]
    wl = pkg.workload.make_workload(3, R=640, P=500)
    with pkg.BatchedPicker(chain, max_pods=512, max_blocks=wl.B, max_batch=wl.R, index_slots=wl.index_slots) as pk:
        pk.publish(wl.pods)
        pk.index_insert(wl.index_hashes, wl.index_pods)
        oix = orc.OracleIndex()
        oix.insert(wl.index_hashes, wl.index_pods)
        d_reqs = torch.from_numpy(wl.reqs.view(np.int64)).cuda()
        d_pick = torch.empty(wl.R, dtype=torch.int32, device="cuda")
        d_score = torch.empty(wl.R, dtype=torch.float64, device="cuda")
        st = torch.cuda.Stream()
        pk._check(pk._lib.eppk_pick_random_topk_device(pk._ctx, d_reqs.data_ptr(), wl.R, None, 3, 77, d_pick.data_ptr(), d_score.data_ptr(), st.cuda_stream), "dev")
        st.synchronize()
        _same(d_pick.cpu().numpy(), d_score.cpu().numpy(), *orc.pick_random_topk(chain, wl.pods, oix, wl.reqs, wl.B, 3, 77), "device entry")
        assert pk.launch_status() == 0


@pytest.mark.parametrize("P", [300, 1500, 4096])
def test_per_pod_capacity_trims_oldest_epochs_first(pkg, orc, P):
    """eppk_index_trim_pods (SEMANTICS.md §6c): a pod listed under more hashes than its capacity loses its oldest epochs; the
    entries of the current epoch stay; pair counts, index sizes and picks equal the oracle's after every step."""
    rng = np.random.default_rng(P)
    chain = [(KV, 1), (PF, 5)]
    B = 8
    pods = pkg.workload.make_pods(7, P, 128)
    with pkg.BatchedPicker(chain, max_pods=P, max_blocks=B, max_batch=256, index_slots=1 << 15) as pk:
        pk.publish(pods)
        oix = orc.OracleIndex()
        hot = rng.choice(P, 6, replace=False)                  # a few pods that every epoch caches something on
        universe = []
        for epoch in range(6):
            h = rng.integers(1, 2**63, 600, dtype=np.uint64)
            p = np.where(rng.random(600) < 0.7, rng.choice(hot, 600), rng.integers(0, P, 600)).astype(np.uint32)
            pk.index_insert(h, p); oix.insert(h, p)
            universe.append(h)
            if epoch % 2 == 1:                                  # some old hashes are touched again: their stamp moves up
                again = universe[0][:100]
                pa = rng.choice(hot, 100).astype(np.uint32)
                pk.index_insert(again, pa); oix.insert(again, pa)
            e = pk.index_advance_epoch()
            assert e == oix.advance_epoch()
        reqs = pkg.picker.make_req_rows(rng.integers(-1, 128, 256), np.full(256, B), np.concatenate(universe)[rng.integers(0, 3600, (256, B))], B)
        for cap in (2000, 700, 150, 10, 0):
            got, want = pk.index_trim_pods(cap), oix.trim_pods(P, cap)
            assert got == want, (cap, got, want)
            assert pk.index_size() == oix.size() and pk.index_selfcheck() == 0
            picks, scores = pk.pick(reqs)
            op, osc, _ = orc.pick_batch(chain, pods, oix, reqs, B)
            _same(picks, scores, op, osc, f"cap {cap}")
        assert pk.index_trim_pods(0) == 0                       # idempotent
