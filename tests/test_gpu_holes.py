"""GPU: holes of a snapshot (eppk_pod_row.flags & EPPK_POD_INACTIVE, SEMANTICS.md §6b) -- never candidates, outside the QUEUE
normalisers and the top tables, scrubbed out of the prefix index at publish, ignored by index inserts -- against the oracle, on
every kernel route (fused / interpreted tail / generic, masked or not, ordered fallbacks, dense rows only)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

Q, KV, L, PF = 1, 2, 3, 4
CHAINS = [[(Q, 2), (KV, 2), (L, 1), (PF, 3)], [(PF, 3), (KV, 5)], [(L, 1), (Q, 2), (KV, 2), (Q, 1), (PF, 3)], [(Q, 1)], [(L, 2), (PF, 1)]]


def _same(picks, scores, op, osc):
    bad = np.nonzero(picks != op)[0]
    assert bad.size == 0, f"{bad.size} picks differ, first {bad[:5]}: gpu {picks[bad[:5]]} oracle {op[bad[:5]]}"
    assert np.array_equal(scores.view(np.uint64), osc.view(np.uint64))


def _holes(rng, wl, frac):
    pods = wl.pods.copy()
    h = rng.random(wl.P) < frac
    # make the extremes of the queue gauge holes too: the QUEUE normalisers must come from the ACTIVE pods only
    q = pods["queue"]
    h[np.argmin(q)] = True
    h[np.argmax(q)] = True
    pods["flags"] = h.astype(np.uint32)
    return pods


@pytest.mark.parametrize("chain", CHAINS)
@pytest.mark.parametrize("masked", [False, True])
@pytest.mark.parametrize("P,frac", [(1000, 0.3), (4096, 0.05), (130, 0.9), (64, 0.5)])
def test_holes_on_every_route(pkg, orc, chain, masked, P, frac):
    rng = np.random.default_rng(P + 7 * len(chain) + masked)
    wl = pkg.workload.make_workload(5, R=384, P=P, masked=masked, pods_per_group=min(8, P))
    pods = _holes(rng, wl, frac)
    with pkg.BatchedPicker(chain, max_pods=max(P, 1), max_blocks=wl.B, max_batch=wl.R, index_slots=wl.index_slots) as pk:
        pk.publish(pods)
        pk.index_insert(wl.index_hashes, wl.index_pods)          # pairs naming a hole are ignored
        oix = orc.OracleIndex()
        oix.insert(wl.index_hashes, wl.index_pods, snapshot=pods)
        assert pk.index_size() == oix.size() and pk.index_selfcheck() == 0
        picks, scores = pk.pick(wl.reqs, wl.mask)
        op, osc, _ = orc.pick_batch(chain, pods, oix, wl.reqs, wl.B, wl.mask)
        _same(picks, scores, op, osc)
        assert not np.any(pods["flags"][picks[picks >= 0]] & 1)
        k = 5
        tp, ts = pk.pick_topk(wl.reqs, k, wl.mask)
        otp, ots = orc.pick_topk(chain, pods, oix, wl.reqs, k, wl.mask)
        assert np.array_equal(tp, otp) and np.array_equal(ts.view(np.uint64), ots.view(np.uint64))


def test_publish_scrubs_new_holes_and_a_reused_slot_starts_empty(pkg, orc):
    chain = CHAINS[0]
    wl = pkg.workload.make_workload(3, R=512, P=700)
    rng = np.random.default_rng(11)
    with pkg.BatchedPicker(chain, max_pods=1024, max_blocks=wl.B, max_batch=wl.R, index_slots=1 << 16) as pk:
        oix = orc.OracleIndex()
        pods = wl.pods.copy()

        def check():
            assert pk.index_size() == oix.size() and pk.index_selfcheck() == 0
            _same(*pk.pick(wl.reqs), *orc.pick_batch(chain, pods, oix, wl.reqs, wl.B)[:2])

        pk.publish(pods)                                          # A: everybody active
        pk.index_insert(wl.index_hashes, wl.index_pods)
        oix.insert(wl.index_hashes, wl.index_pods)
        check()
        cached = np.unique(wl.index_pods)
        gone = rng.choice(cached, size=cached.size // 2, replace=False)
        pods["flags"][gone] = 1                                   # B: half of the pods that hold cache entries leave
        pk.publish(pods)
        oix.scrub_inactive(pods)
        check()
        pk.index_insert(wl.index_hashes, wl.index_pods)          # learning about a hole is ignored; the others are re-stamped
        oix.insert(wl.index_hashes, wl.index_pods, snapshot=pods)
        check()
        pods["flags"][gone[: gone.size // 2]] = 0                 # C: some slots are handed to newcomers -- empty history
        pods["queue"][gone[: gone.size // 2]] = 0
        pk.publish(pods)
        oix.scrub_inactive(pods)
        check()
        import torch                                               # the post-route update works on re-activated slots
        d_reqs = torch.from_numpy(wl.reqs.view(np.int64)).cuda()
        d_pick = torch.empty(wl.R, dtype=torch.int32, device="cuda")
        for _ in range(2):
            pk.pick_device(d_reqs.data_ptr(), wl.R, None, d_pick.data_ptr(), None)
            pk.index_insert_picks_device(d_reqs.data_ptr(), d_pick.data_ptr(), wl.R)
            torch.cuda.synchronize()
            op, _, _ = orc.pick_batch(chain, pods, oix, wl.reqs, wl.B)
            assert np.array_equal(d_pick.cpu().numpy(), op)
            oix.insert_picks(wl.reqs, wl.B, op)
        check()
        assert pk.launch_status() == 0


def test_all_holes_and_dense_rows_only(pkg, orc, monkeypatch):
    wl = pkg.workload.make_workload(3, R=128, P=300)
    pods = wl.pods.copy()
    pods["flags"] = 1
    with pkg.BatchedPicker(wl.chain, max_pods=512, max_blocks=wl.B, max_batch=wl.R, index_slots=wl.index_slots) as pk:
        pk.publish(pods)
        pk.index_insert(wl.index_hashes, wl.index_pods)
        picks, scores = pk.pick(wl.reqs)
        assert np.all(picks == -1) and np.all(scores == 0.0) and pk.index_size() == 0
    monkeypatch.setenv("EPPK_LISTS", "0")                         # the dense-row route alone
    rng = np.random.default_rng(3)
    pods = _holes(rng, wl, 0.4)
    with pkg.BatchedPicker(wl.chain, max_pods=512, max_blocks=wl.B, max_batch=wl.R, index_slots=wl.index_slots) as pk:
        pk.publish(pods)
        pk.index_insert(wl.index_hashes, wl.index_pods)
        oix = orc.OracleIndex()
        oix.insert(wl.index_hashes, wl.index_pods, snapshot=pods)
        _same(*pk.pick(wl.reqs), *orc.pick_batch(wl.chain, pods, oix, wl.reqs, wl.B)[:2])
