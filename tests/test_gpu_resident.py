"""The resident small-batch path (EPPK_RESIDENT=1; csrc/eppk_kernels.hip.h: pick_resident_kernel): eppk_pick_batch / eppk_pick_batch_staged
of at most 32 / 64 unmasked requests are answered by a workgroup that stays on the GPU and polls a doorbell in pinned host memory -- same picks
and scores as the launched kernels (bit for bit against the oracle), across publishes and index updates (the workgroup has no kernel
boundary to refresh its caches: it invalidates them behind every doorbell), idle time-outs and the library's own device-wide waits."""
import time

import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(120)]


def _same(got, want, what):
    assert np.array_equal(got[0], want[0]), f"{what}: picks differ at {np.nonzero(got[0] != want[0])[0][:5]}"
    assert np.array_equal(got[1].view(np.uint64), want[1].view(np.uint64)), f"{what}: scores differ"


@pytest.fixture
def resident(monkeypatch):
    monkeypatch.setenv("EPPK_RESIDENT", "1")


@pytest.mark.parametrize("P,B", [(4096, 32), (1000, 8), (2048, 16)])
def test_small_batches_through_the_resident_workgroup(pkg, orc, resident, eppk_mode, P, B):
    wl = pkg.workload.make_workload(5, R=256, P=P, n_groups=16, B=B)
    oix = orc.OracleIndex()
    oix.insert(wl.index_hashes, wl.index_pods)
    with pkg.BatchedPicker(wl.chain, max_pods=P, max_blocks=wl.B, max_batch=256, index_slots=wl.index_slots * 4) as pk:
        pk.publish(wl.pods)
        pk.index_insert(wl.index_hashes, wl.index_pods)
        on, b0, s0 = pk.resident_stats()
        assert on
        st, _ = pk.staging()
        # two forms of the resident kernel: a wavefront per request below 8 requests, four requests per wavefront from 8 on (where
        # pick_quad_kernel's route exists: not with EPPK_QUAD=0 / EPPK_LISTS=0) -- the default limit is 64 there, else 32
        limit = 64 if eppk_mode in ("default", "quadmin4") else 32
        served = 0
        for rep in range(3):
            for n in (1, 7, 8, 9, 16, 17, 32, 33, 61, 64):
                reqs = wl.reqs[(rep * 64) % 128:(rep * 64) % 128 + n]
                want = orc.pick_batch(wl.chain, wl.pods, oix, reqs, wl.B)[:2]
                _same(pk.pick(reqs), want, f"pick n={n}")
                st[:n] = reqs
                _same(pk.pick_staged(n), want, f"pick_staged n={n}")
                served += 2 if n <= limit else 0
        on, b1, s1 = pk.resident_stats()
        assert b1 - b0 == served and s1 >= 1, (b1 - b0, served, s1)
        # beyond the limit, with a mask, or as fallbacks: the launched path as before
        want = orc.pick_batch(wl.chain, wl.pods, oix, wl.reqs[:65], wl.B)[:2]
        _same(pk.pick(wl.reqs[:65]), want, "n=65")
        assert pk.resident_stats()[1] == b1
        W = (P + 63) // 64
        mask = np.full((8, W), np.uint64(0xAAAAAAAAAAAAAAAA))
        if P % 64:
            mask[:, -1] &= np.uint64((1 << (P % 64)) - 1)
        _same(pk.pick(wl.reqs[:8], mask), orc.pick_batch(wl.chain, wl.pods, oix, wl.reqs[:8], wl.B, mask)[:2], "masked")
        assert pk.resident_stats()[1] == b1
        # the index and the snapshot change under the resident workgroup: it must see both at the next doorbell
        extra_h = wl.reqs[:32, 1 + wl.B // 2:1 + wl.B].reshape(-1).copy()             # the unique tails of 32 requests, now cached on pod 5
        extra_p = np.full(extra_h.size, 5 % P, dtype=np.uint32)
        pk.index_insert(extra_h, extra_p); oix.insert(extra_h, extra_p)
        _same(pk.pick(wl.reqs[:32]), orc.pick_batch(wl.chain, wl.pods, oix, wl.reqs[:32], wl.B)[:2], "after an index insert")
        pods2 = wl.pods.copy()
        pods2["queue"] = (pods2["queue"].astype(np.int64) * 7 + 3) % 61
        pods2["kv_util"] = 1.0 - pods2["kv_util"]
        pk.publish(pods2)
        _same(pk.pick(wl.reqs[:30]), orc.pick_batch(wl.chain, pods2, oix, wl.reqs[:30], wl.B)[:2], "after a publish")
        # requests pick_quad_kernel's body cannot finish itself (reserved hashes 0 / ~0 among their blocks, lists that differ since the
        # insert above): behind the doorbell they take the work-list pass of pick_fast_kernel's body, inside the same resident workgroup
        odd = wl.reqs[:40].copy()
        odd[3, 1 + 2] = np.uint64(0)
        odd[7, 1 + 0] = np.uint64(0xFFFFFFFFFFFFFFFF)
        odd[21, 1 + wl.B - 1] = np.uint64(0)
        rh = np.array([0, 0xFFFFFFFFFFFFFFFF], dtype=np.uint64)
        rp = np.array([3 % P, 9 % P], dtype=np.uint32)
        pk.index_insert(rh, rp); oix.insert(rh, rp)
        _same(pk.pick(odd), orc.pick_batch(wl.chain, pods2, oix, odd, wl.B)[:2], "reserved hashes and differing lists, 40 requests")
        _same(pk.pick(odd[:9]), orc.pick_batch(wl.chain, pods2, oix, odd[:9], wl.B)[:2], "reserved hashes, 9 requests")
        _same(pk.pick(odd[:5]), orc.pick_batch(wl.chain, pods2, oix, odd[:5], wl.B)[:2], "reserved hashes, 5 requests (the other form)")
        e = pk.index_advance_epoch(); assert e == oix.advance_epoch()
        assert pk.index_evict_older(e) == oix.evict_older(e)                          # everything goes (a device-wide wait: the workgroup is parked)
        _same(pk.pick(wl.reqs[:20]), orc.pick_batch(wl.chain, pods2, oix, wl.reqs[:20], wl.B)[:2], "after an eviction")
        assert pk.index_size() == oix.size() and pk.index_selfcheck() == 0 and pk.launch_status() == 0
        # a row out of range is refused by name, nothing is delivered
        bad = wl.reqs[:4].copy()
        bad[2, 0] = np.uint64(1000) << np.uint64(32)
        with pytest.raises(Exception):
            pk.pick(bad)
        _same(pk.pick(wl.reqs[:4]), orc.pick_batch(wl.chain, pods2, oix, wl.reqs[:4], wl.B)[:2], "after a refused batch")


def test_the_resident_workgroup_leaves_when_idle_and_comes_back(pkg, orc, resident, monkeypatch):
    monkeypatch.setenv("EPPK_RESIDENT_IDLE_POLLS", "2000")                            # a few milliseconds of polls
    wl = pkg.workload.make_workload(5, R=32, P=700, n_groups=8)
    oix = orc.OracleIndex()
    oix.insert(wl.index_hashes, wl.index_pods)
    want = orc.pick_batch(wl.chain, wl.pods, oix, wl.reqs, wl.B)[:2]
    with pkg.BatchedPicker(wl.chain, max_pods=1024, max_blocks=wl.B, max_batch=64, index_slots=wl.index_slots) as pk:
        pk.publish(wl.pods)
        pk.index_insert(wl.index_hashes, wl.index_pods)
        for rep in range(4):
            _same(pk.pick(wl.reqs), want, f"round {rep}")
            time.sleep(0.15)                                                          # far beyond the idle limit: the workgroup has left
        on, batches, starts = pk.resident_stats()
        assert batches == 4 and starts >= 3, (batches, starts)
        # back to back: one start serves them all
        for rep in range(50):
            _same(pk.pick(wl.reqs[:16]), (want[0][:16], want[1][:16]), "back to back")
        assert pk.resident_stats()[2] <= starts + 1


def test_without_the_switch_nothing_is_resident(pkg, orc):
    wl = pkg.workload.make_workload(5, R=16, P=300, n_groups=4)
    with pkg.BatchedPicker(wl.chain, max_pods=300, max_blocks=wl.B, max_batch=16, index_slots=wl.index_slots) as pk:
        pk.publish(wl.pods)
        pk.pick(wl.reqs)
        assert pk.resident_stats() == (False, 0, 0)
