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
        # beyond the limit: the launched path as before; with a mask: the masked resident workgroup where pick_quad_kernel's route exists
        want = orc.pick_batch(wl.chain, wl.pods, oix, wl.reqs[:65], wl.B)[:2]
        _same(pk.pick(wl.reqs[:65]), want, "n=65")
        assert pk.resident_stats()[1] == b1
        W = (P + 63) // 64
        mask = np.full((8, W), np.uint64(0xAAAAAAAAAAAAAAAA))
        if P % 64:
            mask[:, -1] &= np.uint64((1 << (P % 64)) - 1)
        _same(pk.pick(wl.reqs[:8], mask), orc.pick_batch(wl.chain, wl.pods, oix, wl.reqs[:8], wl.B, mask)[:2], "masked")
        assert pk.resident_stats()[1] == b1 + (1 if limit == 64 else 0)
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


@pytest.mark.parametrize("P,B", [(4096, 32), (1000, 8)])
def test_what_a_dispatcher_issues_through_the_resident_workgroups(pkg, orc, resident, eppk_mode, P, B):
    """The variants beside plain picks (a resident workgroup each, started on first use): batches with candidate masks (the subset filter,
    handlers/request.go:104-133), ordered fallbacks (PickResult.Fallbacks, handlers/server.go:72-77) with and without masks, and the two
    staging sets of the pipelined host path (eppk_pick_stage_begin rings the doorbell, _end polls the completion word) -- picks, lists and
    scores against the oracle, across an index update and a publish, with requests the quad body defers (reserved hashes) among them."""
    quad = eppk_mode in ("default", "quadmin4")
    wl = pkg.workload.make_workload(5, R=256, P=P, n_groups=16, B=B, masked=True)
    wl.mask[3, :] = 0                                                                 # a request without candidates
    wl.mask[11, :] = 0; wl.mask[11, 0] = np.uint64(2)                                 # ... with exactly one
    oix = orc.OracleIndex()
    oix.insert(wl.index_hashes, wl.index_pods)
    with pkg.BatchedPicker(wl.chain, max_pods=P, max_blocks=wl.B, max_batch=256, index_slots=wl.index_slots * 4) as pk:
        pk.publish(wl.pods)
        pk.index_insert(wl.index_hashes, wl.index_pods)
        pods = wl.pods
        sb = [pk.stage_buffers(0, with_mask=True), pk.stage_buffers(1, with_mask=True)]
        W = (P + 63) // 64

        def round_(tag, reqs_all):
            b0 = pk.resident_stats()[1]
            served = 0
            for n in (1, 5, 16, 33, 64):
                reqs, mask = reqs_all[:n], wl.mask[:n]
                _same(pk.pick(reqs, mask), orc.pick_batch(wl.chain, pods, oix, reqs, wl.B, mask)[:2], f"{tag}: masked n={n}")
                served += 1 if quad else 0
                for k in (2, 4):
                    for m in (None, mask):
                        tp, ts = pk.pick_topk(reqs, k, m)
                        op, osc = orc.pick_topk_batch(wl.chain, pods, oix, reqs, wl.B, k, m)
                        assert np.array_equal(tp, op), f"{tag}: top-{k} n={n} masked={m is not None}"
                        assert np.array_equal(ts.view(np.uint64), osc.view(np.uint64)), f"{tag}: top-{k} scores n={n} masked={m is not None}"
                        served += 1 if quad else 0
                # the two staging sets: plain in set 0, masked in set 1, both in flight
                sb[0][0][:n] = reqs
                sb[1][0][:n] = reqs_all[64:64 + n]
                sb[1][1][:n * W] = wl.mask[64:64 + n].reshape(-1)
                pk.stage_begin(0, n)
                pk.stage_begin(1, n, use_mask=True)
                _same(pk.stage_end(0), orc.pick_batch(wl.chain, pods, oix, reqs, wl.B)[:2], f"{tag}: staged n={n}")
                _same(pk.stage_end(1), orc.pick_batch(wl.chain, pods, oix, reqs_all[64:64 + n], wl.B, wl.mask[64:64 + n])[:2], f"{tag}: staged masked n={n}")
                served += 1 + (1 if quad else 0) if n <= (64 if quad else 32) else 0
            # five call shapes (fast, quad, masked, top-k, top-k masked) over FOUR stream slots: a shape whose unit is not resident while
            # all slots are in use takes the launched path until it has been asked for eight times (then the least recently rung unit
            # makes room) -- most calls are answered by resident workgroups, never more than asked
            got = pk.resident_stats()[1] - b0
            assert served // 2 <= got <= served, (tag, got, served)

        round_("fresh", wl.reqs)
        # the index changes (an update launched on the context's stream), then the snapshot: the next doorbells must see both
        extra_h = wl.reqs[:32, 1 + wl.B // 2:1 + wl.B].reshape(-1).copy()
        extra_p = np.full(extra_h.size, 5 % P, dtype=np.uint32)
        pk.index_insert(extra_h, extra_p); oix.insert(extra_h, extra_p)
        pods = wl.pods.copy()
        pods["queue"] = (pods["queue"].astype(np.int64) * 5 + 1) % 59
        pk.publish(pods)
        round_("after insert + publish", wl.reqs)
        # requests the quad body defers: reserved hashes among their blocks -> the work-list pass inside the same resident workgroup
        odd = wl.reqs.copy()
        odd[2, 1 + 1] = np.uint64(0)
        odd[9, 1 + 0] = np.uint64(0xFFFFFFFFFFFFFFFF)
        odd[70, 1 + wl.B - 1] = np.uint64(0)
        rh = np.array([0, 0xFFFFFFFFFFFFFFFF], dtype=np.uint64)
        rp = np.array([3 % P, 9 % P], dtype=np.uint32)
        pk.index_insert(rh, rp); oix.insert(rh, rp)
        round_("deferred requests", odd)
        round_("deferred requests again", odd)                                        # (the masked form stages its LDS layout again behind a work-list pass)
        # a row out of range is refused by name (begin: small batches are checked on the host), nothing is delivered, the set stays usable
        sb[0][0][:4] = wl.reqs[:4]
        sb[0][0][2, 0] = np.uint64(1000) << np.uint64(32)
        with pytest.raises(Exception):
            pk.stage_begin(0, 4)
        sb[0][0][:4] = wl.reqs[:4]
        pk.stage_begin(0, 4)
        _same(pk.stage_end(0), orc.pick_batch(wl.chain, pods, oix, wl.reqs[:4], wl.B)[:2], "after a refused begin")
        assert pk.index_selfcheck() == 0 and pk.launch_status() == 0


@pytest.mark.parametrize("P,B", [(4096, 32), (700, 8)])
def test_small_learn_batches_update_the_index_inside_the_resident_workgroup(pkg, orc, resident, eppk_mode, P, B):
    """eppk_pick_stage_begin(EPPK_PICK_LEARN) of a small batch: the resident workgroup answers the picks and applies the post-route update
    index[hash[r][i]] U= {pick[r]} itself (0602-prefix-cache-aware-routing-proposal/README.md:101-108), right behind the answer -- no launch.
    A closed loop of small batches (both staging sets, plain and masked, 1 .. 64 requests, returning conversations, requests the quad
    body defers), the oracle replaying the call order; between them what a shim does: epoch ticks, evictions on the device, a trim, a
    publish, a LARGE launched LEARN batch -- every one of them ordered behind the resident update and ahead of the next doorbell."""
    quad = eppk_mode in ("default", "quadmin4")
    wl = pkg.workload.make_workload(5, R=512, P=P, n_groups=24, B=B, masked=True)
    rng = np.random.default_rng(P + B)
    oix = orc.OracleIndex()
    oix.insert(wl.index_hashes, wl.index_pods)
    with pkg.BatchedPicker(wl.chain, max_pods=P, max_blocks=wl.B, max_batch=512, index_slots=1 << 16) as pk:
        pk.publish(wl.pods)
        pk.index_insert(wl.index_hashes, wl.index_pods)
        pods = wl.pods
        sb = [pk.stage_buffers(0, with_mask=True), pk.stage_buffers(1, with_mask=True)]
        W = (P + 63) // 64
        b0 = pk.resident_stats()[1]
        served = 0
        epoch = 1
        odd = wl.reqs.copy()
        odd[5, 1 + 1] = np.uint64(0)                                                 # reserved hashes: deferred by the quad body, learned all the same
        odd[40, 1 + 0] = np.uint64(0xFFFFFFFFFFFFFFFF)
        for step in range(60):
            n = int(rng.choice([1, 2, 7, 8, 16, 31, 48, 64]))
            lo = int(rng.integers(0, 512 - n))
            src = odd if step % 7 == 3 else wl.reqs
            reqs = src[lo:lo + n]
            masked = step % 3 == 2
            mask = wl.mask[lo:lo + n] if masked else None
            which = step & 1
            sb[which][0][:n] = reqs
            if masked:
                sb[which][1][:n * W] = mask.reshape(-1)
            pk.stage_begin(which, n, use_mask=masked, learn=True)
            op, osc, _ = orc.pick_batch(wl.chain, pods, oix, reqs, wl.B, mask)
            oix.insert_picks(reqs, wl.B, op)
            got = pk.stage_end(which)
            sb[which][0][:n] = np.uint64(0xDEAD)                                       # the caller refills the set right after end(): the update must not read it
            _same(got, (op, osc), f"step {step} n={n} masked={masked}")
            served += 1 if quad else 0
            if step % 10 == 4:                                                        # the shim's ageing between two batches
                epoch = pk.index_advance_epoch(); assert epoch == oix.advance_epoch()
                if epoch > 2:
                    pk.index_evict_older_device(epoch - 2)
                    oix.evict_older(epoch - 2)
            if step % 16 == 9:
                assert pk.index_size() == oix.size(), step
                assert pk.index_trim_pods(300) == oix.trim_pods(P, 300), step
            if step == 25:
                pods = wl.pods.copy()
                pods["queue"] = (pods["queue"].astype(np.int64) * 3 + 2) % 53
                pk.publish(pods)
            if step % 20 == 11:                                                       # a launched LEARN batch (beyond the resident limit) in between
                big = wl.reqs[:200]
                sb[which ^ 1][0][:200] = big
                pk.stage_begin(which ^ 1, 200, learn=True)
                op, osc, _ = orc.pick_batch(wl.chain, pods, oix, big, wl.B)
                oix.insert_picks(big, wl.B, op)
                _same(pk.stage_end(which ^ 1), (op, osc), f"step {step}: launched LEARN batch")
        assert pk.resident_stats()[1] - b0 == served
        assert pk.index_size() == oix.size() and pk.index_selfcheck() == 0 and pk.index_dropped() == 0 and pk.launch_status() == 0
        # every hash the oracle holds is listed with the same pods: one more (plain) batch over everything
        _same(pk.pick(wl.reqs[:64]), orc.pick_batch(wl.chain, pods, oix, wl.reqs[:64], wl.B)[:2], "after the loop")


@pytest.mark.parametrize("P,B", [(4096, 32), (1000, 8), (64, 4)])
def test_subset_filters_through_the_masked_resident_workgroup(pkg, orc, resident, eppk_mode, P, B):
    """What a per-request caller with a subset filter hands over (request.go:104-133, 141-163): a few requests, each with a handful of
    candidate endpoints -- nearly every one needs its own QUEUE normalisers.  The masked resident workgroup PARKS such rows and scores them
    itself, four side by side (pick_quad_body<RESIDENT> + quad_park_drain_i: 16 requests x 8 endpoints 79 -> 23 us host-observed), no
    work-list pass; dense masks beside them in the same batches; every size on both sides of the body switch (8 requests)."""
    wl = pkg.workload.make_workload(5, R=256, P=P, n_groups=16, B=B)
    oix = orc.OracleIndex()
    oix.insert(wl.index_hashes, wl.index_pods)
    rng = np.random.default_rng(P + B)
    W = (P + 63) // 64
    sets = {}
    for h, p_ in zip(wl.index_hashes.tolist(), wl.index_pods.tolist()):
        sets.setdefault(h, p_)
    with pkg.BatchedPicker(wl.chain, max_pods=P, max_blocks=wl.B, max_batch=256, index_slots=wl.index_slots * 4) as pk:
        pk.publish(wl.pods)
        pk.index_insert(wl.index_hashes, wl.index_pods)
        _, b0, _ = pk.resident_stats()
        served = 0
        for rep in range(3):
            for n in (1, 5, 8, 9, 16, 31, 32, 47, 64):
                lo = (rep * 61) % 128
                reqs = wl.reqs[lo:lo + n]
                mask = np.zeros((n, W), dtype=np.uint64)
                for r in range(n):
                    if (r + rep) % 5 == 4:                                   # a dense row among them
                        mask[r] = rng.integers(0, 2**63, W, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, W, dtype=np.uint64)
                        if P % 64:
                            mask[r, -1] &= np.uint64((1 << (P % 64)) - 1)
                        continue
                    keep = list(rng.choice(P, size=int(rng.integers(1, min(8, P) + 1)), replace=False))
                    if r % 2 == 0 and int(reqs[r, 1]) in sets:                # a pod the request's prefix is cached on stays a candidate
                        keep.append(sets[int(reqs[r, 1])])
                    for p_ in keep:
                        mask[r, int(p_) // 64] |= np.uint64(1) << np.uint64(int(p_) % 64)
                if n > 2:
                    mask[2] = 0                                              # no candidate at all
                want = orc.pick_batch(wl.chain, wl.pods, oix, reqs, wl.B, mask)[:2]
                _same(pk.pick(reqs, mask), want, f"subset masks n={n} rep={rep}")
                served += 1
                if n in (9, 16, 47):                                         # ordered fallbacks within the subset: the top-k masked unit
                    tw = orc.pick_topk(wl.chain, wl.pods, oix, reqs, 3, mask)
                    tg = pk.pick_topk(reqs, 3, mask)
                    assert np.array_equal(tg[0], tw[0]) and np.array_equal(tg[1].view(np.uint64), tw[1].view(np.uint64)), f"subset masks, top-3, n={n} rep={rep}"
                    served += 1
        if eppk_mode in ("default", "quadmin4"):
            assert pk.resident_stats()[1] - b0 == served, "every masked batch of at most 64 requests is answered by the masked resident workgroup"


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
