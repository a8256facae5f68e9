"""GPU: device groups (include/eppk.h eppk_group_*) -- multi-GPU behind the C ABI, tested on ONE GPU with several member contexts
on device 0: request sharding (ragged, tiny and empty batches, candidate masks), the device-side all-gather of the picks (PEER
and HOST modes; RCCL when two distinct GPUs are visible) and the replicated post-route index update, all against the UNSHARDED
oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _oracle(orc, wl, reqs, mask=None, oix=None):
    if oix is None:
        oix = orc.OracleIndex()
        oix.insert(wl.index_hashes, wl.index_pods)
    p, s, _ = orc.pick_batch(wl.chain, wl.pods, oix, reqs, wl.B, mask)
    return p, s, oix


def _same(picks, scores, op, osc):
    assert np.array_equal(picks, op)
    assert np.array_equal(scores.view(np.uint64), osc.view(np.uint64))


@pytest.mark.parametrize("members", [1, 2, 3, 4])
@pytest.mark.parametrize("n", [0, 1, 2, 5, 777, 2048])
def test_sharded_pick_matches_the_unsharded_oracle(pkg, orc, members, n):
    wl = pkg.workload.make_workload(3, R=2048, P=1000)
    with pkg.DeviceGroup(wl.chain, [0] * members, max_pods=1024, max_blocks=wl.B, max_batch=2048, index_slots=wl.index_slots,
                         min_shard=1) as g:
        assert g.size == members and g.ranks_seen == members
        g.publish(wl.pods)
        g.index_insert(wl.index_hashes, wl.index_pods)
        picks, scores = g.pick(wl.reqs[:n])
        op, osc, _ = _oracle(orc, wl, wl.reqs[:n])
        _same(picks, scores, op, osc)


def test_masked_sharded_pick(pkg, orc):
    wl = pkg.workload.make_workload(5, R=1500, P=4096, masked=True)
    wl.mask[7, :] = 0
    with pkg.DeviceGroup(wl.chain, [0, 0, 0], max_pods=4096, max_blocks=wl.B, max_batch=1500, index_slots=wl.index_slots, min_shard=64) as g:
        g.publish(wl.pods)
        g.index_insert(wl.index_hashes, wl.index_pods)
        picks, scores = g.pick(wl.reqs, wl.mask)
        op, osc, _ = _oracle(orc, wl, wl.reqs, wl.mask)
        assert picks[7] == -1
        _same(picks, scores, op, osc)


def test_small_batches_stay_on_one_device(pkg, orc):
    """min_shard: a batch is spread over ceil(n / min_shard) members at most; results do not depend on the split."""
    wl = pkg.workload.make_workload(3, R=600, P=500)
    op, osc, _ = _oracle(orc, wl, wl.reqs)
    for ms in (1, 100, 256, 600, 4096):
        with pkg.DeviceGroup(wl.chain, [0, 0, 0, 0], max_pods=512, max_blocks=wl.B, max_batch=600, index_slots=wl.index_slots, min_shard=ms) as g:
            g.publish(wl.pods)
            g.index_insert(wl.index_hashes, wl.index_pods)
            _same(*g.pick(wl.reqs), op, osc)


@pytest.mark.parametrize("mode", ["peer", "host"])
@pytest.mark.parametrize("members", [2, 3])
def test_learn_applies_the_gathered_update_on_every_member(pkg, orc, mode, members):
    """EPPK_GROUP_LEARN: after a sharded batch every member's index replica gained index[hash[r][i]] U= {pick[r]} for ALL requests --
    the same state as the oracle's -- so that the next batches (different requests, sharded differently) still match, for several
    generations, with ageing in between."""
    gm = {"peer": pkg.picker.GATHER_PEER, "host": pkg.picker.GATHER_HOST}[mode]
    wl = pkg.workload.make_workload(3, R=1000, P=900)
    batches = [wl.reqs] + [pkg.workload.make_requests(wl, 1000 + i) for i in range(3)]
    with pkg.DeviceGroup(wl.chain, [0] * members, max_pods=1024, max_blocks=wl.B, max_batch=1000, index_slots=1 << 17, gather=gm,
                         min_shard=1) as g:
        g.publish(wl.pods)
        g.index_insert(wl.index_hashes, wl.index_pods)
        oix = orc.OracleIndex()
        oix.insert(wl.index_hashes, wl.index_pods)
        for gen in range(6):
            reqs = batches[gen % len(batches)][: 1000 - 37 * gen]        # ragged, different shard boundaries every generation
            picks, scores = g.pick(reqs, learn=True)
            op, osc, _ = _oracle(orc, wl, reqs, oix=oix)
            _same(picks, scores, op, osc)
            oix.insert_picks(reqs, wl.B, op)
            for i in range(members):
                assert g.member_index_size(i) == oix.size(), (gen, i)
            if gen % 2 == 1:
                e = g.index_advance_epoch()
                assert e == oix.advance_epoch()
                if e > 2:
                    assert g.index_evict_older(e - 1) == oix.evict_older(e - 1)
        # every replica, used on its own, is the oracle's index: the gathered picks reached every member complete and in order
        op, osc, _ = _oracle(orc, wl, batches[1], oix=oix)
        for i in range(members):
            assert g.member_selfcheck(i) == 0
            _same(*g.member_pick(i, batches[1]), op, osc)


def test_gather_without_learn_and_errors(pkg, orc):
    wl = pkg.workload.make_workload(3, R=300, P=300)
    with pkg.DeviceGroup(wl.chain, [0, 0], max_pods=512, max_blocks=wl.B, max_batch=300, index_slots=wl.index_slots, min_shard=1) as g:
        with pytest.raises(pkg.picker.EppkError) as ei:       # no snapshot yet
            g.pick(wl.reqs)
        assert ei.value.code == -4
        g.publish(wl.pods)
        g.index_insert(wl.index_hashes, wl.index_pods)
        op, osc, _ = _oracle(orc, wl, wl.reqs)
        _same(*g.pick(wl.reqs, gather=True), op, osc)
        before = g.member_index_size(0)
        assert g.member_index_size(1) == before               # GATHER alone does not touch the index
        bad = wl.reqs.copy()
        bad[17, 0] = np.uint64(200)                           # adapter 200
        with pytest.raises(pkg.picker.EppkError) as ei:
            g.pick(bad)
        assert ei.value.code == -1 and "row 17" in str(ei.value)
    with pytest.raises(pkg.picker.EppkError):                 # RCCL needs distinct devices
        pkg.DeviceGroup(wl.chain, [0, 0], max_pods=512, max_blocks=wl.B, max_batch=300, index_slots=wl.index_slots, gather=pkg.picker.GATHER_RCCL)
    with pytest.raises(pkg.picker.EppkError):                 # device ordinal out of range
        pkg.DeviceGroup(wl.chain, [0, 99], max_pods=512, max_blocks=wl.B, max_batch=300, index_slots=wl.index_slots)


def test_rccl_gather_on_two_gpus(pkg, orc):
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("EPPK_GATHER_RCCL needs two distinct GPUs in one process")
    wl = pkg.workload.make_workload(3, R=1024, P=900)
    with pkg.DeviceGroup(wl.chain, [0, 1], max_pods=1024, max_blocks=wl.B, max_batch=1024, index_slots=1 << 16,
                         gather=pkg.picker.GATHER_RCCL, min_shard=1) as g:
        assert g.ranks_seen == 2
        g.publish(wl.pods)
        g.index_insert(wl.index_hashes, wl.index_pods)
        oix = orc.OracleIndex()
        oix.insert(wl.index_hashes, wl.index_pods)
        for gen in range(3):
            picks, scores = g.pick(wl.reqs, learn=True)
            op, osc, _ = _oracle(orc, wl, wl.reqs, oix=oix)
            _same(picks, scores, op, osc)
            oix.insert_picks(wl.reqs, wl.B, op)
            assert g.member_index_size(0) == g.member_index_size(1) == oix.size()


@pytest.mark.parametrize("members,bucket", [(2, 1), (4, 4), (3, 2)])
def test_device_resident_shards_with_launch_groups(pkg, orc, members, bucket):
    """eppk_group_pick_device: every member scores ITS rows of `bucket` batches with one launch (rows already in its memory), the
    picks are all-gathered member-major by peer copies; member i's gathered array holds, for every member j, exactly the picks the
    oracle gives for j's rows -- nothing goes through the host between the upload of the rows and the final copy-out."""
    import torch
    wl = pkg.workload.make_workload(3, R=960, P=900)
    batches = [wl.reqs] + [pkg.workload.make_requests(wl, 70 + i) for i in range(bucket - 1)]
    oix = orc.OracleIndex()
    oix.insert(wl.index_hashes, wl.index_pods)
    want = [_oracle(orc, wl, b, oix=oix)[:2] for b in batches]
    per = (wl.R + members - 1) // members
    with pkg.DeviceGroup(wl.chain, [0] * members, max_pods=1024, max_blocks=wl.B, max_batch=bucket * per, index_slots=1 << 16) as g:
        g.publish(wl.pods)
        g.index_insert(wl.index_hashes, wl.index_pods)
        shards, n_rows = [], []
        for i in range(members):
            lo, hi = min(i * per, wl.R), min((i + 1) * per, wl.R)
            rows = np.concatenate([b[lo:hi] for b in batches])           # member i's rows of every batch of the bucket, back to back
            shards.append(torch.from_numpy(rows.view(np.int64)).cuda())
            n_rows.append(rows.shape[0])
        total = sum(n_rows)
        d_picks = [torch.full((n,), -9, dtype=torch.int32, device="cuda") for n in n_rows]
        d_scores = [torch.empty(n, dtype=torch.float64, device="cuda") for n in n_rows]
        d_all = [torch.full((total,), -9, dtype=torch.int32, device="cuda") for _ in range(members)]
        torch.cuda.synchronize()
        for _ in range(2):                                               # twice: the second call reuses every buffer
            g.pick_device([t.data_ptr() for t in shards], n_rows, [t.data_ptr() for t in d_picks], [t.data_ptr() for t in d_scores],
                          [t.data_ptr() for t in d_all])
            g.sync()
        off = 0
        for i in range(members):
            lo, hi = min(i * per, wl.R), min((i + 1) * per, wl.R)
            exp_p = np.concatenate([w[0][lo:hi] for w in want])
            exp_s = np.concatenate([w[1][lo:hi] for w in want])
            assert np.array_equal(d_picks[i].cpu().numpy(), exp_p)
            assert np.array_equal(d_scores[i].cpu().numpy().view(np.uint64), exp_s.view(np.uint64))
            for m in range(members):
                assert np.array_equal(d_all[m].cpu().numpy()[off:off + n_rows[i]], exp_p), (m, i)
            off += n_rows[i]
        with pytest.raises(pkg.picker.EppkError):                        # LEARN needs the whole batch on every member
            g._check(g._lib.eppk_group_pick_device(g._g, None, None, None, None, None, 1), "x")


@pytest.mark.parametrize("members", [2, 4])
def test_group_fallback_lists_match_the_unsharded_oracle(pkg, orc, members):
    """eppk_group_pick_topk / _random_topk: ordered fallbacks (handlers/server.go:72-77; 004-endpoint-picker-protocol/README.md:73) and the
    seeded picker over a sharded batch -- top-4, masked top-4, random-top-3 -- equal the oracle on the UNSHARDED batch (the random rule
    hashes a request's index in the batch: the shard boundaries must not show)."""
    wl = pkg.workload.make_workload(5, R=1100, P=4096, masked=True)
    wl.mask[5, :] = 0
    with pkg.DeviceGroup(wl.chain, [0] * members, max_pods=4096, max_blocks=wl.B, max_batch=1100, index_slots=wl.index_slots, min_shard=64) as g:
        g.publish(wl.pods)
        g.index_insert(wl.index_hashes, wl.index_pods)
        oix = orc.OracleIndex()
        oix.insert(wl.index_hashes, wl.index_pods)
        for n in (1100, 777, 3):
            for mask in (None, wl.mask[:n]):
                tp, ts = g.pick_topk(wl.reqs[:n], 4, mask)
                op, osc = orc.pick_topk_batch(wl.chain, wl.pods, oix, wl.reqs[:n], wl.B, 4, mask)
                assert np.array_equal(tp, op), (n, mask is None)
                assert np.array_equal(ts.view(np.uint64), osc.view(np.uint64)), (n, mask is None)
                rp, rs = g.pick_random_topk(wl.reqs[:n], 3, 4242, mask)
                op, osc = orc.pick_random_topk(wl.chain, wl.pods, oix, wl.reqs[:n], wl.B, 3, 4242, mask)
                _same(rp, rs, op, osc)
        # k = 1 is the pick
        tp, ts = g.pick_topk(wl.reqs, 1)
        op, osc, _ = _oracle(orc, wl, wl.reqs, oix=oix)
        _same(tp[:, 0], ts[:, 0], op, osc)
        with pytest.raises(pkg.picker.EppkError):
            g.pick_topk(wl.reqs, 9)
        bad = wl.reqs.copy()
        bad[901, 0] = np.uint64(200)
        with pytest.raises(pkg.picker.EppkError) as ei:
            g.pick_topk(bad, 2)
        assert "row 901" in str(ei.value)


@pytest.mark.parametrize("mode", ["peer", "host"])
@pytest.mark.parametrize("members", [2, 4])
def test_group_pipelined_learn_with_ageing_between_begins(pkg, orc, mode, members):
    """eppk_group_pick_stage_*: the two-set pipeline over a group, EPPK_PICK_LEARN chained on every member behind the gathered picks,
    the shim's ageing (epoch tick + eppk_group_index_evict_older_device) issued BETWEEN two begins with a set in flight.  The oracle
    replays the call order: batch k is scored against the index that batches 0..k-1 (and the evictions issued before begin k) left."""
    gm = {"peer": pkg.picker.GATHER_PEER, "host": pkg.picker.GATHER_HOST}[mode]
    wl = pkg.workload.make_workload(3, R=1200, P=900)
    batches = [wl.reqs] + [pkg.workload.make_requests(wl, 2000 + i) for i in range(5)]
    with pkg.DeviceGroup(wl.chain, [0] * members, max_pods=1024, max_blocks=wl.B, max_batch=1200, index_slots=1 << 17, gather=gm, min_shard=100) as g:
        g.publish(wl.pods)
        g.index_insert(wl.index_hashes, wl.index_pods)
        oix = orc.OracleIndex()
        oix.insert(wl.index_hashes, wl.index_pods)
        sb = [g.stage_buffers(0)[0], g.stage_buffers(1)[0]]
        n_of = lambda k: 1200 - 53 * k                    # ragged: other shard boundaries every batch
        expect = {}

        def oracle_step(k):
            reqs = batches[k % len(batches)][:n_of(k)]
            op, osc, _ = _oracle(orc, wl, reqs, oix=oix)
            oix.insert_picks(reqs, wl.B, op)
            expect[k] = (op, osc)

        def begin(k):
            n = n_of(k)
            np.copyto(sb[k & 1][:n], batches[k % len(batches)][:n])
            g.stage_begin(k & 1, n, learn=True)
            oracle_step(k)

        n_batches = 8
        begin(0)
        for k in range(1, n_batches + 1):
            if k < n_batches:
                begin(k)
                if k % 2 == 0:                            # ageing with set k & 1 in flight: behind its pick and update, ahead of the next begin's
                    e = g.index_advance_epoch()
                    assert e == oix.advance_epoch()
                    if e > 2:
                        g.index_evict_older_device(e - 1)
                        oix.evict_older(e - 1)
            picks, scores = g.stage_end((k - 1) & 1)
            _same(picks, scores, *expect[k - 1])
        for i in range(members):
            assert g.member_index_size(i) == oix.size(), i
            assert g.member_selfcheck(i) == 0 and g.member_launch_status(i) == 0
        assert g.index_trim_pods(40) == oix.trim_pods(1024, 40)
        op, osc, _ = _oracle(orc, wl, batches[1], oix=oix)
        for i in range(members):
            _same(*g.member_pick(i, batches[1]), op, osc)


def test_group_stage_masks_errors_and_plain_pipeline(pkg, orc):
    wl = pkg.workload.make_workload(5, R=900, P=4096, masked=True)
    with pkg.DeviceGroup(wl.chain, [0, 0, 0], max_pods=4096, max_blocks=wl.B, max_batch=900, index_slots=wl.index_slots, min_shard=64) as g:
        with pytest.raises(pkg.picker.EppkError):             # buffers first
            g.stage_begin(0, 10)
        rows0, m0 = g.stage_buffers(0, with_mask=True)
        rows1, _ = g.stage_buffers(1)
        with pytest.raises(pkg.picker.EppkError) as ei:       # no snapshot yet
            g.stage_begin(0, 10)
        assert ei.value.code == -4
        g.publish(wl.pods)
        g.index_insert(wl.index_hashes, wl.index_pods)
        J = (wl.P + 63) // 64
        np.copyto(rows0[:900], wl.reqs)
        m0[:900 * J] = wl.mask.reshape(-1)
        np.copyto(rows1[:500], wl.reqs[400:900])
        g.stage_begin(0, 900, use_mask=True)
        g.stage_begin(1, 500)
        with pytest.raises(pkg.picker.EppkError):             # begin twice
            g.stage_begin(1, 5)
        _same(*g.stage_end(0), *_oracle(orc, wl, wl.reqs, wl.mask)[:2])
        _same(*g.stage_end(1), *_oracle(orc, wl, wl.reqs[400:900])[:2])
        with pytest.raises(pkg.picker.EppkError):             # end without begin
            g.stage_end(1)
        g.stage_begin(0, 0)
        p, s = g.stage_end(0)
        assert p.size == 0
        rows1[123, 0] = np.uint64(200)                        # adapter 200: checked on the device, reported by end, naming the row
        g.stage_begin(1, 500)
        with pytest.raises(pkg.picker.EppkError) as ei:
            g.stage_end(1)
        assert ei.value.code == -1 and "row 123" in str(ei.value)
        np.copyto(rows1[:500], wl.reqs[:500])                 # ... and the set is usable again
        g.stage_begin(1, 500)
        _same(*g.stage_end(1), *_oracle(orc, wl, wl.reqs[:500])[:2])
        for i in range(3):
            assert g.member_launch_status(i) == 0
