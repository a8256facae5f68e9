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
