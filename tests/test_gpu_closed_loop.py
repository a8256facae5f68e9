"""GPU: the closed loop the prefix scorer relies on (docs/proposals/0602-…/README.md:101-112) at BASELINE.json's full size --
pick -> index[hash[r][i]] U= {pick[r]} (eppk_index_insert_picks_device) -> next, DIFFERENT batch -> ... with ageing in between --
for 8 generations against the oracle running the same loop: picks and scores bit-exact on the EVOLVED index every generation,
index sizes equal, no dropped inserts, index invariants intact.  Plus the other BASELINE configs at their full sizes."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _same(picks, scores, op, osc, what=""):
    bad = np.nonzero(picks != op)[0]
    assert bad.size == 0, f"{what}: {bad.size} picks differ, first {bad[:5]}: gpu {picks[bad[:5]]} oracle {op[bad[:5]]}"
    assert np.array_equal(scores.view(np.uint64), osc.view(np.uint64)), what


@pytest.mark.parametrize("R,slots,async_evict,fused", [(65536, 1 << 24, True, False), (65536, 1 << 24, True, True), (8192, 1 << 21, False, True),
                                                       (8192, 1 << 21, False, False)])
def test_eight_closed_loop_generations_at_full_size(pkg, orc, R, slots, async_evict, fused):
    """fused: eppk_pick_learn_device (one call; the pick kernel tells the update which pairs it has already seen in the index) instead of
    eppk_pick_batch_device + eppk_index_insert_picks_device -- same picks, same scores, same index."""
    import torch
    cores = os.cpu_count() or 1
    wl = pkg.workload.make_workload(5, R=R)
    assert wl.P == 4096 and wl.B == 32
    batches = [wl.reqs] + [pkg.workload.make_requests(wl, 4242 + i) for i in range(3)]
    with pkg.BatchedPicker(wl.chain, max_pods=wl.P, max_blocks=wl.B, max_batch=R, index_slots=slots) as pk:
        pk.publish(wl.pods)
        pk.index_insert(wl.index_hashes, wl.index_pods)
        oix = orc.OracleIndex()
        oix.insert(wl.index_hashes, wl.index_pods)
        d_batches = [torch.from_numpy(b.view(np.int64)).cuda() for b in batches]
        d_pick = torch.empty(R, dtype=torch.int32, device="cuda")
        d_score = torch.empty(R, dtype=torch.float64, device="cuda")
        st = torch.cuda.Stream()
        for gen in range(8):
            b = gen % len(batches)
            if fused:
                pk.pick_learn_device(d_batches[b].data_ptr(), R, None, d_pick.data_ptr(), d_score.data_ptr(), st.cuda_stream)
            else:
                pk.pick_device(d_batches[b].data_ptr(), R, None, d_pick.data_ptr(), d_score.data_ptr(), st.cuda_stream)
                pk.index_insert_picks_device(d_batches[b].data_ptr(), d_pick.data_ptr(), R, st.cuda_stream)
            st.synchronize()
            op, osc, _ = orc.pick_batch(wl.chain, wl.pods, oix, batches[b], wl.B, threads=cores)
            _same(d_pick.cpu().numpy(), d_score.cpu().numpy(), op, osc, f"generation {gen}")
            oix.insert_picks(batches[b], wl.B, op)
            assert pk.index_size() == oix.size(), gen
            if gen in (2, 5):                                  # ageing: hashes not re-inserted for 2 epochs go
                e = pk.index_advance_epoch()
                assert e == oix.advance_epoch()
            if gen == 5:
                if async_evict:
                    pk.index_evict_older_device(e - 1, st.cuda_stream)
                    st.synchronize()
                    assert oix.evict_older(e - 1) > 0
                else:
                    assert pk.index_evict_older(e - 1) == oix.evict_older(e - 1) > 0
                assert pk.index_size() == oix.size()
        assert pk.index_dropped() == 0 and pk.launch_status() == 0
        assert pk.index_selfcheck() == 0


@pytest.mark.parametrize("config", [2, 3, 4])
def test_other_baseline_configs_at_full_size(pkg, orc, config):
    """BASELINE.json configs[1..3] at their own sizes (4k x 256, 8k x 1024 with prefix probe, 16k x 2048 x 128 adapters)."""
    cores = os.cpu_count() or 1
    wl = pkg.workload.make_workload(config)
    with pkg.BatchedPicker(wl.chain, max_pods=wl.P, max_blocks=wl.B, max_batch=wl.R, index_slots=wl.index_slots) as pk:
        pk.publish(wl.pods)
        oix = orc.OracleIndex()
        if wl.index_slots:
            pk.index_insert(wl.index_hashes, wl.index_pods)
            oix.insert(wl.index_hashes, wl.index_pods)
        picks, scores = pk.pick(wl.reqs)
        op, osc, _ = orc.pick_batch(wl.chain, wl.pods, oix, wl.reqs, wl.B, threads=cores)
        _same(picks, scores, op, osc, wl.name)


def test_a_table_past_half_of_its_limit_keeps_the_fast_insert_path(pkg):
    """The capacity test of the index update is per LAUNCH when every pair of the launch fits as a new key; otherwise every new key is
    BOOKED (one sharded atomic) before it is claimed, and the launch admits exactly what was left -- a post-route update into a table
    more than half-way to its limit used to cost 7 x the time of the same update into a roomy one (every new key read all the counters).
    The same three updates go into a table they fill to 3/4 of its limit and into one eight times as large: same index sizes, nothing
    dropped, and the last update -- the one that no longer fits launch-wide -- within 1.5 x of the roomy table's."""
    import torch
    R = 32768
    wl = pkg.workload.make_workload(5, R=R)
    batches = [wl.reqs] + [pkg.workload.make_requests(wl, 900 + i) for i in range(2)]
    times, sizes = {}, {}
    for name, slots in (("tight", 1 << 22), ("roomy", 1 << 25)):
        with pkg.BatchedPicker(wl.chain, max_pods=wl.P, max_blocks=wl.B, max_batch=R, index_slots=slots) as pk:
            pk.publish(wl.pods)
            pk.index_insert(wl.index_hashes, wl.index_pods)
            d_batches = [torch.from_numpy(b.view(np.int64)).cuda() for b in batches]
            d_pick = torch.zeros(R, dtype=torch.int32, device="cuda")           # every request "routed" to pod 0
            st = torch.cuda.Stream()
            ts = []
            for rep in range(3):                                                # (the third batch is what is timed; repeat it as known pairs for a steadier figure)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st)
                pk.index_insert_picks_device(d_batches[rep].data_ptr(), d_pick.data_ptr(), R, st.cuda_stream)
                e1.record(st)
                st.synchronize()
                ts.append(e0.elapsed_time(e1))
            times[name], sizes[name] = ts, pk.index_size()
            assert pk.index_dropped() == 0 and pk.launch_status() == 0 and pk.index_selfcheck() == 0
    # tight: limit 2 Mi keys; 4096 + 3 x 512 Ki = 1.5 Mi keys at the end, and the third launch's 1 Mi pairs no longer fit launch-wide
    assert sizes["tight"] == sizes["roomy"] == 4096 + 3 * R * 16
    assert times["tight"][2] <= 1.5 * times["roomy"][2] + 0.05, times
