"""GPU: the *_device entry points validate request rows IN THE KERNEL (include/eppk.h "trust contract"): an out-of-range row is
not scored (EPPK_NO_PICK / 0.0, every entry of its fallback list), every other row of the batch is bit-exact against the oracle,
and eppk_launch_status reports the sticky flag; eppk_index_insert_picks_device ignores a pick >= max_pods."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

Q, KV, L, PF = 1, 2, 3, 4


def _corrupt(reqs, rows, kinds, B):
    bad = reqs.copy()
    for r, kind in zip(rows, kinds):
        adapter = int(bad[r, 0] & np.uint64(0xFFFFFFFF))
        nb = int(bad[r, 0] >> np.uint64(32))
        if kind == "blocks":
            nb = B + 1 + (r % 5)
        elif kind == "blocks_huge":
            nb = 0xFFFFFFFF
        elif kind == "adapter_hi":
            adapter = 128 + (r % 7)
        elif kind == "adapter_neg":
            adapter = 0xFFFFFFFE            # -2
        bad[r, 0] = np.uint64(adapter) | (np.uint64(nb) << np.uint64(32))
    return bad


@pytest.mark.parametrize("chain", [
    [(Q, 2), (KV, 2), (L, 1), (PF, 3)],                 # fused kernel (lists, uniform route)
    [(PF, 3), (KV, 5)],                                 # fused kernel, interpreted tail
    [(L, 1), (Q, 2), (KV, 2), (Q, 1), (PF, 3)],         # generic kernel
])
@pytest.mark.parametrize("masked", [False, True])
def test_out_of_range_rows_get_no_pick_and_a_flag(pkg, orc, chain, masked):
    import torch
    wl = pkg.workload.make_workload(3, R=700, P=900, masked=masked)
    rows = [0, 1, 63, 64, 333, 699]
    kinds = ["blocks", "adapter_hi", "adapter_neg", "blocks_huge", "blocks", "adapter_hi"]
    bad = _corrupt(wl.reqs, rows, kinds, wl.B)
    with pkg.BatchedPicker(chain, max_pods=1024, max_blocks=wl.B, max_batch=wl.R, index_slots=wl.index_slots) as pk:
        pk.publish(wl.pods)
        pk.index_insert(wl.index_hashes, wl.index_pods)
        oix = orc.OracleIndex()
        oix.insert(wl.index_hashes, wl.index_pods)
        op, osc, _ = orc.pick_batch(chain, wl.pods, oix, wl.reqs, wl.B, wl.mask)
        d_reqs = torch.from_numpy(bad.view(np.int64)).cuda()
        d_mask = torch.from_numpy(wl.mask.view(np.int64)).cuda() if masked else None
        d_pick = torch.full((wl.R,), -7, dtype=torch.int32, device="cuda")
        d_score = torch.full((wl.R,), -7.0, dtype=torch.float64, device="cuda")
        assert pk.launch_status() == 0
        pk.pick_device(d_reqs.data_ptr(), wl.R, d_mask.data_ptr() if masked else None, d_pick.data_ptr(), d_score.data_ptr())
        assert pk.launch_status() == 1            # EPPK_LAUNCH_BAD_REQUEST_ROW (synchronises)
        assert pk.launch_status() == 0            # sticky until read, then cleared
        picks, scores = d_pick.cpu().numpy(), d_score.cpu().numpy()
        good = np.ones(wl.R, dtype=bool)
        good[rows] = False
        assert np.all(picks[rows] == -1) and np.all(scores[rows] == 0.0)
        assert np.array_equal(picks[good], op[good])
        assert np.array_equal(scores[good].view(np.uint64), osc[good].view(np.uint64))
        # ordered fallbacks of a bad row: all EPPK_NO_PICK
        k = 4
        d_tp = torch.full((wl.R * k,), -7, dtype=torch.int32, device="cuda")
        d_ts = torch.full((wl.R * k,), -7.0, dtype=torch.float64, device="cuda")
        pk._check(pk._lib.eppk_pick_topk_device(pk._ctx, d_reqs.data_ptr(), wl.R, d_mask.data_ptr() if masked else None, k,
                                                 d_tp.data_ptr(), d_ts.data_ptr(), None), "pick_topk_device")
        assert pk.launch_status() == 1
        tp, ts = d_tp.cpu().numpy().reshape(wl.R, k), d_ts.cpu().numpy().reshape(wl.R, k)
        assert np.all(tp[rows] == -1) and np.all(ts[rows] == 0.0)
        assert np.array_equal(tp[good][:, 0], op[good])
        # a clean batch afterwards leaves the flag clear
        d_ok = torch.from_numpy(wl.reqs.view(np.int64)).cuda()
        pk.pick_device(d_ok.data_ptr(), wl.R, d_mask.data_ptr() if masked else None, d_pick.data_ptr(), d_score.data_ptr())
        assert pk.launch_status() == 0
        assert np.array_equal(d_pick.cpu().numpy(), op)


def test_host_entry_point_still_rejects_before_launch(pkg):
    wl = pkg.workload.make_workload(3, R=64, P=200)
    bad = _corrupt(wl.reqs, [5], ["blocks"], wl.B)
    with pkg.BatchedPicker(wl.chain, max_pods=1024, max_blocks=wl.B, max_batch=wl.R, index_slots=wl.index_slots) as pk:
        pk.publish(wl.pods)
        with pytest.raises(pkg.picker.EppkError) as ei:
            pk.pick(bad)
        assert ei.value.code == -1 and "row 5" in str(ei.value)
        assert pk.launch_status() == 0


def test_insert_picks_ignores_out_of_range_picks(pkg, orc):
    import torch
    wl = pkg.workload.make_workload(3, R=256, P=500)
    with pkg.BatchedPicker(wl.chain, max_pods=512, max_blocks=wl.B, max_batch=wl.R, index_slots=1 << 15) as pk:
        pk.publish(wl.pods)
        oix = orc.OracleIndex()
        picks = (np.arange(wl.R, dtype=np.int32) * 7) % 500
        picks[[3, 77]] = [512, 2_000_000_000]          # >= max_pods: would shift into other pods' bits in a u16 lane word
        picks[9] = -1
        d_reqs = torch.from_numpy(wl.reqs.view(np.int64)).cuda()
        d_picks = torch.from_numpy(picks).cuda()
        pk.index_insert_picks_device(d_reqs.data_ptr(), d_picks.data_ptr(), wl.R)
        assert pk.launch_status() == 2                 # EPPK_LAUNCH_BAD_PICK
        ok = picks.copy()
        ok[[3, 77]] = -1
        oix.insert_picks(wl.reqs, wl.B, ok)
        assert pk.index_size() == oix.size()
        assert pk.index_selfcheck() == 0
        got, _ = pk.pick(wl.reqs)
        want, _, _ = orc.pick_batch(wl.chain, wl.pods, oix, wl.reqs, wl.B)
        assert np.array_equal(got, want)
