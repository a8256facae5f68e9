"""eppk_host_staging / eppk_pick_batch_staged: rows (and mask rows) built in the library's pinned buffers give the picks and scores of
eppk_pick_batch on the same rows; eppk_pick_batch itself uploads pageable rows in chunks."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("R,P,masked", [(1, 64, False), (700, 1000, False), (700, 1000, True), (40000, 4096, False), (33000, 4096, True)])
def test_staged_pick_equals_pick(pkg, orc, R, P, masked):
    wl = pkg.workload.make_workload(5, R=R, P=P, n_groups=32, masked=masked)
    with pkg.BatchedPicker(wl.chain, max_pods=4096, max_blocks=wl.B, max_batch=R + 5, index_slots=wl.index_slots) as pk:
        pk.publish(wl.pods)
        pk.index_insert(wl.index_hashes, wl.index_pods)
        want_p, want_s = pk.pick(wl.reqs, wl.mask)                   # (40000 x 264 B = 10 MB: five chunks)
        st_reqs, st_mask = pk.staging(with_mask=masked)
        assert st_reqs.shape == (R + 5, 1 + wl.B)
        st_reqs[:R] = wl.reqs
        if masked:
            J = (P + 63) // 64
            st_mask[:R * J].reshape(R, J)[:] = wl.mask
        got_p, got_s = pk.pick_staged(R, use_mask=masked)
        assert np.array_equal(got_p, want_p) and np.array_equal(got_s.view(np.uint64), want_s.view(np.uint64))
        # and both equal the oracle
        oix = orc.OracleIndex()
        oix.insert(wl.index_hashes, wl.index_pods)
        op, os_, _ = orc.pick_batch(wl.chain, wl.pods, oix, wl.reqs, wl.B, wl.mask)
        assert np.array_equal(got_p, op) and np.array_equal(got_s.view(np.uint64), os_.view(np.uint64))
        # a bad row in the staging buffer is refused like anywhere else
        st_reqs[0, 0] = np.uint64(1000) << np.uint64(32)
        with pytest.raises(Exception):
            pk.pick_staged(R, use_mask=masked)
