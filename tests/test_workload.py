"""CPU: the synthetic workload generator is deterministic and shaped as specified."""
import numpy as np


def test_splitmix64_reference_values(pkg):
    # splitmix64(seed=0): first outputs of the published generator
    out = pkg.workload.splitmix64(0, 3)
    assert [hex(int(x)) for x in out] == ["0xe220a8397b1dcdaf", "0x6e789e6aa1b965f4", "0x6c45d188009454f"]


def test_workload_is_deterministic_and_shaped(pkg):
    a = pkg.workload.make_workload(3, R=300, P=500)
    b = pkg.workload.make_workload(3, R=300, P=500)
    assert a.reqs.tobytes() == b.reqs.tobytes() and a.pods.tobytes() == b.pods.tobytes()
    assert a.reqs.shape == (300, 33) and a.pods.shape == (500,)
    assert a.pods["queue"].max() < 64 and (a.pods["kv_util"] <= 1.0).all() and set(np.unique(a.pods["max_lora"])) <= {4, 8}
    pc = lambda w: np.unpackbits(w.view(np.uint8).reshape(w.shape[0], -1), axis=1).sum(1)
    assert (pc(a.pods["active"]) <= a.pods["max_lora"]).all() and (pc(a.pods["waiting"]) <= 2).all()
    assert (a.n_blocks == 32).all() and a.adapter.min() >= -1 and a.adapter.max() < 128
    assert a.index_slots == 16384 and a.index_hashes.shape[0] == 256 * 16 * 8
    # shared prefixes: requests of the most popular group share their first 16 hashes, tails are unique
    first = a.reqs[:, 1]
    vals, counts = np.unique(first, return_counts=True)
    top = vals[np.argmax(counts)]
    same = a.reqs[first == top]
    assert same.shape[0] > 10 and (same[:, 1:17] == same[0, 1:17]).all()
    assert np.unique(a.reqs[:, 17]).shape[0] == 300
    # every shared block of every group is in the index
    assert np.isin(a.reqs[:, 1:17].ravel(), a.index_hashes).all()
    assert not np.isin(a.reqs[:, 17:].ravel(), a.index_hashes).any()


def test_request_seed_changes_only_requests(pkg):
    a = pkg.workload.make_workload(5, R=64)
    b = pkg.workload.make_workload(5, R=64, req_seed=12345)
    assert a.pods.tobytes() == b.pods.tobytes() and np.array_equal(a.index_hashes, b.index_hashes)
    assert a.reqs.tobytes() != b.reqs.tobytes()


def test_masked_workload(pkg):
    w = pkg.workload.make_workload(2, R=50, P=100, masked=True)
    assert w.mask.shape == (50, 2) and (w.mask[:, 1] >> np.uint64(36) == 0).all()


def test_returning_rows_bring_back_rows_of_an_earlier_batch(pkg):
    """workload.returning_rows / make_requests(revisit_of=, revisit_frac=): the declared returning-request workload (a returning request = a
    row of an earlier batch, whole -- shared blocks AND its own tail -- at a scattered position of an otherwise new batch)."""
    w = pkg.workload.make_workload(5, R=512)
    earlier = w.reqs
    fresh = pkg.workload.make_requests(w, 777)
    assert not (fresh[:, 17:] == earlier[:, 17:]).any()
    for f in (0.0, 0.25, 0.5, 1.0):
        rows = pkg.workload.returning_rows(fresh, earlier, f, 42)
        back = (rows == earlier).all(axis=1)
        assert ((rows == fresh).all(axis=1) | back).all() and rows.shape == fresh.shape
        assert abs(back.mean() - f) < 0.08, (f, back.mean())
        if 0.0 < f < 1.0:      # scattered, not a block at the front
            assert back[: 256].any() and back[256:].any() and not back[: 64].all()
    assert np.array_equal(pkg.workload.returning_rows(fresh, earlier, 0.5, 42), pkg.workload.returning_rows(fresh, earlier, 0.5, 42))
    assert np.array_equal(pkg.workload.make_requests(w, 777, revisit_of=earlier, revisit_frac=0.5), pkg.workload.returning_rows(fresh, earlier, 0.5, 777))
