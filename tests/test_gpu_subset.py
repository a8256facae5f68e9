"""GPU: the subset filter of a whole batch resolved on the device (include/eppk.h eppk_subset_masks / eppk_pick_batch_subset;
SEMANTICS.md §5a).  The masks the device builds from entry fingerprints must equal, bit for bit, the string-exact masks of
eppk_subset_mask AND of the oracle's restatement of request.go:104-133 -- on the reference's own cases
(tests/golden/reference_cases.json) and on random batches with shared addresses, IPv6 literals, ports that do not exist, empty
filters and requests without a filter; the picks through eppk_pick_batch_subset must equal the oracle's picks under those masks."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
DOC = json.load(open(os.path.join(HERE, "golden", "reference_cases.json")))

Q, KV, L, PF = 1, 2, 3, 4
CHAIN = [(Q, 2), (KV, 2), (L, 1), (PF, 3)]


def _string_masks(pkg, orc, endpoints, filters):
    J = (len(endpoints) + 63) // 64
    out = np.zeros((len(filters), J), dtype=np.uint64)
    for r, f in enumerate(filters):
        m, n = pkg.picker.subset_mask(endpoints, f)
        om, on = orc.subset_mask([e.address for e in endpoints], [e.port for e in endpoints], f)
        assert n == on and np.array_equal(m, om)
        out[r] = m
    return out


def _picker(pkg, wl_pods, endpoints, B=4, R=64):
    pk = pkg.BatchedPicker(CHAIN, max_pods=max(64, len(endpoints)), max_blocks=B, max_batch=R, index_slots=1024)
    pk.publish(wl_pods)
    pk.set_addresses(endpoints)
    return pk


def test_reference_cases_through_the_device(pkg, orc):
    """Every filter value the reference's request_test.go uses, as ONE batch per pod list."""
    by_pods = {}
    for case in DOC["cases"]:
        pods = tuple((p["address"], p["port"]) for p in case["pods"])
        if not pods:
            continue
        filters = by_pods.setdefault(pods, [None, ""])
        if "table" in case:
            filters += [row["filter_value"] for row in case["table"]]
        else:
            md = case["metadata"]
            if md["kind"] == "string":
                filters.append(md["value"])
            elif md["kind"] == "list":
                filters.append(",".join(md["values"]))
            if case.get("header"):
                filters.append(case["header"])
    assert by_pods
    for pods, filters in by_pods.items():
        endpoints = [pkg.picker.Endpoint(a, p) for a, p in pods]
        rows = pkg.workload.make_pods(3, len(endpoints), 128)
        with _picker(pkg, rows, endpoints, R=max(64, len(filters))) as pk:
            got = pk.subset_masks(filters)
        assert np.array_equal(got, _string_masks(pkg, orc, endpoints, filters)), (pods, filters)


@pytest.mark.parametrize("P,R", [(5, 40), (64, 64), (130, 200), (1000, 512), (4096, 256)])
def test_random_batches_masks_and_picks(pkg, orc, P, R):
    rng = np.random.default_rng(100 + P)
    # addresses shared by several pods (ports differ), a few IPv6 literals
    n_addr = max(1, P // 3)
    addr_of = rng.integers(0, n_addr, P)
    def addr(i):
        return f"10.{(i >> 16) & 255}.{(i >> 8) & 255}.{i & 255}" if i % 7 else f"fd00::{i:x}"
    used = set()
    endpoints = []
    for p in range(P):
        a = addr(int(addr_of[p]))
        port = 8000 + int(rng.integers(0, 4))
        while (a, port) in used:
            port += 1
        used.add((a, port))
        endpoints.append(pkg.picker.Endpoint(a, str(port)))
    filters = []
    for r in range(R):
        u = rng.random()
        if u < 0.15:
            filters.append(None)
        elif u < 0.22:
            filters.append("")
        elif u < 0.27:
            filters.append(" , ,")
        else:
            k = int(rng.integers(1, 12)) if u < 0.9 else int(rng.integers(60, 150))      # (more than 64 entries: two passes of the wave)
            parts = []
            for _ in range(k):
                e = endpoints[int(rng.integers(0, P))]
                v = rng.random()
                host = f"[{e.address}]" if ":" in e.address else e.address
                if v < 0.4:
                    parts.append(f"{host}:{e.port}")
                elif v < 0.7:
                    parts.append(e.address)                     # all ports (an IPv6 literal without brackets fails SplitHostPort: also "all ports")
                elif v < 0.8:
                    parts.append(f"  {host}:{int(e.port) + 100}\t")   # a port nobody listens on
                elif v < 0.9:
                    parts.append("192.0.2.1:80")                # an address nobody has
                else:
                    parts.append(f" {e.address} ")
            filters.append(",".join(parts))
    B = 4
    wl = pkg.workload.make_workload(3, R=R, P=P, B=B)
    with pkg.BatchedPicker(wl.chain, max_pods=max(64, P), max_blocks=B, max_batch=R, index_slots=wl.index_slots) as pk:
        pk.publish(wl.pods)
        pk.index_insert(wl.index_hashes, wl.index_pods)
        with pytest.raises(pkg.EppkError):
            pk.subset_masks(filters[:1])                        # no addresses yet
        pk.set_addresses(endpoints)
        got = pk.subset_masks(filters)
        want = _string_masks(pkg, orc, endpoints, filters)
        assert np.array_equal(got, want)
        picks, scores = pk.pick_subset(wl.reqs, filters)
        p2, s2 = pk.pick(wl.reqs, want)
        assert np.array_equal(picks, p2) and np.array_equal(scores.view(np.uint64), s2.view(np.uint64))
    oix = orc.OracleIndex()
    oix.insert(wl.index_hashes, wl.index_pods)
    op, osc, _ = orc.pick_batch(wl.chain, wl.pods, oix, wl.reqs, B, want)
    assert np.array_equal(picks, op) and np.array_equal(scores.view(np.uint64), osc.view(np.uint64))


def test_holes_and_republish(pkg, orc):
    """A hole has no address; a publish with another pod count invalidates the table until set_addresses is called again."""
    P = 70
    rows = pkg.workload.make_pods(5, P, 128)
    rows["flags"][[3, 64]] = 1
    endpoints = [None if i in (3, 64) else pkg.picker.Endpoint(f"10.0.0.{i}", "8000") for i in range(P)]
    with pkg.BatchedPicker(CHAIN, max_pods=128, max_blocks=4, max_batch=8, index_slots=256) as pk:
        pk.publish(rows)
        pk.set_addresses(endpoints)
        m = pk.subset_masks(["10.0.0.3,10.0.0.4,10.0.0.64:8000,10.0.0.69", None])
        assert [int(x) for x in m[0]] == [1 << 4, 1 << 5]
        assert [int(x) for x in m[1]] == [(1 << 64) - 1, (1 << 6) - 1]
        pk.publish(rows[:10])
        with pytest.raises(pkg.EppkError):
            pk.subset_masks(["10.0.0.4"])
        pk.set_addresses(endpoints[:10])
        assert int(pk.subset_masks(["10.0.0.4"])[0, 0]) == 1 << 4


@pytest.mark.parametrize("chain", [
    [(Q, 2), (KV, 2), (L, 1), (PF, 3)],                 # the headline chain
    [(PF, 3), (KV, 5)],                                 # the reference example's decode profile
    [(L, 1), (Q, 2), (KV, 2), (Q, 1), (PF, 3)],         # a chain only the generic kernel serves
    [(Q, 1), (KV, 1)],                                  # no prefix scorer: nothing is looked up
])
@pytest.mark.parametrize("P,B", [(300, 8), (4096, 8), (2000, 40)])
def test_candidate_major_kernel(pkg, orc, chain, P, B):
    """eppk_pick_batch_candidates_device: masks that leave 0 .. 150 candidates (up to three passes of 64 lanes), holes among
    them, chains of more than 32 blocks, picks and ordered fallbacks (k = 3) against the oracle AND against the general masked
    entry points; an out-of-range row gets EPPK_NO_PICK and raises the launch-status flag."""
    rng = np.random.default_rng(P + B)
    R = 300
    wl = pkg.workload.make_workload(3, R=R, P=P, B=B)
    pods = wl.pods.copy()
    pods["flags"] = (rng.random(P) < 0.1).astype(np.uint32)
    J = (P + 63) // 64
    mask = np.zeros((R, J), dtype=np.uint64)
    for r in range(R):
        n = int(rng.integers(0, 71)) if r % 5 else int(rng.integers(0, 3))
        if r % 17 == 0:
            n = int(rng.integers(100, 151))
        for p in rng.choice(P, size=min(n, P), replace=False):
            mask[r, p >> 6] |= np.uint64(1) << np.uint64(p & 63)
    reqs = wl.reqs.copy()
    reqs[7, 0] = np.uint64(200) | (np.uint64(1) << np.uint64(32))          # adapter 200: out of range
    with pkg.BatchedPicker(chain, max_pods=max(64, P), max_blocks=B, max_batch=R, index_slots=wl.index_slots) as pk:
        pk.publish(pods)
        pk.index_insert(wl.index_hashes, wl.index_pods)
        assert pk.launch_status() == 0
        picks, scores = pk.pick_candidates(reqs, mask)
        assert pk.launch_status() == 1 and picks[7, 0] == -1 and scores[7, 0] == 0.0
        tp, ts = pk.pick_candidates(wl.reqs, mask, 3)
        gp, gs = pk.pick(wl.reqs, mask)
    oix = orc.OracleIndex()
    oix.insert(wl.index_hashes, wl.index_pods, snapshot=pods)
    op, osc, _ = orc.pick_batch(chain, pods, oix, wl.reqs, B, mask)
    ok = np.arange(R) != 7
    assert np.array_equal(picks[ok, 0], op[ok]) and np.array_equal(scores[ok, 0].view(np.uint64), osc[ok].view(np.uint64))
    assert np.array_equal(gp, op) and np.array_equal(gs.view(np.uint64), osc.view(np.uint64))
    otp, ots = orc.pick_topk(chain, pods, oix, wl.reqs, 3, mask)
    assert np.array_equal(tp, otp) and np.array_equal(ts.view(np.uint64), ots.view(np.uint64))


def test_large_single_pick_batches_of_few_candidates_take_the_general_route(pkg, orc):
    """eppk_pick_batch_candidates_device with k = 1 hands batches of 8192 requests or more to the general masked route where pick_quad_kernel
    serves them (it parks every row whose candidates miss a QUEUE extreme and scores four at a time: 64k x 8 candidates 71 us against
    210 through the candidate-major kernel): same picks, same scores, the oracle's; ordered fallbacks from 4096 requests on (top-3: 122 us against 559)."""
    R, P = 8192, 4096
    wl = pkg.workload.make_workload(5, R=R, P=P, masked=True)
    rng = np.random.default_rng(8)
    J = (P + 63) // 64
    mask = np.zeros((R, J), dtype=np.uint64)
    pods = rng.integers(0, P, (R, 6))
    for j in range(6):
        np.bitwise_or.at(mask, (np.arange(R), pods[:, j] // 64), np.uint64(1) << (pods[:, j] % 64).astype(np.uint64))
    mask[11] = 0
    with pkg.BatchedPicker(wl.chain, max_pods=P, max_blocks=wl.B, max_batch=R, index_slots=wl.index_slots) as pk:
        pk.publish(wl.pods)
        pk.index_insert(wl.index_hashes, wl.index_pods)
        l0, _ = pk.quad_stats()
        picks, scores = pk.pick_candidates(wl.reqs, mask)
        l1, d1 = pk.quad_stats()
        tp, ts = pk.pick_candidates(wl.reqs, mask, 2)
        l2, _ = pk.quad_stats()
    if os.environ.get("EPPK_QUAD", "1") != "0" and os.environ.get("EPPK_LISTS", "1") != "0" and int(os.environ.get("EPPK_QUAD_MIN", "4096")) <= R:
        assert l1 == l0 + 1 and l2 == l1 + 1, "k = 1 and k = 2 (from 4096 requests on): one pick_quad_kernel launch each"
        assert d1 <= R // 16
    oix = orc.OracleIndex()
    oix.insert(wl.index_hashes, wl.index_pods)
    op, osc, _ = orc.pick_batch(wl.chain, wl.pods, oix, wl.reqs, wl.B, mask)
    assert picks[11, 0] == -1
    assert np.array_equal(picks[:, 0], op) and np.array_equal(scores[:, 0].view(np.uint64), osc.view(np.uint64))
    otp, ots = orc.pick_topk(wl.chain, wl.pods, oix, wl.reqs, 2, mask)
    assert np.array_equal(tp, otp) and np.array_equal(ts.view(np.uint64), ots.view(np.uint64))
