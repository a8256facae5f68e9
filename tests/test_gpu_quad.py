"""pick_quad_kernel (four requests per wavefront, csrc/eppk_kernels.hip.h) against the oracle: the shapes of a request it scores
itself, the ones it defers to pick_fast_kernel's work-list instantiation, and the bookkeeping between the two launches.

Every case also runs with the route switched off (EPPK_QUAD=0): same picks, same scores, same probe statistics.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def assert_same(picks, scores, opicks, oscores):
    bad = np.nonzero(picks != opicks)[0]
    assert bad.size == 0, f"{bad.size} picks differ, first at {bad[:5]}: gpu {picks[bad[:5]]} oracle {opicks[bad[:5]]}"
    sb = np.nonzero(scores.view(np.uint64) != oscores.view(np.uint64))[0]
    assert sb.size == 0, f"{sb.size} scores differ bitwise, first {sb[:5]}: {scores[sb[:5]]} vs {oscores[sb[:5]]}"


class quad_env:
    """EPPK_QUAD / EPPK_QUAD_MIN are read when a context is created."""
    def __init__(self, on):
        self.on = on

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in ("EPPK_QUAD", "EPPK_QUAD_MIN")}
        os.environ["EPPK_QUAD"] = "1" if self.on else "0"
        os.environ["EPPK_QUAD_MIN"] = "4"                    # (by default only batches of 4096 requests and more take the route)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def pairs_by_pod(sets):
    """[(hashes, pods)] insert calls, one per pod in ascending order: every hash's list then holds its pods in ascending order, so
    hashes with the same pod set have bitwise identical lists (a single call appends in whatever order the device gets to them)."""
    by_pod = {}
    for h, pods in sets.items():
        for p in pods:
            by_pod.setdefault(int(p), []).append(h)
    return [(np.array(hs, dtype=np.uint64), np.full(len(hs), p, dtype=np.uint32)) for p, hs in sorted(by_pod.items())]


def run(pkg, orc, wl, sets, reqs=None, slots=None, launches=1, expect_quad=True, mask=None):
    """Index = {hash: pods} built pod by pod; `launches` picks of the same batch; returns (quad launches, deferred requests)."""
    reqs = wl.reqs if reqs is None else reqs
    if slots is None:                                        # load <= 1/4, libeppk's recommended sizing
        slots = max(wl.index_slots, 64)
        while slots < 4 * len(sets):
            slots *= 2
    calls = pairs_by_pod(sets)
    oix = orc.OracleIndex()
    for h, p in calls:
        oix.insert(h, p)
    opicks, oscores, _ = orc.pick_batch(wl.chain, wl.pods, oix, reqs, wl.B, mask)
    out = {}
    for on in (True, False):
        with quad_env(on):
            with pkg.BatchedPicker(wl.chain, max_pods=wl.P, max_blocks=wl.B, max_batch=max(reqs.shape[0], 1),
                                   index_slots=slots) as pk:
                pk.publish(wl.pods)
                for h, p in calls:
                    pk.index_insert(h, p)
                pk.profile(True)
                for _ in range(launches):
                    picks, scores = pk.pick(reqs, mask)
                    assert_same(picks, scores, opicks, oscores)
                stats = pk.profile_bytes()
                out[on] = (pk.quad_stats(), stats)
                assert pk.index_selfcheck() == 0
    (ql, qd), st_on = out[True]
    (zl, zd), st_off = out[False]
    assert (zl, zd) == (0, 0), "EPPK_QUAD=0 must keep every launch on the fast kernel"
    assert st_on == st_off, f"probe statistics differ between the routes: {st_on} vs {st_off}"
    if expect_quad:
        assert ql >= 1, "the quad route was not taken"
    return ql, qd


def group_sets(wl, n_pods=None):
    """{hash: pod tuple} of the workload's pre-populated index (make_workload: every shared block of a group on the group's pods)."""
    sets = {}
    for h, p in zip(wl.index_hashes.tolist(), wl.index_pods.tolist()):
        sets.setdefault(h, []).append(p)
    return {h: tuple(sorted(set(ps)))[:n_pods] for h, ps in sets.items()}


def hashes_of(wl, reqs=None):
    reqs = wl.reqs if reqs is None else reqs
    return reqs[:, 1:1 + wl.B]


@pytest.mark.parametrize("R,P", [(1024, 4096), (1023, 4096), (5, 4096), (4, 64), (777, 1000), (2050, 2048), (96, 700)])
def test_common_shape_is_scored_by_the_quad_kernel(pkg, orc, R, P):
    wl = pkg.workload.make_workload(5, R=R, P=P, n_groups=32)
    ql, qd = run(pkg, orc, wl, group_sets(wl))
    assert ql == 1 and qd == 0, f"{qd} of {R} requests deferred although every request has the common shape"


@pytest.mark.parametrize("chain", [[(4, 3)], [(4, 3), (3, 1)], [(3, 1), (4, 3)], [(1, 2), (2, 2), (4, 3)], [(2, 1), (3, 2), (4, -3)]])
def test_chains(pkg, orc, chain):
    wl = pkg.workload.make_workload(3, R=640, P=900, n_groups=24)
    wl.chain = [(int(k), int(w)) for k, w in chain]
    ql, qd = run(pkg, orc, wl, group_sets(wl))
    assert ql == 1 and qd == 0


def test_long_matches_and_ragged_block_counts(pkg, orc):
    """Requests whose cached prefix is 0, 1, 5, 16, 17, 19, 20, 21, 24, 31 or all 32 blocks long (steps 5..7 of the probe are fetched on
    demand, lists of hits 16..31 are compared), with n_blocks from 0 to 32."""
    wl = pkg.workload.make_workload(5, R=704, P=4096, n_groups=16)
    sets = group_sets(wl)
    hs = hashes_of(wl)
    reqs = wl.reqs.copy()
    want = [0, 1, 5, 16, 17, 19, 20, 21, 24, 31, 32]
    half = wl.B // 2
    gp = {}
    for r in range(wl.R):
        n_hit = want[r % len(want)]
        pods = sets[int(hs[r, 0])]                       # the group's pods: the tail blocks are cached on the same ones
        for b in range(half, n_hit):
            sets[int(hs[r, b])] = pods
        gp[r] = n_hit
    # ragged n_blocks (the header's high half): some rows end before their cached prefix does
    nbs = [32, 32, 31, 17, 16, 5, 1, 0, 20, 21, 24]
    hdr = reqs[:, 0].copy()
    for r in range(wl.R):
        nb = nbs[(r // len(want)) % len(nbs)]
        hdr[r] = (hdr[r] & np.uint64(0xFFFFFFFF)) | (np.uint64(nb) << np.uint64(32))
    reqs[:, 0] = hdr
    # requests with n_hit < 16 miss INSIDE the shared prefix: drop those blocks from the index for a few groups' worth of rows
    for r in range(wl.R):
        if gp[r] < half:
            # a private copy of the row's hashes: flip a bit from block n_hit on so that the walk ends there
            reqs[r, 1 + gp[r]:] ^= np.uint64(0x9E3779B97F4A7C15)
    ql, qd = run(pkg, orc, wl, sets, reqs=reqs)
    assert ql == 1 and qd == 0


def test_lists_of_up_to_24_pods_and_overflowed_ones(pkg, orc):
    wl = pkg.workload.make_workload(5, R=512, P=4096, n_groups=8)
    base = group_sets(wl)
    rng = np.random.default_rng(7)
    sizes = {}
    sets = {}
    for h, pods in base.items():
        g = pods                                             # (the group is identified by its pod tuple)
        if g not in sizes:
            sizes[g] = (1, 7, 16, 17, 23, 24, 25, 40)[len(sizes) % 8]
        extra = rng.choice(wl.P, size=64, replace=False).tolist()
        rng2 = np.random.default_rng(hash(g) & 0xFFFF)       # the same pods for every block of the group
        more = rng2.choice(wl.P, size=64, replace=False).tolist()
        allp = list(dict.fromkeys(list(pods) + more))[:sizes[g]]
        sets[h] = tuple(sorted(allp))
    ql, qd = run(pkg, orc, wl, sets)
    assert ql == 1
    assert qd > 0, "lists of 25 and 40 pods have overflowed: those requests belong to the dense rows"
    assert qd < wl.R, "lists of up to 24 pods are the quad kernel's"


def test_differing_lists_are_deferred(pkg, orc):
    wl = pkg.workload.make_workload(5, R=384, P=4096, n_groups=8)
    sets = group_sets(wl)
    keys = sorted(sets)
    for n, h in enumerate(keys):
        if n % 5 == 0:                                       # one block in five is cached on one more pod
            sets[h] = tuple(sorted(set(sets[h]) | {(n * 37) % wl.P}))
    ql, qd = run(pkg, orc, wl, sets)
    assert ql == 1 and qd > 0


def test_reserved_hashes_are_deferred(pkg, orc):
    wl = pkg.workload.make_workload(5, R=256, P=4096, n_groups=8)
    sets = group_sets(wl)
    reqs = wl.reqs.copy()
    reqs[3, 1 + 2] = np.uint64(0)                            # reserved: presence words behind the table
    reqs[70, 1 + 20] = np.uint64(0xFFFFFFFFFFFFFFFF)
    reqs[71, 1] = np.uint64(0)
    sets[0] = (5, 9)
    ql, qd = run(pkg, orc, wl, sets, reqs=reqs)
    assert ql == 1 and 3 <= qd <= 8


def test_displaced_keys_in_a_crowded_table(pkg, orc):
    """A table near its load limit: buckets overflow, keys live in later buckets; the first missing key of a request may have to be
    followed through the chain (and is found there, or confirmed absent)."""
    wl = pkg.workload.make_workload(5, R=1024, P=4096, n_groups=120)       # 1920 keys
    ql, qd = run(pkg, orc, wl, group_sets(wl), slots=4096)                 # 512 buckets x 7 words: load 0.54
    assert ql == 1 and qd == 0


def test_backoff_on_a_workload_of_differing_lists(pkg, orc):
    """Every request defers: the library stops trying after the first report (and tries again 64 launches later); results never change."""
    wl = pkg.workload.make_workload(5, R=256, P=4096, n_groups=4)
    sets = group_sets(wl)
    for n, h in enumerate(sorted(sets)):
        sets[h] = tuple(sorted(set(sets[h]) | {(n * 131 + 1) % wl.P}))     # a different extra pod on every block
    ql, qd = run(pkg, orc, wl, sets, launches=80)
    assert 1 <= ql <= 4, f"{ql} of 80 launches went through the quad kernel"     # the first one, and one more after the pause of 64
    assert qd == ql * wl.R


@pytest.mark.parametrize("n_streams", [1, 3, 5, 10])
def test_many_launches_in_flight_on_several_streams(pkg, orc, n_streams):
    """Picks enqueued back to back on several caller streams without any synchronisation in between: every stream has its own
    work-list buffer (launches of one stream are ordered, launches of different streams share nothing); a ninth and a tenth stream
    stay on the fast kernel.  Half of the batches defer a few requests (differing lists), so the work-list pass has work."""
    import torch
    wl = pkg.workload.make_workload(5, R=2048, P=4096, n_groups=16)
    sets = group_sets(wl)
    batches = [wl.reqs] + [pkg.workload.make_requests(wl, 1000 + i) for i in range(5)]
    first, counts = np.unique(np.concatenate([hashes_of(wl, b)[:, 0] for b in batches]), return_counts=True)
    rare = int(first[np.argmin(counts)])                     # the least popular prefix group: one of its blocks sits on one more pod
    sets[rare] = tuple(sorted(set(sets[rare]) | {(sets[rare][0] + 1) % wl.P}))
    assert 0 < counts.min() < len(batches) * wl.R // 16      # (few enough deferrals that the route is not paused)
    calls = pairs_by_pod(sets)
    oix = orc.OracleIndex()
    for h, p in calls:
        oix.insert(h, p)
    want = [orc.pick_batch(wl.chain, wl.pods, oix, b, wl.B)[:2] for b in batches]
    dev = torch.device("cuda", 0)
    with quad_env(True):
        with pkg.BatchedPicker(wl.chain, max_pods=wl.P, max_blocks=wl.B, max_batch=wl.R, index_slots=4 * wl.index_slots) as pk:
            pk.publish(wl.pods)
            for h, p in calls:
                pk.index_insert(h, p)
            d_batches = [torch.from_numpy(b.view(np.int64)).to(dev) for b in batches]
            streams = [torch.cuda.Stream(device=dev) for _ in range(n_streams)]
            n_launch = 6 * n_streams
            d_picks = [torch.full((wl.R,), -7, dtype=torch.int32, device=dev) for _ in range(n_launch)]
            d_scores = [torch.empty(wl.R, dtype=torch.float64, device=dev) for _ in range(n_launch)]
            torch.cuda.synchronize()
            for i in range(n_launch):
                pk.pick_device(d_batches[i % len(batches)].data_ptr(), wl.R, None, d_picks[i].data_ptr(), d_scores[i].data_ptr(),
                               streams[i % n_streams].cuda_stream)
            torch.cuda.synchronize()
            for i in range(n_launch):
                op, os_ = want[i % len(batches)]
                assert_same(d_picks[i].cpu().numpy(), d_scores[i].cpu().numpy(), op, os_)
            ql, qd = pk.quad_stats()
            assert ql == (n_launch if n_streams <= 8 else 6 * 8) and qd > 0
            assert pk.launch_status() == 0


@pytest.mark.parametrize("R,P,density", [(1024, 4096, 0.5), (1000, 4096, 0.12), (640, 1000, 0.5), (512, 777, 0.3), (256, 2048, 0.5), (96, 64, 0.5)])
def test_candidate_masks(pkg, orc, R, P, density):
    """Masked batches on the quad route: random subsets, rows without any candidate, rows whose candidates miss the snapshot-wide QUEUE
    extremes (the request's own normalisers: every candidate in full, by the wavefront that found the row -- quad_exact_rows -- not
    deferred since round 6), rows whose listed pods are no candidates."""
    wl = pkg.workload.make_workload(5, R=R, P=P, n_groups=24, masked=True)
    rng = np.random.default_rng(R + P)
    W = (P + 63) // 64
    mask = wl.mask.copy()
    if density < 0.5:
        for _ in range(2):
            mask &= rng.integers(0, 2**63, (R, W), dtype=np.uint64) | (rng.integers(0, 2, (R, W), dtype=np.uint64) << np.uint64(63))
    mask[5] = 0                                              # no candidate at all: EPPK_NO_PICK
    mask[R // 2] = 0
    if P % 64:
        mask[:, -1] &= np.uint64((1 << (P % 64)) - 1)
    # a few rows with very few candidates (they almost surely miss a pod at the minimum / maximum queue depth)
    for r in range(7, R, 97):
        keep = rng.choice(P, size=3, replace=False)
        mask[r] = 0
        for p in keep:
            mask[r, p // 64] |= np.uint64(1) << np.uint64(p % 64)
    ql, qd = run(pkg, orc, wl, group_sets(wl), mask=mask)
    assert ql == 1
    assert qd <= R // 16, f"{qd} of {R} masked requests deferred"


@pytest.mark.parametrize("R,P", [(1024, 4096), (777, 3000), (500, 2048), (300, 1000), (128, 64), (61, 12)])
def test_subset_filters_that_leave_a_handful_of_endpoints(pkg, orc, R, P):
    """What the reference's subset filter produces (request.go:104-133: the endpoints named by the request's metadata): 1 .. 8 candidates
    per request.  Nearly every such request misses a pod at the snapshot-wide minimum or maximum queue depth, so EVERY row of a wavefront
    is scored in full with its own normalisers -- four rows side by side, each by its own 16 lanes (quad_exact_rows_par) -- and none is
    deferred.  Half of the requests keep one of the pods their prefix is cached on among the candidates (a listed candidate)."""
    wl = pkg.workload.make_workload(5, R=R, P=P, n_groups=24, masked=True)
    rng = np.random.default_rng(3 * R + P)
    W = (P + 63) // 64
    sets = group_sets(wl)
    mask = np.zeros((R, W), dtype=np.uint64)
    for r in range(R):
        n = int(rng.integers(1, min(8, P) + 1))
        keep = list(rng.choice(P, size=n, replace=False))
        pods = sets.get(int(wl.reqs[r, 1]), ())
        if r % 2 == 0 and pods:
            keep.append(pods[int(rng.integers(0, len(pods)))])
        for p_ in keep:
            mask[r, int(p_) // 64] |= np.uint64(1) << np.uint64(int(p_) % 64)
    mask[3] = 0                                              # no candidate at all: EPPK_NO_PICK
    ql, qd = run(pkg, orc, wl, sets, mask=mask)
    assert ql == 1
    assert qd <= R // 16, f"{qd} of {R} masked requests deferred"
    # ordered fallbacks within the subset (PickResult.Fallbacks, server.go:72-77): the same rows, k rounds each -- more rounds than
    # some requests have candidates (the lists end in EPPK_NO_PICK)
    calls = pairs_by_pod(sets)
    oix = orc.OracleIndex()
    for h, p_ in calls:
        oix.insert(h, p_)
    for k in (2, 5):
        want_p, want_s = orc.pick_topk(wl.chain, wl.pods, oix, wl.reqs, k, mask)
        for on in (True, False):
            with quad_env(on):
                with pkg.BatchedPicker(wl.chain, max_pods=max(P, 64), max_blocks=wl.B, max_batch=R, index_slots=max(wl.index_slots, 1024)) as pk:
                    pk.publish(wl.pods)
                    for h, p_ in calls:
                        pk.index_insert(h, p_)
                    got_p, got_s = pk.pick_topk(wl.reqs, k, mask)
                    ql, qd = pk.quad_stats()
            assert np.array_equal(got_p, want_p), f"k {k}, quad {on}: {np.count_nonzero(got_p != want_p)} entries differ, first rows {np.nonzero((got_p != want_p).any(axis=1))[0][:5]}"
            assert np.array_equal(got_s.view(np.uint64), want_s.view(np.uint64)), f"k {k}, quad {on}: scores differ"
            assert (got_p[3] == -1).all()
            if on:
                assert qd <= R // 16, f"k {k}: {qd} of {R} masked requests deferred"


@pytest.mark.parametrize("R,P,k", [(512, 4096, 2), (777, 4096, 8), (300, 1000, 3), (256, 64, 8), (128, 12, 8)])
def test_ordered_fallbacks(pkg, orc, R, P, k):
    """eppk_pick_topk on the quad route: the k best candidates in order (listed pods and top-table entries merged in the row); lists of
    up to 24 pods, tables that run out (few pods), requests with long matches."""
    wl = pkg.workload.make_workload(5, R=R, P=P, n_groups=12)
    sets = group_sets(wl)
    rng = np.random.default_rng(R * k)
    grown = {}
    for h, pods in list(sets.items()):                       # some groups cached on many pods (the second id of a lane comes into play)
        if pods not in grown:
            extra = rng.choice(P, size=min(P, 14), replace=False).tolist() if len(grown) % 3 == 0 else []
            grown[pods] = tuple(sorted(set(pods) | set(extra)))[:24]
        sets[h] = grown[pods]
    calls = pairs_by_pod(sets)
    oix = orc.OracleIndex()
    for h, p in calls:
        oix.insert(h, p)
    want_p, want_s = orc.pick_topk(wl.chain, wl.pods, oix, wl.reqs, k)
    with quad_env(True):
        with pkg.BatchedPicker(wl.chain, max_pods=max(P, 64), max_blocks=wl.B, max_batch=R, index_slots=max(wl.index_slots, 1024)) as pk:
            pk.publish(wl.pods)
            for h, p in calls:
                pk.index_insert(h, p)
            got_p, got_s = pk.pick_topk(wl.reqs, k)
            ql, qd = pk.quad_stats()
    assert np.array_equal(got_p, want_p), f"{np.count_nonzero(got_p != want_p)} entries differ"
    assert np.array_equal(got_s.view(np.uint64), want_s.view(np.uint64))
    assert ql == 1 and qd < R


def test_unordered_and_learned_lists_stay_on_the_quad_route(pkg, orc):
    """The library keeps every list in ascending order itself (index_canon_kernel behind every update launch), so equal pod SETS
    are equal list LINES whatever order the pairs arrived in: ONE insert call with the pairs shuffled, then two generations of a closed
    loop whose chain pushes the picks AWAY from the cached pods (negative prefix weight: every learn appends new pods to all 16 blocks
    of a group, in device order) -- the quad kernel still scores every request itself, bit-exact against the oracle."""
    rng = np.random.default_rng(99)
    wl = pkg.workload.make_workload(5, R=256, P=4096, n_groups=32)
    wl.chain = [(1, 2), (2, 2), (3, 1), (4, -3)]
    batches = [wl.reqs] + [pkg.workload.make_requests(wl, 500 + i) for i in range(2)]
    perm = rng.permutation(wl.index_hashes.shape[0])
    oix = orc.OracleIndex()
    oix.insert(wl.index_hashes, wl.index_pods)
    with quad_env(True):
        with pkg.BatchedPicker(wl.chain, max_pods=wl.P, max_blocks=wl.B, max_batch=wl.R, index_slots=1 << 17) as pk:
            pk.publish(wl.pods)
            pk.index_insert(wl.index_hashes[perm], wl.index_pods[perm])
            grew = 0
            for g, reqs in enumerate(batches):
                picks, scores = pk.pick(reqs)
                op, osc, _ = orc.pick_batch(wl.chain, wl.pods, oix, reqs, wl.B)
                assert_same(picks, scores, op, osc)
                hs = hashes_of(wl, reqs)
                before = oix.size()
                pk.index_insert(hs.reshape(-1), np.repeat(picks.astype(np.uint32), wl.B))
                oix.insert_picks(reqs, wl.B, op)
                grew += oix.size() - before
                assert pk.index_size() == oix.size() and pk.index_selfcheck() == 0, g
            ql, qd = pk.quad_stats()
            assert ql == len(batches) and qd == 0, f"{qd} requests left the quad route"
            assert grew > 0 and pk.launch_status() == 0


@pytest.mark.parametrize("R,P,k,density", [(1024, 4096, 4, 0.5), (600, 4096, 8, 0.25), (300, 1000, 3, 0.5), (256, 64, 8, 0.5), (640, 2048, 2, 0.5),
                                           (2048, 4096, 8, 0.12), (512, 96, 8, 0.25), (777, 48, 6, 0.25), (1500, 130, 8, 0.06)])
def test_ordered_fallbacks_with_candidate_masks(pkg, orc, R, P, k, density):
    """eppk_pick_topk with a candidate mask per request on the quad route (protocol: ordered fallbacks within the subset hint,
    docs/proposals/004-endpoint-picker-protocol/README.md:73, request.go:104-133): the merge of listed pods and table entries runs over
    the request's candidates only; rows without any candidate, rows with fewer than k candidates (padded with EPPK_NO_PICK), rows whose
    candidates miss the snapshot-wide QUEUE extremes (deferred to the masked work-list pass)."""
    wl = pkg.workload.make_workload(5, R=R, P=P, n_groups=12, masked=True)
    rng = np.random.default_rng(R * k + P)
    W = (P + 63) // 64
    mask = wl.mask.copy()
    d = 0.5
    while d > density * 1.01:                                # every AND with a random word halves the density
        mask &= rng.integers(0, 2**63, (R, W), dtype=np.uint64) | (rng.integers(0, 2, (R, W), dtype=np.uint64) << np.uint64(63))
        d *= 0.5
    mask[3] = 0                                              # no candidate at all
    for r in range(11, R, 53):                               # two or three candidates: fewer than k
        keep = rng.choice(P, size=min(P, 2 + (r & 1)), replace=False)
        mask[r] = 0
        for p in keep:
            mask[r, p // 64] |= np.uint64(1) << np.uint64(p % 64)
    if P % 64:
        mask[:, -1] &= np.uint64((1 << (P % 64)) - 1)
    sets = group_sets(wl)
    calls = pairs_by_pod(sets)
    oix = orc.OracleIndex()
    for h, p in calls:
        oix.insert(h, p)
    want_p, want_s = orc.pick_topk(wl.chain, wl.pods, oix, wl.reqs, k, mask)
    for on in (True, False):
        with quad_env(on):
            with pkg.BatchedPicker(wl.chain, max_pods=max(P, 64), max_blocks=wl.B, max_batch=R, index_slots=max(wl.index_slots, 1024)) as pk:
                pk.publish(wl.pods)
                for h, p in calls:
                    pk.index_insert(h, p)
                got_p, got_s = pk.pick_topk(wl.reqs, k, mask)
                ql, qd = pk.quad_stats()
        assert np.array_equal(got_p, want_p), f"quad {on}: {np.count_nonzero(got_p != want_p)} entries differ"
        assert np.array_equal(got_s.view(np.uint64), want_s.view(np.uint64))
        assert (got_p[3] == -1).all()
        assert (ql, qd < R) == ((1, True) if on else (0, True))


@pytest.mark.parametrize("P,B", [(4096, 32), (1000, 16), (2000, 8)])
def test_requests_that_come_back(pkg, orc, P, B):
    """A request that RETURNS after the index learned its pick: the tail blocks are listed on that one pod, the prefix blocks on the
    group's pods and that pod -- two pod sets, the second a single pod: since protocol v5 (round 6) pick_quad_kernel reads both from the
    set ids in the bucket lines and scores the request in place (rounds 2-5 deferred it to the work-list pass: 64 us per 64k returning
    batch against 23 for new requests; now 33) -- against the oracle over several generations of pick + LEARN on the SAME and on half-new
    batches, with candidate masks, as ordered fallbacks, and for the shapes next to it (fewer than four prefix hits, a tail pod outside
    the prefix list, tail blocks on two pods: three sets, deferred)."""
    with quad_env(True):
        import torch
        R = 1024
        wl = pkg.workload.make_workload(5, R=R, P=P, B=B, n_groups=12, masked=True)
        other = pkg.workload.make_requests(wl, 4242)
        half = wl.reqs.copy(); half[::2] = other[::2]
        dev = torch.device("cuda", 0)
        st = torch.cuda.Stream()
        d_pick = torch.empty(R * 4, dtype=torch.int32, device=dev); d_score = torch.empty(R * 4, dtype=torch.float64, device=dev)
        with pkg.BatchedPicker(wl.chain, max_pods=P, max_blocks=B, max_batch=R, index_slots=1 << 18) as pk:
            pk.publish(wl.pods)
            pk.index_insert(wl.index_hashes, wl.index_pods)
            oix = orc.OracleIndex()
            oix.insert(wl.index_hashes, wl.index_pods)
            q0 = pk.quad_stats()
            deferred = []
            for gen, reqs in enumerate([wl.reqs, wl.reqs, half, wl.reqs, half, other, wl.reqs]):
                d_reqs = torch.from_numpy(reqs.view(np.int64)).to(dev)
                l0, d0 = pk.quad_stats()
                pk.pick_learn_device(d_reqs.data_ptr(), R, None, d_pick.data_ptr(), d_score.data_ptr(), st.cuda_stream)
                torch.cuda.synchronize()
                l1, d1 = pk.quad_stats()
                deferred.append(d1 - d0)
                op, osc, _ = orc.pick_batch(wl.chain, wl.pods, oix, reqs, B)
                assert_same(d_pick[:R].cpu().numpy(), d_score[:R].cpu().numpy(), op, osc)
                oix.insert_picks(reqs, B, op)
                assert pk.index_size() == oix.size() and pk.index_selfcheck() == 0, gen
            assert deferred[0] == 0, deferred               # (generation 1 re-sends generation 0's batch: every request is a returning one ...
            assert deferred[1] <= R // 16, deferred         #  ... and the quad kernel scores it itself: prefix set + the one pod of its tail)
            # the same index through masks and ordered fallbacks (returning requests again)
            d_reqs = torch.from_numpy(wl.reqs.view(np.int64)).to(dev)
            d_mask = torch.from_numpy(wl.mask.view(np.int64)).to(dev)
            pk.pick_device(d_reqs.data_ptr(), R, d_mask.data_ptr(), d_pick.data_ptr(), d_score.data_ptr(), st.cuda_stream)
            torch.cuda.synchronize()
            op, osc, _ = orc.pick_batch(wl.chain, wl.pods, oix, wl.reqs, B, wl.mask)
            assert_same(d_pick[:R].cpu().numpy(), d_score[:R].cpu().numpy(), op, osc)
            for mask, dm in ((None, None), (wl.mask, d_mask.data_ptr())):
                pk._check(pk._lib.eppk_pick_topk_device(pk._ctx, d_reqs.data_ptr(), R, dm, 4, d_pick.data_ptr(), d_score.data_ptr(), st.cuda_stream), "topk")
                torch.cuda.synchronize()
                tp, ts = orc.pick_topk_batch(wl.chain, wl.pods, oix, wl.reqs, B, 4, mask)
                assert np.array_equal(d_pick.cpu().numpy().reshape(R, 4), tp)
                assert np.array_equal(d_score.cpu().numpy().view(np.uint64).reshape(R, 4), ts.view(np.uint64))
            # neighbouring shapes (hand-made): tail pod outside A; tail on two pods; fewer than four prefix hits; one-pod lists throughout
            odd = wl.reqs[:64].copy()
            hs = odd[:, 1:1 + B]
            extra_h, extra_p = [], []
            for r in range(64):
                tail = hs[r, B // 2:]
                if r % 4 == 0:                       # a second pod on the tail blocks
                    extra_h += list(tail); extra_p += [(7 * r + 3) % P] * tail.size
                elif r % 4 == 1:                     # ... on the LAST block only (three lists)
                    extra_h.append(tail[-1]); extra_p.append((5 * r + 1) % P)
            pk.index_insert(np.array(extra_h, dtype=np.uint64), np.array(extra_p, dtype=np.uint32))
            oix.insert(np.array(extra_h, dtype=np.uint64), np.array(extra_p, dtype=np.uint32))
            short = odd.copy()
            short[:, 1:1 + B] = np.roll(hs, -(B // 2 - 2), axis=1)   # two prefix hits, then the tail, then misses
            for reqs in (odd, short):
                picks, scores = pk.pick(reqs)
                op, osc, _ = orc.pick_batch(wl.chain, wl.pods, oix, reqs, B)
                assert_same(picks, scores, op, osc)
            assert pk.launch_status() == 0


def test_two_pod_sets_per_request_are_scored_in_place(pkg, orc):
    """Protocol v5 (round 6): pick_quad_kernel reads the identity of a hit's pod set from its bucket line.  A request whose hits name TWO
    sets, at most one of them a list and the other a single pod, is scored in place -- matched[p] = cA [p in A] + cb [p == b] -- in every
    arrangement: list then pod (a returning request), pod then list (hit 0 names the single pod), pod then another pod, the single pod on
    the list or not, the single pod beyond the snapshot's pods; three sets, two lists and a dense set among the hits are deferred.  Every
    variant against the oracle, masked as well, with the route on and off (run())."""
    wl = pkg.workload.make_workload(5, R=320, P=4096, n_groups=8)
    base = group_sets(wl)
    hs = hashes_of(wl)
    sets = dict(base)
    rng = np.random.default_rng(5)
    kinds = 10
    expect_deferred = 0
    for r in range(wl.R):
        A = base[int(hs[r, 0])]                              # the group's pods (blocks 0..15 of the row)
        kind = r % kinds
        tail = [int(h) for h in hs[r, 16:]]                  # the row's own 16 blocks: nobody else asks for them
        on = int(A[r % len(A)]); off = int((A[0] + 1 + r) % wl.P)
        while off in A:
            off = (off + 1) % wl.P
        if kind == 0:                                        # returning: tail on one pod of the list
            for h in tail: sets[h] = (on,)
        elif kind == 1:                                      # ... on a pod that is NOT on the list
            for h in tail: sets[h] = (off,)
        elif kind == 2:                                      # only 5 tail blocks learned (m = 21: steps 5..7 on demand)
            for h in tail[:5]: sets[h] = (on,)
        elif kind == 3:                                      # tail on TWO different single pods: three sets -> deferred
            for h in tail[:8]: sets[h] = (on,)
            for h in tail[8:]: sets[h] = (off,)
            expect_deferred += 1
        elif kind == 4:                                      # tail on a second LIST: two lists -> deferred
            for h in tail: sets[h] = tuple(sorted({on, off}))
            expect_deferred += 1
        elif kind == 5:                                      # tail on a dense set (more than 24 pods): no id -> deferred
            big = tuple(sorted(set(rng.choice(wl.P, 30, replace=False).tolist())))
            for h in tail[:4]: sets[h] = big
            expect_deferred += 1
        # kinds 6..9: new requests (one set), the common shape
    # rows whose FIRST hit names a single pod and the rest a list / another pod: private copies of the shared blocks
    reqs = wl.reqs.copy()
    for r in range(0, wl.R, 16):
        priv = rng.integers(1, 2**63, 6, dtype=np.uint64)
        reqs[r, 1:7] = priv
        reqs[r, 7:] ^= np.uint64(0x5DEECE66D)                # the rest of the row: misses
        A = base[int(hs[r, 0])]
        sets[int(priv[0])] = (int(A[1]),) if (r // 16) % 2 == 0 else ((int(A[0]) + 3) % wl.P,)     # the single pod: on the list / not on it
        for h in priv[1:4]: sets[int(h)] = A                 # then the list
        for h in priv[4:]: sets[int(h)] = A
    for r in range(8, wl.R, 32):                             # pod then ANOTHER pod
        priv = rng.integers(1, 2**63, 5, dtype=np.uint64)
        reqs[r, 1:6] = priv
        reqs[r, 6:] ^= np.uint64(0x5DEECE66D)
        sets[int(priv[0])] = (11,)
        for h in priv[1:]: sets[int(h)] = (4000,)
    ql, qd = run(pkg, orc, wl, sets, reqs=reqs)
    assert ql == 1
    # exactly the three-set / two-list / dense rows are deferred (rows rewritten above may have replaced a few of them)
    assert 0 < qd <= expect_deferred, (qd, expect_deferred)
    mask = pkg.workload.make_workload(5, R=wl.R, P=wl.P, n_groups=8, masked=True).mask
    run(pkg, orc, wl, sets, reqs=reqs, mask=mask)
