"""GPU, BASELINE.json's full size (64k requests x 4096 pods, full scorer chain + prefix index): the pick variants whose kernels only
reach their steady state at that size -- a wavefront of pick_quad_kernel walks FOUR blocks of four requests per 64k batch (rows two
blocks ahead, key buckets one ahead), and until round 4 ordered fallbacks, masked fallbacks and the random-top-k picker were compared
with the oracle at one block per wavefront at most.  Every case against the threaded oracle (orc_pick_batch_mt / orc_pick_topk), picks
and scores bit for bit, in the one-launch form (every workgroup scores what it deferred itself) and in the two-launch form
(EPPK_QUAD_TAIL=0: pick_quad_kernel + the work-list pass of pick_fast_kernel):
  * ordered fallbacks k = 2 / 4 / 8                         PickResult.Fallbacks, pkg/lwepp/handlers/server.go:72-77
  * candidate masks at 50 % and 12.5 % density              the subset filter as a bitmask, pkg/lwepp/handlers/request.go:104-133
  * masked fallbacks k = 4, random-top-3                    docs/proposals/0845-…/examples/example.yaml:25
  * the pipelined staging sets with EPPK_PICK_LEARN, four generations: pick -> the index learns the picks -> next batch
                                                            docs/proposals/0602-…/README.md:101-108
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

R, P = 65536, 4096
FORMS = {"one-launch": {}, "two-launch": {"EPPK_QUAD_TAIL": "0"}}


def _splitmix(z):
    z = (z + np.uint64(0x9E3779B97F4A7C15))
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def _fmix(z):
    """SEMANTICS.md 3b's splitmix64(z): the finaliser alone (the caller has added the increment)."""
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


class Full:
    """The C5 workload at full size, its index in the oracle, masks of two densities, and a cache of oracle answers (a case runs in
    two library forms against the same answer)."""

    def __init__(self, pkg, orc):
        self.pkg, self.orc = pkg, orc
        self.cores = os.cpu_count() or 1
        self.wl = wl = pkg.workload.make_workload(5, masked=True)
        assert wl.R == R and wl.P == P and wl.B == 32
        self.oix = orc.OracleIndex()
        self.oix.insert(wl.index_hashes, wl.index_pods)
        W = P // 64
        with np.errstate(over="ignore"):
            a = _splitmix(np.arange(R * W, dtype=np.uint64) + np.uint64(0xA11CE)).reshape(R, W)
            b = _splitmix(np.arange(R * W, dtype=np.uint64) + np.uint64(0xB0B0000)).reshape(R, W)
        self.masks = {"50": wl.mask, "12": (wl.mask & a & b)}        # ~2048 / ~512 candidates per request
        # what a subset filter leaves (request.go:104-133): 1 .. 8 endpoints per request -- nearly every request needs its own QUEUE normalisers
        sub = np.zeros((R, W), dtype=np.uint64)
        rng = np.random.default_rng(0x5B5E7)
        pods = rng.integers(0, P, (R, 8))
        keep = rng.integers(1, 9, R)
        for j in range(8):
            on = j < keep
            np.bitwise_or.at(sub, (np.arange(R)[on], pods[on, j] // 64), np.uint64(1) << (pods[on, j] % 64).astype(np.uint64))
        sub[12345] = 0
        self.masks["subset8"] = sub
        self.cache = {}

    def oracle_pick(self, mask_key):
        key = ("pick", mask_key)
        if key not in self.cache:
            wl = self.wl
            p, s, _ = self.orc.pick_batch(wl.chain, wl.pods, self.oix, wl.reqs, wl.B, self.masks.get(mask_key), threads=self.cores)
            self.cache[key] = (p, s)
        return self.cache[key]

    def oracle_topk(self, k, mask_key):
        key = ("topk", k, mask_key)
        if key not in self.cache:
            wl = self.wl
            self.cache[key] = self.orc.pick_topk_batch(wl.chain, wl.pods, self.oix, wl.reqs, wl.B, k, self.masks.get(mask_key), threads=self.cores)
        return self.cache[key]

    def picker(self):
        wl = self.wl
        pk = self.pkg.BatchedPicker(wl.chain, max_pods=P, max_blocks=wl.B, max_batch=R, index_slots=wl.index_slots)
        pk.publish(wl.pods)
        pk.index_insert(wl.index_hashes, wl.index_pods)
        return pk


@pytest.fixture(scope="module")
def full(pkg, orc):
    return Full(pkg, orc)


@pytest.fixture(params=list(FORMS))
def form(request, monkeypatch):
    for k, v in FORMS[request.param].items():
        monkeypatch.setenv(k, v)
    return request.param


def _same(got, want, what):
    gp, gs = got
    wp, ws = want
    bad = np.nonzero((gp != wp).reshape(gp.shape[0], -1).any(axis=1))[0]
    assert bad.size == 0, f"{what}: {bad.size} requests differ, first {bad[:4]}: gpu {gp[bad[:4]]} oracle {wp[bad[:4]]}"
    assert np.array_equal(gs.view(np.uint64), ws.view(np.uint64)), f"{what}: scores differ bitwise"


def _took_quad(pk):
    launches, deferred = pk.quad_stats()
    assert launches >= 1, "the batch did not take pick_quad_kernel"
    return deferred


@pytest.mark.parametrize("k", [2, 4, 8])
def test_ordered_fallbacks_at_full_size(full, form, k):
    with full.picker() as pk:
        got = pk.pick_topk(full.wl.reqs, k)
        _same(got, full.oracle_topk(k, None), f"top-{k}, {form}")
        assert np.array_equal(got[0][:, 0], full.oracle_pick(None)[0])           # entry 0 is the pick
        _took_quad(pk)
        assert pk.launch_status() == 0


@pytest.mark.parametrize("density", ["50", "12"])
def test_masked_picks_at_full_size(full, form, density):
    with full.picker() as pk:
        got = pk.pick(full.wl.reqs, full.masks[density])
        _same(got, full.oracle_pick(density), f"masked pick ({density} %), {form}")
        assert pk.launch_status() == 0
        # the same batch again (work-list buffers, counters and reports of the first launch are reused)
        _same(pk.pick(full.wl.reqs, full.masks[density]), full.oracle_pick(density), f"masked pick ({density} %), {form}, second launch")


def test_subset_filters_at_full_size(full, form):
    """64k requests, each with 1 .. 8 candidate endpoints: single picks and ordered fallbacks (more rounds than some requests have
    candidates); on the quad route every row is parked and scored by its own wavefront -- nothing is deferred."""
    with full.picker() as pk:
        got = pk.pick(full.wl.reqs, full.masks["subset8"])
        _same(got, full.oracle_pick("subset8"), f"subset filter, pick, {form}")
        assert got[0][12345] == -1
        assert _took_quad(pk) <= R // 64
        gk = pk.pick_topk(full.wl.reqs, 4, full.masks["subset8"])
        _same(gk, full.oracle_topk(4, "subset8"), f"subset filter, top-4, {form}")
        assert pk.launch_status() == 0


def test_masked_fallbacks_at_full_size(full, form):
    with full.picker() as pk:
        _same(pk.pick_topk(full.wl.reqs, 4, full.masks["50"]), full.oracle_topk(4, "50"), f"masked top-4, {form}")
        _took_quad(pk)
        assert pk.launch_status() == 0


def test_random_top3_at_full_size(full, form, orc):
    """SEMANTICS.md 3b on the threaded fallback lists (numpy restatement of the selection rule), and the oracle's own
    orc_pick_random_topk on the first 2048 requests."""
    seed = 0xC0FFEE
    wl = full.wl
    lists, totals = full.oracle_topk(3, None)
    n = (lists >= 0).sum(axis=1).astype(np.uint64)
    with np.errstate(over="ignore"):
        u = _fmix(np.uint64(seed) + (np.arange(R, dtype=np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15))
    sel = (u % np.maximum(n, np.uint64(1))).astype(np.int64)
    want_p = np.where(n > 0, lists[np.arange(R), sel], -1).astype(np.int32)
    want_s = np.where(n > 0, totals[np.arange(R), sel], 0.0)
    op, osc = orc.pick_random_topk(wl.chain, wl.pods, full.oix, wl.reqs[:2048], wl.B, 3, seed)
    assert np.array_equal(op, want_p[:2048]) and np.array_equal(osc.view(np.uint64), want_s[:2048].view(np.uint64))
    with full.picker() as pk:
        _same(pk.pick_random_topk(wl.reqs, 3, seed), (want_p, want_s), f"random-top-3, {form}")
        assert pk.launch_status() == 0


def test_unmasked_pick_defers_nothing_and_matches(full, form):
    """The headline launch itself in both forms: nothing deferred, picks and scores equal."""
    with full.picker() as pk:
        _same(pk.pick(full.wl.reqs), full.oracle_pick(None), f"pick, {form}")
        assert _took_quad(pk) == 0


def test_pipelined_learn_at_full_size(pkg, orc, full):
    """eppk_pick_stage_begin(EPPK_PICK_LEARN) / _end over two staging sets, 64k requests per batch, four generations: batch k + 1 is
    uploaded while batch k is scored and its post-route update runs, and must see what batch k taught the index."""
    wl = full.wl
    cores = os.cpu_count() or 1
    batches = [wl.reqs] + [pkg.workload.make_requests(wl, 7700 + i) for i in range(3)]
    with pkg.BatchedPicker(wl.chain, max_pods=P, max_blocks=wl.B, max_batch=R, index_slots=1 << 24) as pk:
        pk.publish(wl.pods)
        pk.index_insert(wl.index_hashes, wl.index_pods)
        oix = orc.OracleIndex()
        oix.insert(wl.index_hashes, wl.index_pods)
        bufs = [pk.stage_buffers(0)[0], pk.stage_buffers(1)[0]]

        def check(s, b):
            picks, scores = pk.stage_end(s)
            op, osc, _ = orc.pick_batch(wl.chain, wl.pods, oix, batches[b], wl.B, threads=cores)
            _same((picks, scores), (op, osc), f"generation {b}")
            oix.insert_picks(batches[b], wl.B, op)

        bufs[0][:R] = batches[0]
        pk.stage_begin(0, R, learn=True)
        for b in range(1, 4):
            bufs[b & 1][:R] = batches[b]
            pk.stage_begin(b & 1, R, learn=True)
            check((b & 1) ^ 1, b - 1)
        check(1, 3)
        # the update of the last batch may still be running: every index entry point orders itself behind it
        assert pk.index_size() == oix.size()
        assert pk.index_selfcheck() == 0 and pk.index_dropped() == 0 and pk.launch_status() == 0


def test_index_maintenance_right_behind_a_learn_batch(pkg, orc):
    """Round-3 advisor: eppk_pick_stage_end returns when the PICKS are there; the LEARN update (three kernels on the set's private stream)
    may still run.  An eviction, a removal, an insert or a publish issued at once on the context's stream must queue behind it."""
    Rs = 32768
    wl = pkg.workload.make_workload(5, R=Rs)
    cores = os.cpu_count() or 1
    with pkg.BatchedPicker(wl.chain, max_pods=P, max_blocks=wl.B, max_batch=Rs, index_slots=1 << 23) as pk:
        pk.publish(wl.pods)
        pk.index_insert(wl.index_hashes, wl.index_pods)
        oix = orc.OracleIndex()
        oix.insert(wl.index_hashes, wl.index_pods)
        buf = pk.stage_buffers(0)[0]
        for gen in range(3):
            reqs = wl.reqs if gen == 0 else pkg.workload.make_requests(wl, 600 + gen)
            buf[:Rs] = reqs
            pk.stage_begin(0, Rs, learn=True)
            picks, scores = pk.stage_end(0)
            # ... and immediately, without any synchronisation of the caller's:
            e = pk.index_advance_epoch()
            n_ev = pk.index_evict_older(e)                     # everything stamped before this epoch -- the batch just learned included
            op, osc, _ = orc.pick_batch(wl.chain, wl.pods, oix, reqs, wl.B, threads=cores)
            _same((picks, scores), (op, osc), f"generation {gen}")
            oix.insert_picks(reqs, wl.B, op)
            assert oix.advance_epoch() == e
            assert n_ev == oix.evict_older(e)
            pod = int(op[0]) if op[0] >= 0 else 0
            pk.index_remove_pod(pod)
            oix.remove_pod(pod)
            pk.index_insert(wl.index_hashes, wl.index_pods)
            oix.insert(wl.index_hashes, wl.index_pods)
            assert pk.index_selfcheck() == 0
            assert pk.index_size() == oix.size()
        assert pk.launch_status() == 0 and pk.index_dropped() == 0


@pytest.mark.parametrize("Rs,keep", [(16384, 0), (16384, 1), (512, 1)])
def test_ageing_while_staging_sets_are_in_flight(pkg, orc, Rs, keep):
    """The ageing step of a router's closed loop -- eppk_index_advance_epoch + eppk_index_evict_older_device -- issued BETWEEN two begins
    of the pipelined LEARN path, a set in flight each time (include/eppk.h: the one index entry point allowed there): the eviction must
    queue behind the pick and the update of every set begun before it and ahead of the pick of every set begun after it, i.e. the
    index ages in the order of the calls.  The oracle replays exactly that order.  (16k requests: upload + pick_quad_kernel<LEARN>;
    512: the zero-copy form.)"""
    wl = pkg.workload.make_workload(5, R=Rs)
    cores = os.cpu_count() or 1
    n_batches = 9
    batches = [wl.reqs] + [pkg.workload.make_requests(wl, 8800 + i) for i in range(1, 4)]
    with pkg.BatchedPicker(wl.chain, max_pods=P, max_blocks=wl.B, max_batch=Rs, index_slots=1 << 23) as pk:
        pk.publish(wl.pods)
        pk.index_insert(wl.index_hashes, wl.index_pods)
        oix = orc.OracleIndex()
        oix.insert(wl.index_hashes, wl.index_pods)
        bufs = [pk.stage_buffers(0)[0], pk.stage_buffers(1)[0]]
        expect = {}

        def begin(b):
            reqs = batches[b % len(batches)]
            bufs[b & 1][:Rs] = reqs
            pk.stage_begin(b & 1, Rs, learn=True)
            op, osc, _ = orc.pick_batch(wl.chain, wl.pods, oix, reqs, wl.B, threads=cores)      # the oracle, in the order of the calls
            oix.insert_picks(reqs, wl.B, op)
            expect[b] = (op, osc)

        def tick():
            e = pk.index_advance_epoch()
            assert oix.advance_epoch() == e
            if e > keep:
                pk.index_evict_older_device(e - keep)          # no count, no wait: set (b & 1) is between begin and end
                oix.evict_older(e - keep)

        begin(0)
        for b in range(1, n_batches):
            begin(b)
            if b % 2 == 0:
                tick()
            _same(pk.stage_end((b - 1) & 1), expect.pop(b - 1), f"batch {b - 1}")
        tick()                                                 # ... and once with the last set still in flight
        _same(pk.stage_end((n_batches - 1) & 1), expect.pop(n_batches - 1), f"batch {n_batches - 1}")
        assert pk.index_size() == oix.size()
        assert pk.index_selfcheck() == 0 and pk.index_dropped() == 0 and pk.launch_status() == 0
