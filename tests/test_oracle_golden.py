"""CPU: pin the oracle (oracle/oracle.c) against the committed golden vectors (tests/golden/cases.npz).

The reference holds no golden vectors for the scorer chain (parity UNPINNED); the
vectors come from an independent numpy restatement (tests/golden/gen_golden.py) and python-xxhash.
"""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cases.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def case_names(gold):
    return sorted({k.split("/")[0] for k in gold.files if k.endswith("/pick")})


def load_case(gold, name):
    return {k.split("/", 1)[1]: gold[k] for k in gold.files if k.startswith(name + "/")}


def rows_of(pkg, c):
    B = int(c["B"])
    return pkg.picker.make_req_rows(c["adapter"], c["n_blocks"], c["hashes"][:, :B] if B else None, B), B


def test_golden_file_has_cases(gold):
    assert len(case_names(gold)) >= 10


def test_oracle_matches_numpy_restatement_bitwise(gold, pkg, orc):
    for name in case_names(gold):
        c = load_case(gold, name)
        reqs, B = rows_of(pkg, c)
        oix = orc.OracleIndex()
        oix.insert(c["index_hashes"], c["index_pods"])
        for p in c["removed"].tolist():
            oix.remove_pod(int(p))
        chain = [(int(k), int(w)) for k, w in c["chain"]]
        mask = c["mask"] if c["mask"].size else None
        picks, scores, _ = orc.pick_batch(chain, c["pods"], oix, reqs, B, mask)
        assert np.array_equal(picks, c["pick"]), name
        assert np.array_equal(scores.view(np.uint64), c["score"].view(np.uint64)), name


def test_sparse_cpu_algorithm_equals_the_per_request_loop(pkg, orc):
    """orc_pick_batch_sparse (snapshot tables + the pods each prefix walk names) against orc_pick_batch on the bench workloads'
    shapes (C1..C3 here; C5 in bench.py's `parity` of every run): picks equal, scores bitwise equal, tables reused across batches,
    holes and an empty snapshot."""
    for cfg in (1, 2, 3):
        wl = pkg.workload.make_workload(cfg)
        oix = orc.OracleIndex()
        oix.insert(wl.index_hashes, wl.index_pods)
        tb = orc.OracleTables(wl.chain, wl.pods)
        for b in range(2):
            reqs = wl.reqs if b == 0 else pkg.workload.make_requests(wl, 1234 + cfg)
            n = min(reqs.shape[0], 2048)
            p0, s0, _ = orc.pick_batch(wl.chain, wl.pods, oix, reqs[:n], wl.B, threads=4)
            p1, s1 = tb.pick_batch(oix, reqs[:n], wl.B, threads=4)
            assert np.array_equal(p0, p1), (cfg, b)
            assert np.array_equal(s0.view(np.uint64), s1.view(np.uint64)), (cfg, b)
    pods = wl.pods.copy()
    pods["flags"][::3] = 1                      # EPPK_POD_INACTIVE
    oix.scrub_inactive(pods)
    p0, s0, _ = orc.pick_batch(wl.chain, pods, oix, wl.reqs[:512], wl.B)
    p1, s1 = orc.OracleTables(wl.chain, pods).pick_batch(oix, wl.reqs[:512], wl.B, threads=2)
    assert np.array_equal(p0, p1) and np.array_equal(s0.view(np.uint64), s1.view(np.uint64))
    pods["flags"][:] = 1
    p1, s1 = orc.OracleTables(wl.chain, pods).pick_batch(oix, wl.reqs[:64], wl.B)
    assert (p1 == -1).all() and (s1 == 0.0).all()


@pytest.mark.parametrize("cfg,masked", [(2, False), (3, False), (3, True), (5, True)])
def test_batch_fallback_lists_equal_the_per_request_formulation(pkg, orc, cfg, masked):
    """orc_pick_topk (C, threaded: what can check the GPU's fallback lists at full batch sizes) against binding.pick_topk (a lexsort
    over orc_score_row's totals, one request at a time): lists and scores bitwise equal for k = 1 / 3 / 8, with candidate masks that
    leave some requests fewer than k candidates or none, with holes in the snapshot; column 0 is the pick of orc_pick_batch."""
    wl = pkg.workload.make_workload(cfg, R=300, P=777 if cfg != 5 else 4096, masked=masked)
    mask = wl.mask
    if masked:
        mask = mask.copy()
        mask[0, :] = 0                                        # no candidate
        mask[1, :] = 0; mask[1, 0] = np.uint64(0b101)         # two candidates
    pods = wl.pods.copy()
    pods["flags"][5::7] = 1                                   # EPPK_POD_INACTIVE: holes
    oix = orc.OracleIndex()
    if wl.index_slots:
        oix.insert(wl.index_hashes, wl.index_pods)
        oix.scrub_inactive(pods)
    p1, s1, _ = orc.pick_batch(wl.chain, pods, oix, wl.reqs, wl.B, mask)
    for k in (1, 3, 8):
        a_p, a_s = orc.pick_topk(wl.chain, pods, oix, wl.reqs, k, mask)
        for threads in (1, 5):
            b_p, b_s = orc.pick_topk_batch(wl.chain, pods, oix, wl.reqs, wl.B, k, mask, threads=threads)
            assert np.array_equal(a_p, b_p), (k, threads, np.nonzero((a_p != b_p).any(axis=1))[0][:5])
            assert np.array_equal(a_s.view(np.uint64), b_s.view(np.uint64)), (k, threads)
        assert np.array_equal(b_p[:, 0], p1) and np.array_equal(b_s[:, 0].view(np.uint64), s1.view(np.uint64))
        if masked:
            assert (b_p[0] == -1).all() and (b_s[0] == 0.0).all() and (b_p[1, 2:] == -1).all()


def test_oracle_mt_equals_sequential(gold, pkg, orc):
    c = load_case(gold, "full_masked")
    reqs, B = rows_of(pkg, c)
    oix = orc.OracleIndex()
    oix.insert(c["index_hashes"], c["index_pods"])
    chain = [(int(k), int(w)) for k, w in c["chain"]]
    a = orc.pick_batch(chain, c["pods"], oix, reqs, B, c["mask"])
    b = orc.pick_batch(chain, c["pods"], oix, reqs, B, c["mask"], threads=3)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint64), b[1].view(np.uint64))


def test_xxh64_known_answers_oracle_and_library(gold, pkg, orc):
    lib = pkg.load_library()
    data = gold["xxh64/data"].tobytes()
    off = 0
    for n, e0, e2a in zip(gold["xxh64/lens"].tolist(), gold["xxh64/seed0"].tolist(), gold["xxh64/seed_2a"].tolist()):
        m = data[off:off + n]
        off += n
        assert orc.xxh64(m, 0) == e0 and orc.xxh64(m, 0x2A) == e2a
        assert lib.eppk_xxh64(m, len(m), 0) == e0 and lib.eppk_xxh64(m, len(m), 0x2A) == e2a
    # the three values SURVEY.md §8c lists (seed 0)
    assert orc.xxh64(b"") == 0xEF46DB3751D8E999
    assert orc.xxh64(b"a") == 0xD24EC4F1A98C6E5B
    assert orc.xxh64(b"abc") == 0x44BC2CF5AD770999


def test_xxh64_against_python_xxhash_random():
    xxhash = pytest.importorskip("xxhash")
    import __graft_entry__ as g
    orc = g.load_oracle()
    lib = g.load_package().load_library()
    rng = np.random.default_rng(1)
    for n in list(range(0, 70)) + [127, 128, 129, 1000, 4097]:
        m = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        seed = int(rng.integers(0, 2**63))
        want = xxhash.xxh64(m, seed=seed).intdigest()
        assert orc.xxh64(m, seed) == want
        assert lib.eppk_xxh64(m, n, seed) == want


def test_chain_hash_known_answer(gold, pkg, orc):
    prompt = gold["chain/prompt"].tobytes()
    want = gold["chain/expected"]
    assert np.array_equal(orc.hash_prompt(b"m", prompt, 64, 8), want)          # only 3 full blocks in 200 bytes
    assert np.array_equal(pkg.picker.hash_prompt(b"m", prompt, 64, 8), want)
    assert pkg.picker.hash_prompt(b"m", prompt, 64, 2).shape[0] == 2          # max_out bounds the chain
    assert pkg.picker.hash_prompt(b"m", prompt[:63], 64, 8).shape[0] == 0       # no full block
    # a different model name changes every hash (LoRA-aware chains, 0602-…/README.md:121)
    assert not np.any(pkg.picker.hash_prompt(b"n", prompt, 64, 8) == want)


def test_hand_computed_scores(pkg, orc):
    """Small cases computed by hand from SEMANTICS.md."""
    Q, KV, L, PF = 1, 2, 3, 4
    pods = np.zeros(3, dtype=pkg.picker.POD_DTYPE)
    pods["queue"] = [0, 5, 10]
    pods["kv_util"] = [0.5, 0.25, 1.0]
    reqs = pkg.picker.make_req_rows(np.array([-1]), np.array([0]), None, 0)
    # queue scores [1, .5, 0], kv scores [.5, .75, 0]; totals 2*q + 2*kv = [3, 2.5, 0]
    picks, scores, _ = orc.pick_batch([(Q, 2), (KV, 2)], pods, None, reqs, 0)
    assert picks[0] == 0 and scores[0] == 3.0
    assert np.array_equal(orc.score_row([(Q, 2), (KV, 2)], pods, None, reqs[0]), [3.0, 2.5, 0.0])
    # all queues equal -> queue score 1.0 everywhere; tie on total broken by lowest index
    pods["queue"] = 7
    pods["kv_util"] = 0.5
    picks, scores, _ = orc.pick_batch([(Q, 1), (KV, 1)], pods, None, reqs, 0)
    assert picks[0] == 0 and scores[0] == 1.5
    # LoRA tiers: pod0 active, pod1 free slot, pod2 full + waiting, request for adapter 5
    pods["max_lora"] = [1, 4, 1]
    pods["active"][0, 0] = 1 << 5
    pods["active"][2, 0] = 1 << 9
    pods["waiting"][2, 0] = 1 << 5
    r5 = pkg.picker.make_req_rows(np.array([5]), np.array([0]), None, 0)
    assert np.array_equal(orc.score_row([(L, 10)], pods, None, r5[0]), [10.0, 8.0, 6.0])
    # base-model request: in neither set -> free slot 0.8, else 0.0
    assert np.array_equal(orc.score_row([(L, 10)], pods, None, reqs[0]), [0.0, 8.0, 0.0])
    # prefix: 4 blocks; pod0 holds blocks 0,1,2 ; pod1 holds 0 ; pod2 holds 0,1,3 (block 3 never reached: block 2 set = {0})
    oix = orc.OracleIndex()
    oix.insert([11, 12, 13, 11, 11, 12, 14], [0, 0, 0, 1, 2, 2, 2])
    rp = pkg.picker.make_req_rows(np.array([-1]), np.array([4]), np.array([[11, 12, 13, 99]], dtype=np.uint64), 4)
    assert np.array_equal(orc.score_row([(PF, 4)], pods, oix, rp[0]), [3.0, 1.0, 2.0])
    # interior gap: pod2 misses block 2 but the walk continues while ANY pod has the block; 14 at position 3
    rp2 = pkg.picker.make_req_rows(np.array([-1]), np.array([4]), np.array([[11, 12, 13, 14]], dtype=np.uint64), 4)
    assert np.array_equal(orc.score_row([(PF, 4)], pods, oix, rp2[0]), [3.0, 1.0, 3.0])
    # clamp: kv_util > 1 -> score < 0 -> 0 ; kv_util < 0 -> score > 1 -> 1 ; NaN -> 0
    pods["kv_util"] = [1.5, -0.5, np.nan]
    assert np.array_equal(orc.score_row([(KV, 3)], pods, None, reqs[0]), [0.0, 3.0, 0.0])


def test_index_semantics(orc):
    ix = orc.OracleIndex()
    ix.insert([5, 5, 5, 7], [3, 1, 3, 2])          # set semantics, sorted
    assert ix.lookup(5).tolist() == [1, 3] and ix.lookup(7).tolist() == [2] and ix.lookup(9).size == 0
    assert ix.size() == 2
    ix.remove_pod(2)
    assert ix.lookup(7).size == 0 and ix.size() == 1   # empty set behaves as absent


def test_golden_vectors_regenerate_identically(tmp_path):
    """The committed fixture is exactly what the committed script produces."""
    pytest.importorskip("xxhash")
    import importlib.util
    import shutil
    src = os.path.join(os.path.dirname(GOLD), "gen_golden.py")
    dst = tmp_path / "gen_golden.py"
    shutil.copy(src, dst)
    spec = importlib.util.spec_from_file_location("gen_golden_tmp", dst)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.main()
    a, b = np.load(GOLD), np.load(tmp_path / "cases.npz")
    assert sorted(a.files) == sorted(b.files)
    for k in a.files:
        x, y = a[k], b[k]
        assert x.dtype == y.dtype and x.shape == y.shape and x.tobytes() == y.tobytes(), k


@pytest.mark.parametrize("seed", range(40))
def test_oracle_vs_numpy_restatement_random(pkg, orc, seed):
    """Differential test of the two independent CPU restatements (oracle/oracle.c: per-request loops; tests/golden/gen_golden.py:
    whole-matrix numpy) on fresh random cases: random chains (any order, duplicates, negative weights), masks, tie-heavy gauges,
    removed pods, 0..12 blocks.  Picks identical, scores bitwise identical."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_golden", os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gen_golden.py"))
    gg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gg)
    rng = np.random.Generator(np.random.PCG64(900000 + seed))
    P = int(rng.choice([1, 2, 17, 63, 64, 65, 130, 200]))
    B = int(rng.choice([0, 1, 3, 8, 12]))
    kinds = [gg.Q, gg.KV, gg.L] + ([gg.PF] if B else [])
    chain = [(int(rng.choice(kinds)), int(rng.integers(-3, 6))) for _ in range(int(rng.integers(1, 7)))]
    removed = tuple(int(x) for x in rng.choice(P, size=int(rng.integers(0, min(P, 4))), replace=False)) if rng.random() < 0.3 else ()
    holes = tuple(int(x) for x in rng.choice(P, size=int(rng.integers(1, P + 1)), replace=False)) if seed % 3 == 2 else ()   # (sometimes ALL pods)
    c = gg.rand_case(rng, 32, P, B, chain, masked=bool(rng.random() < 0.5), n_groups=int(rng.integers(1, 5)),
                     tie_heavy=bool(rng.random() < 0.4), removed=removed, holes=holes)
    hashes = c["hashes"][:, :B] if B else None
    reqs = pkg.picker.make_req_rows(c["adapter"], c["n_blocks"], hashes, B)
    oix = orc.OracleIndex()
    oix.insert(c["index_hashes"], c["index_pods"])
    for p in removed:
        oix.remove_pod(p)
    oix.scrub_inactive(c["pods"])                      # the index side of publishing a snapshot with holes (SEMANTICS.md §6b)
    mask = c["mask"] if c["mask"].size else None
    picks, scores, _ = orc.pick_batch(chain, c["pods"], oix, reqs, B, mask)
    assert np.array_equal(picks, c["pick"]), (seed, chain, P, B)
    assert np.array_equal(scores.view(np.uint64), c["score"].view(np.uint64)), (seed, chain, P, B)
    # the second CPU algorithm (tables + named pods; what bench.py times as the CPU baseline) on the unmasked form of the case
    p0, s0, _ = orc.pick_batch(chain, c["pods"], oix, reqs, B, None)
    for th in (1, 3):
        p1, s1 = orc.OracleTables(chain, c["pods"]).pick_batch(oix, reqs, B, threads=th)
        assert np.array_equal(p0, p1) and np.array_equal(s0.view(np.uint64), s1.view(np.uint64)), (seed, chain, P, B, th)
    # ordered fallbacks (SEMANTICS.md §3a): the oracle's list against the matrix form's stable sort
    index = {}
    for h, p_ in zip(c["index_hashes"].tolist(), c["index_pods"].tolist()):
        index.setdefault(h, set()).add(p_)
    for p_ in removed + holes:
        for s_ in index.values():
            s_.discard(p_)
    k = int(rng.integers(1, 9))
    want_p, want_s = gg.numpy_topk(chain, c["pods"], index, c["adapter"], c["n_blocks"], c["hashes"], mask, k)
    got_p, got_s = orc.pick_topk(chain, c["pods"], oix, reqs, k, mask)
    assert np.array_equal(got_p, want_p), (seed, chain, P, B, k)
    assert np.array_equal(got_s.view(np.uint64), want_s.view(np.uint64)), (seed, chain, P, B, k)


class _DictIndex:
    """SEMANTICS.md §6 / §6a as a dict of sets with per-hash stamps -- a second statement of the index state rules."""

    def __init__(self):
        self.sets, self.stamp, self.epoch = {}, {}, 1

    def insert(self, hashes, pods):
        for h, p in zip(hashes, pods):
            self.sets.setdefault(int(h), set()).add(int(p))
            self.stamp[int(h)] = self.epoch                      # every insert stamps the hash, present pod or not

    def remove_pod(self, pod):
        for h in list(self.sets):
            self.sets[h].discard(pod)
            if not self.sets[h]:
                del self.sets[h]                                 # an empty set behaves as absent

    def advance_epoch(self):
        self.epoch += 1
        return self.epoch

    def evict_older(self, e):
        gone = [h for h in self.sets if self.stamp[h] < e]
        for h in gone:
            del self.sets[h]
        return len(gone)

    def size(self):
        return len(self.sets)


@pytest.mark.parametrize("seed", range(12))
def test_index_state_rules_against_a_dict_model(pkg, orc, seed):
    """Random sequences of insert / insert-after-pick / remove_pod / epoch tick / evict on the oracle's index and on a dict
    model: live hash counts, eviction counts, every pod set and the picks computed from either index (oracle loops vs the numpy
    matrix form) agree after every operation."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_golden", os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gen_golden.py"))
    gg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gg)
    rng = np.random.Generator(np.random.PCG64(31000 + seed))
    P, B, R = int(rng.choice([5, 40, 130])), int(rng.choice([2, 6])), 24
    chain = [[(gg.PF, 3), (gg.KV, 1)], [(gg.Q, 2), (gg.KV, 2), (gg.L, 1), (gg.PF, 3)], [(gg.PF, 2), (gg.Q, 1)]][seed % 3]
    pods = gg.rand_pods(rng, P)
    universe = rng.integers(1, 2**63, (6, B), dtype=np.uint64)
    oix, dix = orc.OracleIndex(), _DictIndex()

    def batch():
        hs = universe[rng.integers(0, universe.shape[0], R)].copy()
        for r in range(R):
            if rng.random() < 0.5:
                cut = int(rng.integers(0, B))
                hs[r, cut:] = rng.integers(1, 2**63, B - cut, dtype=np.uint64)
        adapter = rng.integers(-1, 128, R).astype(np.int32)
        nb = rng.integers(0, B + 1, R).astype(np.uint32)
        return adapter, nb, hs

    for step in range(25):
        op = rng.choice(["insert", "insert", "insert_picks", "remove_pod", "tick", "evict"])
        if op == "insert":
            c = universe[int(rng.integers(0, universe.shape[0]))][: int(rng.integers(1, B + 1))]
            ps = rng.integers(0, P, c.size).astype(np.uint32)
            oix.insert(c, ps); dix.insert(c.tolist(), ps.tolist())
        elif op == "insert_picks":
            adapter, nb, hs = batch()
            reqs = pkg.picker.make_req_rows(adapter, nb, hs, B)
            picks, _, _ = orc.pick_batch(chain, pods, oix, reqs, B)
            oix.insert_picks(reqs, B, picks)
            for r in range(R):
                if picks[r] >= 0:
                    dix.insert(hs[r, : int(nb[r])].tolist(), [int(picks[r])] * int(nb[r]))
        elif op == "remove_pod":
            p = int(rng.integers(0, P))
            oix.remove_pod(p); dix.remove_pod(p)
        elif op == "tick":
            assert oix.advance_epoch() == dix.advance_epoch()
        else:
            e = max(dix.epoch - int(rng.integers(0, 3)), 0)
            assert oix.evict_older(e) == dix.evict_older(e), (seed, step)
        assert oix.size() == dix.size(), (seed, step, op)
        for h in universe.ravel().tolist():
            assert set(oix.lookup(h).tolist()) == dix.sets.get(h, set()), (seed, step, op)
        adapter, nb, hs = batch()
        reqs = pkg.picker.make_req_rows(adapter, nb, hs, B)
        picks, scores, _ = orc.pick_batch(chain, pods, oix, reqs, B)
        want_p, want_s = gg.numpy_pick(chain, pods, dix.sets, adapter, nb, hs, None)
        assert np.array_equal(picks, want_p), (seed, step, op)
        assert np.array_equal(scores.view(np.uint64), want_s.view(np.uint64)), (seed, step, op)


def test_chain_hash_against_python_xxhash_random(pkg, orc):
    """SEMANTICS.md §4 restated with python-xxhash: h[-1] = XXH64(model), h[i] = XXH64(block_i || LE64(h[i-1])), only full blocks,
    at most max_blocks -- against the oracle's and the library's host chain on random prompts, block sizes and limits."""
    xxhash = pytest.importorskip("xxhash")
    import struct
    rng = np.random.default_rng(44)
    for _ in range(60):
        bc = int(rng.choice([1, 7, 8, 16, 64, 100]))
        mb = int(rng.integers(0, 9))
        model = rng.integers(0, 256, int(rng.integers(0, 20)), dtype=np.uint8).tobytes()
        prompt = rng.integers(0, 256, int(rng.integers(0, 5 * bc + 3)), dtype=np.uint8).tobytes()
        h = xxhash.xxh64(model, seed=0).intdigest()
        want = []
        for i in range(min(len(prompt) // bc, mb)):
            h = xxhash.xxh64(prompt[i * bc:(i + 1) * bc] + struct.pack("<Q", h), seed=0).intdigest()
            want.append(h)
        want = np.array(want, dtype=np.uint64)
        assert np.array_equal(orc.hash_prompt(model, prompt, bc, mb), want), (bc, mb, len(prompt))
        assert np.array_equal(pkg.picker.hash_prompt(model, prompt, bc, mb), want), (bc, mb, len(prompt))


@pytest.mark.parametrize("seed", range(6))
def test_oracle_properties(pkg, orc, seed):
    """Properties any implementation of SEMANTICS.md must have, checked on the oracle (the GPU suite checks the same ones on the
    kernel at full size): request order does not matter; the ordered fallback list is prefix-closed and its head is the pick;
    masking the pick out promotes the first fallback; the post-pick index update is idempotent."""
    rng = np.random.default_rng(5150 + seed)
    wl = pkg.workload.make_workload(3, R=96, P=int(rng.choice([50, 200, 700])))
    oix = orc.OracleIndex()
    oix.insert(wl.index_hashes, wl.index_pods)
    picks, scores, _ = orc.pick_batch(wl.chain, wl.pods, oix, wl.reqs, wl.B)
    # (1) a batch is a set of independent requests
    perm = rng.permutation(wl.R)
    p2, s2, _ = orc.pick_batch(wl.chain, wl.pods, oix, wl.reqs[perm], wl.B)
    assert np.array_equal(p2, picks[perm]) and np.array_equal(s2.view(np.uint64), scores[perm].view(np.uint64))
    # (2) fallback lists: head == pick, lists of different k agree on their common prefix, totals never increase
    t4p, t4s = orc.pick_topk(wl.chain, wl.pods, oix, wl.reqs, 4)
    t2p, t2s = orc.pick_topk(wl.chain, wl.pods, oix, wl.reqs, 2)
    assert np.array_equal(t4p[:, 0], picks) and np.array_equal(t4p[:, :2], t2p) and np.array_equal(t4s[:, :2].view(np.uint64), t2s.view(np.uint64))
    assert np.all(t4s[:, :-1] >= t4s[:, 1:])
    ties = t4s[:, :-1] == t4s[:, 1:]
    assert np.all(t4p[:, :-1][ties] < t4p[:, 1:][ties])                       # equal totals: lowest index first
    # (3) without the picked pod the first fallback is picked -- when removing it leaves the QUEUE normalisers alone
    W = (wl.P + 63) // 64
    mask = np.full((wl.R, W), np.uint64(0xFFFFFFFFFFFFFFFF), dtype=np.uint64)
    if wl.P % 64:
        mask[:, -1] = np.uint64((1 << (wl.P % 64)) - 1)
    q = wl.pods["queue"]
    unique_extreme = {int(np.argmin(q)) if (q == q.min()).sum() == 1 else -1, int(np.argmax(q)) if (q == q.max()).sum() == 1 else -1}
    for r in range(wl.R):
        mask[r, picks[r] >> 6] &= ~(np.uint64(1) << np.uint64(picks[r] & 63))
    p3, _, _ = orc.pick_batch(wl.chain, wl.pods, oix, wl.reqs, wl.B, mask)
    keep = np.array([int(p) not in unique_extreme for p in picks])
    assert np.array_equal(p3[keep], t4p[keep, 1])
    # (4) index[hash] |= {pick} twice is the same as once
    oix.insert_picks(wl.reqs, wl.B, picks)
    n1 = oix.size()
    a = orc.pick_batch(wl.chain, wl.pods, oix, wl.reqs, wl.B)
    oix.insert_picks(wl.reqs, wl.B, picks)
    b = orc.pick_batch(wl.chain, wl.pods, oix, wl.reqs, wl.B)
    assert oix.size() == n1 and np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint64), b[1].view(np.uint64))
