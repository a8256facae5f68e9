"""Development aid: one sequence of tests/test_gpu_fuzz.py::test_fuzz_index_maintenance, verbose (the operations, and at the first
disagreement with the oracle what differs).  python scripts/dbg_fuzz_ix.py <seed> [<seed> ...]   (library mode through the environment)"""
import sys, os
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
import __graft_entry__ as g
pkg, orc = g.load_package(), g.load_oracle()
from test_gpu_fuzz import Q, KV, L, PF
def run(seed):
    rng = np.random.default_rng(5000 + seed)
    P = int(rng.choice([40, 300, 1500, 4096])); B = int(rng.choice([4, 8, 16]))
    chain = [[(KV, 1), (PF, 5)], [(Q, 1), (KV, 2), (L, 1), (PF, 4)], [(PF, 3), (KV, 5)], [(PF, 2), (Q, 1), (PF, 1)]][seed % 4]
    pods = pkg.workload.make_pods(int(rng.integers(1, 1 << 30)), P, 128)
    universe = rng.integers(1, 2**63, (24, B), dtype=np.uint64)
    R = 96
    def probe_batch():
        hs = universe[rng.integers(0, universe.shape[0], R)].copy()
        for r in range(R):
            if rng.random() < 0.5:
                cut = int(rng.integers(0, B)); hs[r, cut:] = rng.integers(1, 2**63, B - cut, dtype=np.uint64)
        return pkg.picker.make_req_rows(rng.integers(-1, 128, R), np.full(R, B), hs, B)
    print(f"seed {seed}: P {P} B {B} chain {chain}", flush=True)
    with pkg.BatchedPicker(chain, max_pods=P, max_blocks=B, max_batch=R, index_slots=8192) as pk:
        pk.publish(pods); oix = orc.OracleIndex()
        for step in range(14):
            op = rng.choice(["insert", "insert", "insert_picks", "remove_pod", "tick_evict", "republish", "trim"]); how = ""
            if op == "insert":
                ci = rng.integers(0, universe.shape[0], 3)
                ih = np.concatenate([universe[c, : int(rng.integers(1, B + 1))] for c in ci]); ip = rng.integers(0, P, ih.size).astype(np.uint32)
                pk.index_insert(ih, ip); oix.insert(ih, ip, snapshot=pods)
            elif op == "republish":
                pods = pods.copy(); flip = rng.random(P) < 0.15
                pods["flags"] = np.where(flip, pods["flags"] ^ 1, pods["flags"]).astype(np.uint32); pods["queue"] = rng.integers(0, 64, P)
                pk.publish(pods); oix.scrub_inactive(pods)
            elif op == "insert_picks":
                reqs = probe_batch(); d_reqs = torch.from_numpy(reqs.view(np.int64)).cuda()
                if rng.random() < 0.5:
                    how = "pick_learn_device"
                    d_picks = torch.empty(R, dtype=torch.int32, device="cuda")
                    pk.pick_learn_device(d_reqs.data_ptr(), R, None, d_picks.data_ptr(), None); torch.cuda.synchronize(); picks = d_picks.cpu().numpy()
                else:
                    how = "pick + insert_picks_device"
                    picks, _ = pk.pick(reqs); d_picks = torch.from_numpy(picks).cuda()
                    pk.index_insert_picks_device(d_reqs.data_ptr(), d_picks.data_ptr(), R); torch.cuda.synchronize()
                op_picks, _, _ = orc.pick_batch(chain, pods, oix, reqs, B)
                if not np.array_equal(picks, op_picks): print("   picks of the insert_picks batch differ at", np.nonzero(picks != op_picks)[0][:8])
                oix.insert_picks(reqs, B, op_picks)
            elif op == "trim":
                cap = int(rng.integers(1, 12)); a, b = pk.index_trim_pods(cap), oix.trim_pods(P, cap)
                if a != b: print(f"   trim {cap}: {a} vs oracle {b}")
            elif op == "remove_pod":
                pod = int(rng.integers(0, P)); pk.index_remove_pod(pod); oix.remove_pod(pod)
            else:
                e = pk.index_advance_epoch(); eo = oix.advance_epoch(); keep = int(rng.integers(1, 3))
                a, b = pk.index_evict_older(max(e - keep, 0)), oix.evict_older(max(e - keep, 0))
                if a != b: print(f"   evict: {a} vs oracle {b}")
            sc = pk.index_selfcheck(); sz = (pk.index_size(), oix.size()); dr = pk.index_dropped(); ls = pk.launch_status()
            reqs = probe_batch(); picks, scores = pk.pick(reqs); opk, osc, _ = orc.pick_batch(chain, pods, oix, reqs, B)
            bad = np.nonzero((picks != opk) | (scores.view(np.uint64) != osc.view(np.uint64)))[0]
            print(f"  step {step:2d} {op:13s} {how:27s} selfcheck {sc} size {sz} dropped {dr} status {ls} probe rows differing {bad.size}", flush=True)
            if bad.size or sc or sz[0] != sz[1]:
                for r in bad[:4]:
                    nb = int(reqs[r, 0] >> np.uint64(32)); print(f"    row {r}: adapter {np.int32(reqs[r,0] & np.uint64(0xFFFFFFFF))} nb {nb} gpu ({picks[r]}, {scores[r]!r}) oracle ({opk[r]}, {osc[r]!r})")
                return False
    return True
if __name__ == "__main__":
    for s in sys.argv[1:]:
        run(int(s))
