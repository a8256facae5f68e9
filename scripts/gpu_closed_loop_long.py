"""Development aid: a LONG closed loop against the oracle -- pick -> the index learns the picks -> next batch, an epoch tick + eviction every
second generation -- far beyond the 8 generations of tests/test_gpu_closed_loop.py: python scripts/gpu_closed_loop_long.py [generations] [R]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g
pkg, orc = g.load_package(), g.load_oracle()
gens = int(sys.argv[1]) if len(sys.argv) > 1 else 60
R = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
cores = os.cpu_count() or 1
wl = pkg.workload.make_workload(5, R=R)
batches = [wl.reqs] + [pkg.workload.make_requests(wl, 6000 + i) for i in range(5)]
keep = 2
t0 = time.time()
with pkg.BatchedPicker(wl.chain, max_pods=wl.P, max_blocks=wl.B, max_batch=R, index_slots=max(1 << 22, 1 << int(np.ceil(np.log2(256 * R))))) as pk:   # (room for the hashes of keep + 2 generations at load <= 1/4)
    pk.publish(wl.pods); pk.index_insert(wl.index_hashes, wl.index_pods)
    oix = orc.OracleIndex(); oix.insert(wl.index_hashes, wl.index_pods)
    d_b = [torch.from_numpy(b.view(np.int64)).cuda() for b in batches]
    d_pick = torch.empty(R, dtype=torch.int32, device="cuda"); d_score = torch.empty(R, dtype=torch.float64, device="cuda")
    st = torch.cuda.Stream(); bad = 0
    for gen in range(gens):
        b = gen % len(batches)
        if gen % 3:
            pk.pick_learn_device(d_b[b].data_ptr(), R, None, d_pick.data_ptr(), d_score.data_ptr(), st.cuda_stream)
        else:
            pk.pick_device(d_b[b].data_ptr(), R, None, d_pick.data_ptr(), d_score.data_ptr(), st.cuda_stream)
            pk.index_insert_picks_device(d_b[b].data_ptr(), d_pick.data_ptr(), R, st.cuda_stream)
        if gen % 2 == 1:
            e = pk.index_advance_epoch()
            if e > keep: pk.index_evict_older_device(e - keep + 1, st.cuda_stream)
        st.synchronize()
        op, osc, _ = orc.pick_batch(wl.chain, wl.pods, oix, batches[b], wl.B, threads=cores)
        oix.insert_picks(batches[b], wl.B, op)
        if gen % 2 == 1:
            assert oix.advance_epoch() == e
            if e > keep: oix.evict_older(e - keep + 1)
        p, s = d_pick.cpu().numpy(), d_score.cpu().numpy()
        ok = np.array_equal(p, op) and np.array_equal(s.view(np.uint64), osc.view(np.uint64)) and pk.index_size() == oix.size()
        if gen % 10 == 9: ok = ok and pk.index_selfcheck() == 0
        if not ok:
            bad += 1; print(f"generation {gen}: picks differ {int((p != op).sum())}, size {pk.index_size()} vs {oix.size()}, selfcheck {pk.index_selfcheck()}", flush=True)
            if bad >= 3: break
    print(f"closed loop, {gens} generations of {R} requests x {wl.P} pods, ageing every 2 (keep {keep}): {bad} generations differ; index {pk.index_size()} hashes, dropped {pk.index_dropped()}, "
          f"launch status {pk.launch_status()}, selfcheck {pk.index_selfcheck()}; {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
