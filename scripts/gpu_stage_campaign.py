"""Development aid: a randomized campaign over the PIPELINED host path (eppk_pick_stage_begin / _end over two staging sets): random batch
sizes on either side of the zero-copy limit and of the quad route's minimum, LEARN or not, candidate masks or not, and the ageing step
(eppk_index_advance_epoch + eppk_index_evict_older_device) issued at random points while a set is in flight; the oracle replays the
calls in order.   python scripts/gpu_stage_campaign.py [seconds] [first seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg, orc = g.load_package(), g.load_oracle()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
cores = os.cpu_count() or 1
t0 = time.time(); n_runs = n_batches = 0; fails = []
while time.time() - t0 < budget and len(fails) < 5:
    seed = seed0 + n_runs; n_runs += 1
    rng = np.random.default_rng(90000 + seed)
    Rmax = int(rng.choice([600, 5000, 20000]))
    P = int(rng.choice([300, 1500, 4096]))
    wl = pkg.workload.make_workload(5, R=Rmax, P=P, n_groups=int(rng.choice([4, 64])), masked=True)
    pool = [wl.reqs] + [pkg.workload.make_requests(wl, 100 + seed * 7 + i) for i in range(2)]
    keep = int(rng.integers(0, 3))
    with pkg.BatchedPicker(wl.chain, max_pods=P, max_blocks=wl.B, max_batch=Rmax, index_slots=1 << 21) as pk:
        pk.publish(wl.pods); pk.index_insert(wl.index_hashes, wl.index_pods)
        oix = orc.OracleIndex(); oix.insert(wl.index_hashes, wl.index_pods)
        bufs = [pk.stage_buffers(s, with_mask=True) for s in (0, 1)]
        expect = {}; learn_all = bool(rng.random() < 0.7)
        def begin(b):
            n = int(rng.choice([1, 17, 600, Rmax // 2, Rmax])); n = max(1, min(n, Rmax))
            src = pool[int(rng.integers(0, len(pool)))]; off = int(rng.integers(0, Rmax - n + 1))
            reqs = src[off:off + n]
            use_mask = (not learn_all) and rng.random() < 0.4
            learn = learn_all
            s = b & 1
            bufs[s][0][:n] = reqs
            m = None
            if use_mask:
                m = wl.mask[off:off + n]
                W = (P + 63) // 64
                bufs[s][1][: n * W] = m.reshape(-1)
            pk.stage_begin(s, n, use_mask=use_mask, learn=learn)
            op, osc, _ = orc.pick_batch(wl.chain, wl.pods, oix, reqs, wl.B, m, threads=cores)
            if learn: oix.insert_picks(reqs, wl.B, op)
            expect[b] = (op, osc)
        def tick():
            e = pk.index_advance_epoch(); assert oix.advance_epoch() == e
            if e > keep:
                pk.index_evict_older_device(e - keep); oix.evict_older(e - keep)
        def end(b):
            got = pk.stage_end(b & 1); want = expect.pop(b)
            if not (np.array_equal(got[0], want[0]) and np.array_equal(got[1].view(np.uint64), want[1].view(np.uint64))):
                fails.append((seed, b, int((got[0] != want[0]).sum()))); return False
            return True
        nb = int(rng.integers(3, 9)); ok = True
        begin(0)
        for b in range(1, nb):
            begin(b)
            if learn_all and rng.random() < 0.4: tick()      # (ageing is the one index entry point allowed with a set in flight -- every set a LEARN set here,
            ok = end(b - 1) and ok                            #  or all of them read-only: then the eviction queues behind their picks)
            if not ok: break
        if ok:
            if learn_all and rng.random() < 0.5: tick()
            ok = end(nb - 1)
        n_batches += nb
        if ok and (pk.index_size() != oix.size() or pk.index_selfcheck() != 0 or pk.launch_status() != 0):
            fails.append((seed, "final", pk.index_size(), oix.size()))
print(f"stage campaign: {n_runs} pipelines, {n_batches} batches in {time.time() - t0:.0f} s; {len(fails)} failures {fails}")
sys.exit(1 if fails else 0)
