#!/usr/bin/env python3
"""What a batch of RETURNING requests costs: the same 64k batch is picked + learned, then picked again (every request now finds its 16
tail blocks listed on the pod it was routed to and its 16 prefix blocks on the group's pods: differing lists)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
import torch
pkg, orc = g.load_package(), g.load_oracle()
R = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
wl = pkg.workload.make_workload(5, R=R)
dev = torch.device("cuda", 0)
d_reqs = torch.from_numpy(wl.reqs.view(np.int64)).to(dev)
d_pick = torch.empty(R, dtype=torch.int32, device=dev); d_score = torch.empty(R, dtype=torch.float64, device=dev)
st = torch.cuda.Stream()
with pkg.BatchedPicker(wl.chain, max_pods=wl.P, max_blocks=wl.B, max_batch=R, index_slots=1 << 23) as pk:
    pk.publish(wl.pods); pk.index_insert(wl.index_hashes, wl.index_pods)
    def timed(n=20):
        for _ in range(3):
            pk.pick_device(d_reqs.data_ptr(), R, None, d_pick.data_ptr(), d_score.data_ptr(), st.cuda_stream)
        torch.cuda.synchronize(); pk.profile(True)
        for _ in range(n):
            pk.pick_device(d_reqs.data_ptr(), R, None, d_pick.data_ptr(), d_score.data_ptr(), st.cuda_stream)
        torch.cuda.synchronize()
        ms = np.asarray(pk.profile_drain(), dtype=np.float64); pk.profile(False)
        return ms.mean() * 1e3
    l0, d0 = pk.quad_stats()
    t_new = timed()
    l1, d1 = pk.quad_stats()
    pk.pick_learn_device(d_reqs.data_ptr(), R, None, d_pick.data_ptr(), d_score.data_ptr(), st.cuda_stream)
    torch.cuda.synchronize()
    l1b, d1b = pk.quad_stats()
    t_back = timed()
    l2, d2 = pk.quad_stats()
    picks = d_pick.cpu().numpy(); scores = d_score.cpu().numpy()
    ok = None
    if R <= 8192:
        oix = orc.OracleIndex(); oix.insert(wl.index_hashes, wl.index_pods)
        op, _, _ = orc.pick_batch(wl.chain, wl.pods, oix, wl.reqs, wl.B); oix.insert_picks(wl.reqs, wl.B, op)
        op2, os2, _ = orc.pick_batch(wl.chain, wl.pods, oix, wl.reqs, wl.B)
        ok = bool(np.array_equal(picks, op2)) and bool(np.array_equal(scores.view(np.uint64), os2.view(np.uint64)))
    print(f"R={R}: new requests {t_new:7.1f} us per batch (quad launches {l1 - l0}, deferred per launch {(d1 - d0) / max(1, l1 - l0):.0f}); "
          f"the same requests coming back {t_back:7.1f} us (quad launches {l2 - l1b}, deferred per launch {(d2 - d1b) / max(1, l2 - l1b):.0f}); equal oracle: {ok}")
