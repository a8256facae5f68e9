#!/usr/bin/env python3
"""1/8-density candidate masks, the two-launch form (EPPK_QUAD_TAIL=0): run under `rocprofv3 --kernel-trace --stats` to see what the
work-list pass over the ~50 deferred requests (exact evaluation with the request's own QUEUE normalisers) costs by itself."""
import os, sys
os.environ["EPPK_QUAD_TAIL"] = "0"
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
import torch
pkg = g.load_package()
R = 65536
wl = pkg.workload.make_workload(5, R=R, masked=True)
dev = torch.device("cuda", 0)
d_reqs = torch.from_numpy(wl.reqs.view(np.int64)).to(dev)
d_pick = torch.empty(R, dtype=torch.int32, device=dev); d_score = torch.empty(R, dtype=torch.float64, device=dev)
st = torch.cuda.Stream()
rng = np.random.default_rng(1)
W = (wl.P + 63) // 64
r = lambda: rng.integers(0, 2**63, (R, W), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, (R, W), dtype=np.uint64)
m12 = wl.mask & r() & r()
with pkg.BatchedPicker(wl.chain, max_pods=wl.P, max_blocks=wl.B, max_batch=R, index_slots=wl.index_slots) as pk:
    pk.publish(wl.pods); pk.index_insert(wl.index_hashes, wl.index_pods)
    d_mask = torch.from_numpy(m12.view(np.int64)).to(dev)
    for _ in range(30):
        pk.pick_device(d_reqs.data_ptr(), R, d_mask.data_ptr(), d_pick.data_ptr(), d_score.data_ptr(), st.cuda_stream)
    torch.cuda.synchronize()
    print("quad stats", pk.quad_stats())
