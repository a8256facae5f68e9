#!/usr/bin/env python3
"""Masked batches at several candidate densities: kernel time of the batch and how much of it pick_quad_kernel defers."""
import os, sys
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
m50 = wl.mask
m25 = m50 & r(); m12 = m25 & r(); m6 = m12 & r(); m3 = m6 & r()
with pkg.BatchedPicker(wl.chain, max_pods=wl.P, max_blocks=wl.B, max_batch=R, index_slots=wl.index_slots) as pk:
    pk.publish(wl.pods); pk.index_insert(wl.index_hashes, wl.index_pods)
    for name, m in [x for x in (("50%", m50), ("25%", m25), ("12%", m12), ("6%", m6), ("3%", m3)) if not os.environ.get("MASK_ONLY") or x[0] in os.environ["MASK_ONLY"].split(",")]:
        d_mask = torch.from_numpy(m.view(np.int64)).to(dev)
        l0, d0 = pk.quad_stats()
        for _ in range(3):
            pk.pick_device(d_reqs.data_ptr(), R, d_mask.data_ptr(), d_pick.data_ptr(), d_score.data_ptr(), st.cuda_stream)
        torch.cuda.synchronize()
        pk.profile(True)
        for _ in range(20):
            pk.pick_device(d_reqs.data_ptr(), R, d_mask.data_ptr(), d_pick.data_ptr(), d_score.data_ptr(), st.cuda_stream)
        torch.cuda.synchronize()
        ms = np.asarray(pk.profile_drain(), dtype=np.float64); pk.profile(False)
        l1, d1 = pk.quad_stats()
        cand = np.unpackbits(m[:256].view(np.uint8)).sum() / 256
        print(f"mask {name:4s}: {cand:7.1f} candidates/request  kernel avg {ms.mean() * 1e3:7.1f} us  quad launches {l1 - l0:3d}  deferred per quad launch {(d1 - d0) / max(1, l1 - l0):9.1f}", flush=True)
