#!/usr/bin/env python3
"""Single picks on masks that leave a handful of candidates (what a subset filter produces): the general masked route (pick_quad_kernel parks
every row and scores four at a time; below EPPK_QUAD_MIN the fast kernel's list route) against the candidate-major kernel, by batch size.
Kernel time per batch (the library's own events)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
import torch
pkg = g.load_package()
RMAX = 65536
wl = pkg.workload.make_workload(5, R=RMAX, masked=True)
dev = torch.device("cuda", 0)
d_reqs = torch.from_numpy(wl.reqs.view(np.int64)).to(dev)
d_pick = torch.empty(RMAX, dtype=torch.int32, device=dev); d_score = torch.empty(RMAX, dtype=torch.float64, device=dev)
st = torch.cuda.Stream()
rng = np.random.default_rng(2)
W = (wl.P + 63) // 64
def timed(pk, fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); pk.profile(True)
    for _ in range(n): fn()
    torch.cuda.synchronize()
    ms = np.asarray(pk.profile_drain(), dtype=np.float64); pk.profile(False)
    return ms.mean() * 1e3
with pkg.BatchedPicker(wl.chain, max_pods=wl.P, max_blocks=wl.B, max_batch=RMAX, index_slots=wl.index_slots) as pk:
    pk.publish(wl.pods); pk.index_insert(wl.index_hashes, wl.index_pods)
    for ncand in (2, 8, 32):
        mask = np.zeros((RMAX, W), dtype=np.uint64)
        pods = rng.integers(0, wl.P, (RMAX, ncand))
        for j in range(ncand):
            np.bitwise_or.at(mask, (np.arange(RMAX), pods[:, j] // 64), np.uint64(1) << (pods[:, j] % 64).astype(np.uint64))
        d_mask = torch.from_numpy(mask.view(np.int64)).to(dev)
        for R in (256, 1024, 4096, 16384, 65536):
            a = timed(pk, lambda: pk.pick_device(d_reqs.data_ptr(), R, d_mask.data_ptr(), d_pick.data_ptr(), d_score.data_ptr(), st.cuda_stream))
            ref = d_pick[:R].clone(); refs = d_score[:R].clone()
            b = timed(pk, lambda: pk.pick_candidates_device(d_reqs.data_ptr(), R, d_mask.data_ptr(), 1, d_pick.data_ptr(), d_score.data_ptr(), st.cuda_stream))
            same = bool(torch.equal(ref, d_pick[:R])) and bool(torch.equal(refs, d_score[:R]))
            print(f"{ncand:3d} candidates  R = {R:6d}: general {a:7.1f} us   candidate-major {b:7.1f} us   same picks and scores: {same}", flush=True)
            if ncand == 8:
                d_p3 = torch.empty(R * 3, dtype=torch.int32, device=dev); d_s3 = torch.empty(R * 3, dtype=torch.float64, device=dev)
                a3 = timed(pk, lambda: pk._check(pk._lib.eppk_pick_topk_device(pk._ctx, d_reqs.data_ptr(), R, d_mask.data_ptr(), 3, d_p3.data_ptr(), d_s3.data_ptr(), st.cuda_stream), 'topk'))
                r3 = d_p3.clone(); rs3 = d_s3.clone()
                b3 = timed(pk, lambda: pk.pick_candidates_device(d_reqs.data_ptr(), R, d_mask.data_ptr(), 3, d_p3.data_ptr(), d_s3.data_ptr(), st.cuda_stream))
                same3 = bool(torch.equal(r3, d_p3)) and bool(torch.equal(rs3, d_s3))
                print(f"      top-3           R = {R:6d}: general {a3:7.1f} us   candidate-major {b3:7.1f} us   same: {same3}", flush=True)
