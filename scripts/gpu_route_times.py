#!/usr/bin/env python3
"""Kernel time of the pick on routes other than the headline one (run on the GPU box): candidate masks (50 % random subsets and
sparse 1/8 subsets), snapshots with holes, ordered fallbacks k = 2 / 4 / 8 -- C5 shape, device-resident inputs, HIP events."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402


def main():
    import torch
    pkg = g.load_package()
    R = int(os.environ.get("ROUTE_R", "65536"))
    wl = pkg.workload.make_workload(5, R=R, masked=True)
    out = {"workload": wl.name, "requests": R}
    dev = torch.device("cuda", 0)
    d_reqs = torch.from_numpy(wl.reqs.view(np.int64)).to(dev)
    d_pick = torch.empty(R * 8, dtype=torch.int32, device=dev)
    d_score = torch.empty(R * 8, dtype=torch.float64, device=dev)
    st = torch.cuda.Stream()
    rng = np.random.default_rng(1)
    W = (wl.P + 63) // 64
    sparse = wl.mask & rng.integers(0, 2**63, (R, W), dtype=np.uint64) & rng.integers(0, 2**63, (R, W), dtype=np.uint64)
    masks = {"unmasked": None, "mask_50pct": wl.mask, "mask_12pct": sparse}

    def timed(pk, fn, n=60):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        pk.profile(True)
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        ms = np.asarray(pk.profile_drain(), dtype=np.float64)
        pk.profile(False)
        return float(ms.mean()), float(np.percentile(ms, 99))

    with pkg.BatchedPicker(wl.chain, max_pods=wl.P, max_blocks=wl.B, max_batch=R, index_slots=wl.index_slots) as pk:
        pk.publish(wl.pods)
        pk.index_insert(wl.index_hashes, wl.index_pods)
        for name, m in masks.items():
            d_mask = torch.from_numpy(m.view(np.int64)).to(dev) if m is not None else None
            mp = d_mask.data_ptr() if d_mask is not None else None
            avg, p99 = timed(pk, lambda: pk.pick_device(d_reqs.data_ptr(), R, mp, d_pick.data_ptr(), d_score.data_ptr(), st.cuda_stream))
            out[name] = {"kernel_avg_us": avg * 1e3, "kernel_p99_us": p99 * 1e3, "decisions_per_s": R / (avg * 1e-3)}
        for k in (2, 4, 8):
            avg, p99 = timed(pk, lambda: pk._check(pk._lib.eppk_pick_topk_device(pk._ctx, d_reqs.data_ptr(), R, None, k, d_pick.data_ptr(), d_score.data_ptr(),
                                                                                  st.cuda_stream), "topk"), n=30)
            out[f"topk_{k}"] = {"kernel_avg_us": avg * 1e3, "kernel_p99_us": p99 * 1e3}
        pods = wl.pods.copy()
        pods["flags"] = (rng.random(wl.P) < 0.1).astype(np.uint32)
        pk.publish(pods)
        avg, p99 = timed(pk, lambda: pk.pick_device(d_reqs.data_ptr(), R, None, d_pick.data_ptr(), d_score.data_ptr(), st.cuda_stream))
        out["holes_10pct_unmasked"] = {"kernel_avg_us": avg * 1e3, "kernel_p99_us": p99 * 1e3, "decisions_per_s": R / (avg * 1e-3)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
