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
        d_mask50 = torch.from_numpy(wl.mask.view(np.int64)).to(dev)
        for k in (2, 4, 8):                       # ordered fallbacks WITH candidate masks (50 % subsets): pick_quad_kernel<MASKED, TOPK>
            avg, p99 = timed(pk, lambda: pk._check(pk._lib.eppk_pick_topk_device(pk._ctx, d_reqs.data_ptr(), R, d_mask50.data_ptr(), k, d_pick.data_ptr(),
                                                                                  d_score.data_ptr(), st.cuda_stream), "topk"), n=30)
            ql1, qd1 = pk.quad_stats()
            out[f"mask_50pct_topk_{k}"] = {"kernel_avg_us": avg * 1e3, "kernel_p99_us": p99 * 1e3, "quad_launches_so_far": ql1, "quad_deferred_so_far": qd1}
        pods = wl.pods.copy()
        pods["flags"] = (rng.random(wl.P) < 0.1).astype(np.uint32)
        pk.publish(pods)
        avg, p99 = timed(pk, lambda: pk.pick_device(d_reqs.data_ptr(), R, None, d_pick.data_ptr(), d_score.data_ptr(), st.cuda_stream))
        out["holes_10pct_unmasked"] = {"kernel_avg_us": avg * 1e3, "kernel_p99_us": p99 * 1e3, "decisions_per_s": R / (avg * 1e-3)}
        # the subset filter of the whole batch on the device: 8 entries per request (half exact ports, half all-ports) -> mask rows
        import time
        endpoints = [pkg.picker.Endpoint(f"10.{p >> 8}.{p & 255}.{p % 7}", str(8000 + p % 4)) for p in range(wl.P)]
        pk.publish(wl.pods)
        pk.set_addresses(endpoints)
        n_f = 512                                                  # distinct filter strings, reused round robin over the batch
        filt = []
        for i in range(n_f):
            es = [endpoints[int(x)] for x in rng.integers(0, wl.P, 8)]
            filt.append(",".join(f"{e.address}:{e.port}" if j & 1 else e.address for j, e in enumerate(es)))
        t0 = time.perf_counter()
        per = [pkg.picker.subset_entries(f) for f in filt]
        t_tok = (time.perf_counter() - t0) / n_f
        keys = np.concatenate([per[r % n_f] for r in range(R)])
        off = np.arange(R + 1, dtype=np.uint32) * 8
        d_keys = torch.from_numpy(keys.view(np.int64)).to(dev)
        d_off = torch.from_numpy(off.view(np.int32)).to(dev)
        d_m = torch.empty((R, W), dtype=torch.int64, device=dev)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        with torch.cuda.stream(st):
            for _ in range(3):
                pk._check(pk._lib.eppk_subset_masks_device(pk._ctx, d_keys.data_ptr(), d_off.data_ptr(), R, d_m.data_ptr(), st.cuda_stream), "subset")
            ev[0].record(st)
            for _ in range(20):
                pk._check(pk._lib.eppk_subset_masks_device(pk._ctx, d_keys.data_ptr(), d_off.data_ptr(), R, d_m.data_ptr(), st.cuda_stream), "subset")
            ev[1].record(st)
        torch.cuda.synchronize()
        us = ev[0].elapsed_time(ev[1]) * 1e3 / 20
        t0 = time.perf_counter()
        for i in range(16):
            pkg.picker.subset_mask(endpoints, filt[i])
        t_host = (time.perf_counter() - t0) / 16
        got = d_m[:n_f].cpu().numpy().view(np.uint64)
        same = all(np.array_equal(got[i], pkg.picker.subset_mask(endpoints, filt[i])[0]) for i in range(0, n_f, 37))
        avg, p99 = timed(pk, lambda: pk.pick_device(d_reqs.data_ptr(), R, d_m.data_ptr(), d_pick.data_ptr(), d_score.data_ptr(), st.cuda_stream), n=20)
        ref_pick = d_pick[:R].clone()
        cavg, cp99 = timed(pk, lambda: pk.pick_candidates_device(d_reqs.data_ptr(), R, d_m.data_ptr(), 1, d_pick.data_ptr(), d_score.data_ptr(), st.cuda_stream))
        same_picks = bool(torch.equal(ref_pick, d_pick[:R]))
        c3avg, _ = timed(pk, lambda: pk.pick_candidates_device(d_reqs.data_ptr(), R, d_m.data_ptr(), 3, d_pick.data_ptr(), d_score.data_ptr(), st.cuda_stream), n=20)
        cand = float(np.mean([bin(int(x)).count("1") for x in got[:64].reshape(-1)]) * W)
        out["subset_filter_8_entries"] = {"mask_kernel_us": us, "requests_per_s": R / (us * 1e-6), "host_tokenise_us_per_request": t_tok * 1e6,
                                          "host_string_mask_us_per_request": t_host * 1e6, "masks_equal_string_exact": bool(same),
                                          "candidates_per_request": cand,
                                          "general_masked_kernel_avg_us": avg * 1e3,
                                          "candidate_major_kernel_avg_us": cavg * 1e3, "candidate_major_kernel_p99_us": cp99 * 1e3,
                                          "candidate_major_top3_avg_us": c3avg * 1e3, "picks_equal_general_kernel": same_picks,
                                          "decisions_per_s": R / (cavg * 1e-3)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
