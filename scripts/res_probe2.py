#!/usr/bin/env python3
"""Resident units side by side: alternate between variants (plain / LEARN / masked / top-4) call by call -- every unit must answer in
microseconds while the others stay resident (each on a hardware queue of its own)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["EPPK_RESIDENT"] = "1"
import __graft_entry__ as g
pkg = g.load_package()
wl = pkg.workload.make_workload(5, R=32768, masked=True)      # (enough rows that no request is sent twice: a revisit of a learned prompt takes the work-list pass)
n = 16
pk = pkg.BatchedPicker(wl.chain, max_pods=wl.P, max_blocks=wl.B, max_batch=256, index_slots=1 << 20)
pk.publish(wl.pods); pk.index_insert(wl.index_hashes, wl.index_pods)
sb, sbm = pk.stage_buffers(0, with_mask=True)
st, stm = pk.staging(with_mask=True)
W = (wl.P + 63) // 64
p, sc = np.empty(n * 4, dtype=np.int32), np.empty(n * 4, dtype=np.float64)
a_p, a_s = p.ctypes.data, sc.ctypes.data
lib, ctx = pk._lib, pk._ctx
calls = {
    "plain": lambda: pk.pick_staged_into(n, a_p, a_s),
    "learn": lambda: (lib.eppk_pick_stage_begin(ctx, 0, n, 0, 1), lib.eppk_pick_stage_end(ctx, 0, a_p, a_s)),
    "masked": lambda: pk.pick_staged_into(n, a_p, a_s, use_mask=True),
    "top4": lambda: lib.eppk_pick_topk(ctx, st.ctypes.data, n, None, 4, a_p, a_s),
    "learn_masked": lambda: (lib.eppk_pick_stage_begin(ctx, 0, n, 1, 1), lib.eppk_pick_stage_end(ctx, 0, a_p, a_s)),
}
base = 0
for names in (("plain",), ("learn",), ("plain", "learn"), ("plain", "learn", "masked", "top4"), ("plain", "learn", "masked", "top4", "learn_masked")):
    lat = {k: [] for k in names}
    for i in range(400):
        k = names[i % len(names)]
        off = ((base + i) * n) % (wl.R - n)
        np.copyto(st[:n], wl.reqs[off:off + n]); np.copyto(sb[:n], wl.reqs[off:off + n])
        stm[:n * W] = wl.mask[off:off + n].reshape(-1); sbm[:n * W] = wl.mask[off:off + n].reshape(-1)
        t_w = time.perf_counter()
        while (time.perf_counter() - t_w) < 40e-6:
            pass
        t0 = time.perf_counter(); calls[k](); lat[k].append(time.perf_counter() - t0)
    base += 400
    print(" | ".join(f"{k}: p50 {np.percentile(np.asarray(v[8:]) * 1e6, 50):6.1f} max {np.max(np.asarray(v[8:])) * 1e6:8.1f} us" for k, v in lat.items()), "| starts", pk.resident_stats()[2], flush=True)
pk.close()
