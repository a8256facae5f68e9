"""Measurement tool: host-observed latency of one batch through the staged host path (eppk_pick_batch_staged) by batch size --
what a dispatcher that drains a few dozen to a few thousand pending requests per call sees (C5 snapshot: 4096 pods, 32 blocks)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
pkg = graft.load_package()
wl = pkg.workload.make_workload(5)
pk = pkg.BatchedPicker(wl.chain, max_pods=wl.P, max_blocks=wl.B, max_batch=wl.R, index_slots=wl.index_slots)
pk.publish(wl.pods); pk.index_insert(wl.index_hashes, wl.index_pods)
st, _ = pk.staging()
orc = graft.load_oracle(); oix = orc.OracleIndex(); oix.insert(wl.index_hashes, wl.index_pods)
out = {}
for n in (1, 16, 128, 512, 2048, 4096, 8192, 16384, 32768, 65536):
    np.copyto(st[:n], wl.reqs[:n])
    p, s = pk.pick_staged(n)
    if n <= 2048:
        op, os_, _ = orc.pick_batch(wl.chain, wl.pods, oix, wl.reqs[:n], wl.B)
        assert np.array_equal(p, op) and np.array_equal(s.view(np.uint64), os_.view(np.uint64)), n
    lat = []
    for i in range(400 if n <= 8192 else 100):
        off = (i * n) % max(1, wl.R - n + 1)
        np.copyto(st[:n], wl.reqs[off:off + n])            # fresh rows every call, as a dispatcher writes them (not timed)
        t0 = time.perf_counter(); pk.pick_staged(n); lat.append(time.perf_counter() - t0)
    lat = np.asarray(lat[20:]) * 1e6
    out[n] = dict(p50_us=float(np.percentile(lat, 50)), p99_us=float(np.percentile(lat, 99)), min_us=float(lat.min()))
    print(f"n={n:6d}: p50 {out[n]['p50_us']:8.1f} us  p99 {out[n]['p99_us']:8.1f} us  min {out[n]['min_us']:8.1f} us", flush=True)
print(json.dumps({"staged_latency_by_batch": out, "zero_copy_max": os.environ.get("EPPK_ZERO_COPY_MAX")}))
