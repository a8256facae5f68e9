"""Measurement tool: kernel duration (HIP events riding on the dispatch) vs host-observed latency of small staged batches."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
pkg = graft.load_package()
wl = pkg.workload.make_workload(5)
pk = pkg.BatchedPicker(wl.chain, max_pods=wl.P, max_blocks=wl.B, max_batch=wl.R, index_slots=wl.index_slots)
pk.publish(wl.pods); pk.index_insert(wl.index_hashes, wl.index_pods)
st, _ = pk.staging()
for n in (1, 16, 128, 512, 2048):
    np.copyto(st[:n], wl.reqs[:n])
    for _ in range(20): pk.pick_staged(n)
    pk.profile(1)
    lat = []
    for i in range(200):
        t0 = time.perf_counter(); pk.pick_staged(n); lat.append(time.perf_counter() - t0)
    k = np.asarray(pk.profile_drain()) * 1e3
    pk.profile(False)
    print(f"n={n:5d}: kernel avg {k.mean():6.1f} us (min {k.min():5.1f}), host-observed p50 with events {np.percentile(np.asarray(lat) * 1e6, 50):6.1f} us", flush=True)
