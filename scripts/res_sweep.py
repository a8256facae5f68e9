"""Development aid: host-observed latency of one small batch through the resident path by batch size, for the two forms of the resident
kernel (EPPK_RESIDENT_QUAD_FROM=1: every batch rings pick_quad_kernel's body; =100000: every batch pick_fast_kernel's), bare ctypes call."""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
wl = pkg.workload.make_workload(5, R=4096)
os.environ["EPPK_RESIDENT"] = "1"
os.environ["EPPK_RESIDENT_MAX"] = "128"
for qf in ("100000", "1"):
    os.environ["EPPK_RESIDENT_QUAD_FROM"] = qf
    pk = pkg.BatchedPicker(wl.chain, max_pods=wl.P, max_blocks=wl.B, max_batch=128, index_slots=wl.index_slots)
    pk.publish(wl.pods); pk.index_insert(wl.index_hashes, wl.index_pods)
    st, _ = pk.staging()
    out = []
    for n in (1, 2, 4, 8, 12, 16, 24, 32, 48, 64):
        p, s = np.empty(n, dtype=np.int32), np.empty(n, dtype=np.float64)
        lat = []
        for i in range(320):
            off = (i * n) % (wl.R - n)
            np.copyto(st[:n], wl.reqs[off:off + n])
            t0 = time.perf_counter(); pk.pick_staged_into(n, p.ctypes.data, s.ctypes.data); lat.append(time.perf_counter() - t0)
        lat = np.asarray(lat[20:]) * 1e6
        out.append(f"{n}: {np.percentile(lat, 50):.1f}/{np.percentile(lat, 99):.1f}")
    print(("pick_fast_kernel's body" if qf != "1" else "pick_quad_kernel's body") + "  n: p50/p99 us  " + "  ".join(out), flush=True)
    pk.close()
