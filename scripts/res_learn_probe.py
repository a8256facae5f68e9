#!/usr/bin/env python3
"""Resident latency path: host-observed p50 of one 16-request batch -- plain pick (eppk_pick_batch_staged) and pick + LEARN
(eppk_pick_stage_begin(EPPK_PICK_LEARN) + _end) -- as a function of the idle time the host leaves between two calls."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["EPPK_RESIDENT"] = os.environ.get("EPPK_RESIDENT", "1")
import __graft_entry__ as g
pkg = g.load_package()
wl = pkg.workload.make_workload(5, R=65536)      # (enough rows that no request is sent twice)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
pk = pkg.BatchedPicker(wl.chain, max_pods=wl.P, max_blocks=wl.B, max_batch=256, index_slots=1 << 20)
pk.publish(wl.pods); pk.index_insert(wl.index_hashes, wl.index_pods)
st, _ = pk.staging(); sb, _ = pk.stage_buffers(0)
p, sc = np.empty(n, dtype=np.int32), np.empty(n, dtype=np.float64)
a_p, a_s = p.ctypes.data, sc.ctypes.data
lib, ctx = pk._lib, pk._ctx
def run(call, buf, gap_us, reps=300):
    lat = []
    for i in range(reps + 20):
        off = ((run.base + i) * n) % (wl.R - n)
        np.copyto(buf[:n], wl.reqs[off:off + n])
        t_w = time.perf_counter()
        while (time.perf_counter() - t_w) * 1e6 < gap_us:
            pass
        t0 = time.perf_counter(); call(); lat.append(time.perf_counter() - t0)
    run.base += reps + 20
    lat = np.asarray(lat[20:]) * 1e6
    return np.percentile(lat, 50), np.percentile(lat, 99), lat.mean()
run.base = 0
def plain(): pk.pick_staged_into(n, a_p, a_s)
def learn():
    lib.eppk_pick_stage_begin(ctx, 0, n, 0, 1); lib.eppk_pick_stage_end(ctx, 0, a_p, a_s)
def nolearn():
    lib.eppk_pick_stage_begin(ctx, 0, n, 0, 0); lib.eppk_pick_stage_end(ctx, 0, a_p, a_s)
for name, call, buf in (("plain pick_staged", plain, st), ("stage begin/end", nolearn, sb), ("stage begin(LEARN)/end", learn, sb)):
    for gap in (0, 10, 200):
        p50, p99, mean = run(call, buf, gap)
        print(f"{name:24s} n={n:3d} idle {gap:5d} us: p50 {p50:6.1f}  p99 {p99:6.1f}  mean {mean:6.1f} us", flush=True)
# where the time goes: begin and end apart
for gap in (0, 20, 200):
    tb, te = [], []
    for i in range(320):
        off = ((run.base + i) * n) % (wl.R - n)
        np.copyto(sb[:n], wl.reqs[off:off + n])
        t_w = time.perf_counter()
        while (time.perf_counter() - t_w) * 1e6 < gap:
            pass
        t0 = time.perf_counter(); lib.eppk_pick_stage_begin(ctx, 0, n, 0, 1); t1 = time.perf_counter(); lib.eppk_pick_stage_end(ctx, 0, a_p, a_s); t2 = time.perf_counter()
        tb.append(t1 - t0); te.append(t2 - t1)
    run.base += 320
    tb, te = np.asarray(tb[20:]) * 1e6, np.asarray(te[20:]) * 1e6
    print(f"LEARN idle {gap:4d} us: begin p50 {np.percentile(tb, 50):6.1f} p99 {np.percentile(tb, 99):6.1f}   end p50 {np.percentile(te, 50):6.1f} p99 {np.percentile(te, 99):6.1f}", flush=True)
print("resident stats", pk.resident_stats(), "index", pk.index_size(), "selfcheck", pk.index_selfcheck())
pk.close()
