"""Development aid: the resident small-batch path (EPPK_RESIDENT=1) end to end with a hard time limit -- parity of 1 / 16 / 64 requests
against the oracle and host-observed latency of eppk_pick_batch_staged, resident vs launched, on one box."""
import os, sys, time, faulthandler
faulthandler.dump_traceback_later(50, exit=True)
sys.path.insert(0, ".")
import numpy as np
import __graft_entry__ as g
pkg = g.load_package(); orc = g.load_oracle()
wl = pkg.workload.make_workload(5, R=4096)
oix = orc.OracleIndex(); oix.insert(wl.index_hashes, wl.index_pods)
for resident in ("1", "0"):
    os.environ["EPPK_RESIDENT"] = resident
    os.environ.setdefault("EPPK_RESIDENT_MAX", "128")
    pk = pkg.BatchedPicker(wl.chain, max_pods=wl.P, max_blocks=wl.B, max_batch=256, index_slots=wl.index_slots)
    pk.publish(wl.pods); pk.index_insert(wl.index_hashes, wl.index_pods)
    st, _ = pk.staging()
    for n in (1, 16, 17, 32, 64, 128):
        lat = []
        for i in range(420):
            off = (i * n) % (wl.R - n)
            np.copyto(st[:n], wl.reqs[off:off + n])
            t0 = time.perf_counter(); p, s = pk.pick_staged(n); lat.append(time.perf_counter() - t0)
        op, os_, _ = orc.pick_batch(wl.chain, wl.pods, oix, wl.reqs[off:off + n], wl.B)
        lat = np.asarray(lat[20:]) * 1e6
        print(f"EPPK_RESIDENT={resident} n={n:3d}: p50 {np.percentile(lat, 50):7.1f} us  p99 {np.percentile(lat, 99):7.1f} us  min {lat.min():7.1f} us   equal {np.array_equal(p, op) and np.array_equal(s.view(np.uint64), os_.view(np.uint64))}", flush=True)
    print("stats", pk.resident_stats(), flush=True)
    pk.close()
