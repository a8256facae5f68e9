"""Development aid: where the time of ONE doorbell of the resident small-batch path goes.  Needs a measurement build of the library
(eppk_pick_resident.hip compiled with -DEPPK_RESIDENT_STAMPS, linked into ab/stamps/libeppk.so: scripts/README.md) -- the kernel then leaves
100 MHz timestamps in the control block and EPPK_RESIDENT_DEBUG=1 prints them per call.
    EPPK_LIB=ab/stamps/libeppk.so python scripts/res_stamps.py 2> stamps.txt ; python scripts/res_stamps.py --digest stamps.txt"""
import os, re, sys, time
if "--digest" in sys.argv:
    import numpy as np
    rows = {}
    n = None
    for line in open(sys.argv[sys.argv.index("--digest") + 1]):
        m = re.search(r"ring \d+ \(n = (\d+)\)", line)
        if m: n = int(m.group(1))
        m = re.search(r"ring -> done ([\d.]+) us.*invalidated (\d+), -> body done (\d+), -> released (\d+)", line)
        if m and n is not None: rows.setdefault(n, []).append([float(m.group(1))] + [int(m.group(i)) / 100.0 for i in (2, 3, 4)])
    for n, v in sorted(rows.items()):
        a = np.asarray(v[5:])
        p = np.percentile(a, 50, axis=0)
        print(f"n={n:3d} ({len(a)} calls) p50: ring -> done (host clock) {p[0]:6.2f} us | on the device: bell seen -> barrier + acquire + scalar cache invalidate {p[1]:5.2f} us, "
              f"-> argument block + pick body {p[2]:5.2f} us, -> release + barrier {p[3]:5.2f} us | the rest (doorbell and completion word over PCIe, poll phase) {p[0] - p[1] - p[2] - p[3]:5.2f} us")
    sys.exit(0)
os.environ["EPPK_RESIDENT"] = "1"
os.environ["EPPK_RESIDENT_DEBUG"] = "1"
sys.path.insert(0, ".")
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
wl = pkg.workload.make_workload(5, R=4096)
pk = pkg.BatchedPicker(wl.chain, max_pods=wl.P, max_blocks=wl.B, max_batch=64, index_slots=wl.index_slots)
pk.publish(wl.pods); pk.index_insert(wl.index_hashes, wl.index_pods)
st, _ = pk.staging()
for n in (1, 16, 32):
    p, s = np.empty(n, dtype=np.int32), np.empty(n, dtype=np.float64)
    for i in range(60):
        off = (i * n) % (wl.R - n)
        np.copyto(st[:n], wl.reqs[off:off + n])
        pk.pick_staged_into(n, p.ctypes.data, s.ctypes.data)
        time.sleep(0.0002)
pk.close()
