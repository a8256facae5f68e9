import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["EPPK_RESIDENT"] = "1"
import __graft_entry__ as g
pkg = g.load_package()
wl = pkg.workload.make_workload(5, R=4096)
n = 16
pk = pkg.BatchedPicker(wl.chain, max_pods=wl.P, max_blocks=wl.B, max_batch=256, index_slots=1 << 20)
pk.publish(wl.pods); pk.index_insert(wl.index_hashes, wl.index_pods)
sb, _ = pk.stage_buffers(0)
st, _ = pk.staging()
p, sc = np.empty(n, dtype=np.int32), np.empty(n, dtype=np.float64)
lib, ctx = pk._lib, pk._ctx
mode = sys.argv[1]
for i in range(14):
    np.copyto(sb[:n], wl.reqs[i * n:(i + 1) * n]); np.copyto(st[:n], wl.reqs[i * n:(i + 1) * n])
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < 200e-6: pass
    if mode == "alt" and i % 2 == 1:
        pk.pick_staged_into(n, p.ctypes.data, sc.ctypes.data)
    else:
        lib.eppk_pick_stage_begin(ctx, 0, n, 0, 1); lib.eppk_pick_stage_end(ctx, 0, p.ctypes.data, sc.ctypes.data)
pk.close()
