import os, sys
sys.path.insert(0, os.getcwd())
os.environ["EPPK_RESIDENT"] = "1"
import numpy as np
import __graft_entry__ as g
pkg, orc = g.load_package(), g.load_oracle()
wl = pkg.workload.make_workload(5, R=64, P=300)
with pkg.BatchedPicker(wl.chain, max_pods=300, max_blocks=wl.B, max_batch=64, index_slots=wl.index_slots) as pk:
    pk.publish(wl.pods[:0])
    print("topk on an empty snapshot:", pk.pick_topk(wl.reqs[:4], 2)[0].tolist())
    print("pick on an empty snapshot:", pk.pick(wl.reqs[:4])[0].tolist())
    sb, _ = pk.stage_buffers(0); sb[:4] = wl.reqs[:4]
    pk.stage_begin(0, 4); print("staged:", pk.stage_end(0)[0].tolist())
    pk.publish(wl.pods); pk.index_insert(wl.index_hashes, wl.index_pods)
    oix = orc.OracleIndex(); oix.insert(wl.index_hashes, wl.index_pods)
    tp, ts = pk.pick_topk(wl.reqs[:4], 2)
    op, osc = orc.pick_topk_batch(wl.chain, wl.pods, oix, wl.reqs[:4], wl.B, 2)
    print("after a real publish equal oracle:", bool(np.array_equal(tp, op)), "resident stats", pk.resident_stats())
