import os, sys, json
import numpy as np
sys.path.insert(0, "/root/repo") if os.path.exists("/root/repo/__graft_entry__.py") else sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import __graft_entry__ as g
import torch
pkg = g.load_package()
wl = pkg.workload.make_workload(5, R=65536)
pk = pkg.BatchedPicker(wl.chain, max_pods=4096, max_blocks=32, max_batch=65536, index_slots=wl.index_slots)
pk.publish(wl.pods); pk.index_insert(wl.index_hashes, wl.index_pods)
lib = pk._lib
R = 65536
d_reqs = torch.from_numpy(wl.reqs.view(np.int64)).cuda()
side = torch.cuda.Stream(); torch.cuda.set_stream(side); st = side.cuda_stream
out = {}
for k in (2, 4, 8):
    d_pick = torch.empty(R * k, dtype=torch.int32, device="cuda"); d_sc = torch.empty(R * k, dtype=torch.float64, device="cuda")
    for _ in range(2): lib.eppk_pick_topk_device(pk._ctx, d_reqs.data_ptr(), R, None, k, d_pick.data_ptr(), d_sc.data_ptr(), st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): lib.eppk_pick_topk_device(pk._ctx, d_reqs.data_ptr(), R, None, k, d_pick.data_ptr(), d_sc.data_ptr(), st)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    out[f"topk{k}_64k_x_4096"] = {"ms": ms, "decisions_per_s": R / ms * 1e3}
print(json.dumps(out))
