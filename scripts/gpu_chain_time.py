#!/usr/bin/env python3
"""GPU box: kernel time of the C5 batch under alternative scorer chains (which kernel serves them: eppk_chain_is_fused)."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
import torch
pkg = g.load_package()
Q, KV, L, PF = 1, 2, 3, 4
wl = pkg.workload.make_workload(5, R=65536)
d_reqs = torch.from_numpy(wl.reqs.view(np.int64)).cuda()
d_pick = torch.empty(65536, dtype=torch.int32, device="cuda"); d_sc = torch.empty(65536, dtype=torch.float64, device="cuda")
side = torch.cuda.Stream(); torch.cuda.set_stream(side); st = side.cuda_stream
out = {}
for name, chain in [("queue2,kv2,lora1,prefix3 (BASELINE)", wl.chain), ("prefix3,kv5 (reference example)", [(PF, 3), (KV, 5)]),
                    ("lora1,queue2,prefix3,kv2", [(L, 1), (Q, 2), (PF, 3), (KV, 2)]), ("prefix3,queue1,prefix3 (generic)", [(PF, 3), (Q, 1), (PF, 3)])]:
    pk = pkg.BatchedPicker(chain, max_pods=4096, max_blocks=32, max_batch=65536, index_slots=wl.index_slots)
    pk.publish(wl.pods); pk.index_insert(wl.index_hashes, wl.index_pods)
    for _ in range(3): pk.pick_device(d_reqs.data_ptr(), 65536, None, d_pick.data_ptr(), d_sc.data_ptr(), st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): pk.pick_device(d_reqs.data_ptr(), 65536, None, d_pick.data_ptr(), d_sc.data_ptr(), st)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    out[name] = {"kernel_kind": pk.chain_is_fused(), "ms_per_batch": ms, "decisions_per_s": 65536 / ms * 1e3}
    pk.close()
print(json.dumps(out))
