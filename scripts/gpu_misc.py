#!/usr/bin/env python3
"""GPU box: secondary measurements — snapshot publish latency, on-device prompt hashing throughput, masked batch time."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
import torch
pkg = g.load_package()
out = {}
wl = pkg.workload.make_workload(5, R=65536)
pk = pkg.BatchedPicker(wl.chain, max_pods=4096, max_blocks=32, max_batch=65536, index_slots=wl.index_slots)
ts = []
for _ in range(12):
    t0 = time.perf_counter(); pk.publish(wl.pods); ts.append(time.perf_counter() - t0)
out["publish_ms_P4096"] = {"p50": float(np.median(ts[2:]) * 1e3), "max": float(np.max(ts[2:]) * 1e3)}
pk.index_insert(wl.index_hashes, wl.index_pods)
# device hashing: 64k prompts x 2 KiB
R, stride = 65536, 2048
d_p = torch.randint(0, 256, (R, stride), dtype=torch.uint8, device="cuda")
d_l = torch.full((R,), stride, dtype=torch.int32, device="cuda")
d_s = torch.randint(0, 2**62, (R,), dtype=torch.int64, device="cuda")
d_a = torch.zeros(R, dtype=torch.int32, device="cuda")
d_rows = torch.empty((R, 33), dtype=torch.int64, device="cuda")
side = torch.cuda.Stream(); torch.cuda.set_stream(side); st = side.cuda_stream; assert st != 0
for _ in range(3): pk.hash_prompts_device(d_p.data_ptr(), stride, d_l.data_ptr(), d_s.data_ptr(), d_a.data_ptr(), R, 64, d_rows.data_ptr(), st)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): pk.hash_prompts_device(d_p.data_ptr(), stride, d_l.data_ptr(), d_s.data_ptr(), d_a.data_ptr(), R, 64, d_rows.data_ptr(), st)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
out["hash_prompts_64k_x_2KiB"] = {"ms": ms, "GBps": R * stride / ms / 1e6, "prompts_per_s": R / ms * 1e3}
# masked batch (generic kernel) at C5 size
wm = pkg.workload.make_workload(5, R=8192, masked=True)
d_reqs = torch.from_numpy(wm.reqs.view(np.int64)).cuda(); d_mask = torch.from_numpy(wm.mask.view(np.int64)).cuda()
d_pick = torch.empty(8192, dtype=torch.int32, device="cuda"); d_sc = torch.empty(8192, dtype=torch.float64, device="cuda")
for _ in range(2): pk.pick_device(d_reqs.data_ptr(), 8192, d_mask.data_ptr(), d_pick.data_ptr(), d_sc.data_ptr(), st)
e0.record()
for _ in range(5): pk.pick_device(d_reqs.data_ptr(), 8192, d_mask.data_ptr(), d_pick.data_ptr(), d_sc.data_ptr(), st)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
out["masked_generic_8k_x_4096"] = {"ms": ms, "decisions_per_s": 8192 / ms * 1e3}
# post-pick index update (SEMANTICS.md §6) at C5 size: 64k requests x 32 blocks appended after a pick
R5 = 65536
d_reqs5 = torch.from_numpy(wl.reqs.view(np.int64)).cuda(); d_pick5 = torch.empty(R5, dtype=torch.int32, device="cuda"); d_sc5 = torch.empty(R5, dtype=torch.float64, device="cuda")
pk2 = pkg.BatchedPicker(wl.chain, max_pods=4096, max_blocks=32, max_batch=65536, index_slots=1 << 22)
pk2.publish(wl.pods); pk2.index_insert(wl.index_hashes, wl.index_pods)
pk2.pick_device(d_reqs5.data_ptr(), R5, None, d_pick5.data_ptr(), d_sc5.data_ptr(), st)
torch.cuda.synchronize()
ts = []
for i in range(4):
    e0.record(); pk2.index_insert_picks_device(d_reqs5.data_ptr(), d_pick5.data_ptr(), R5, st); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
out["insert_picks_64k_x_32"] = {"first_ms": ts[0], "repeat_ms": float(np.median(ts[1:])), "index_keys_after": pk2.index_size()}
e0.record()
for _ in range(5): pk2.pick_device(d_reqs5.data_ptr(), R5, None, d_pick5.data_ptr(), d_sc5.data_ptr(), st)
e1.record(); torch.cuda.synchronize()
out["pick_after_insert_picks_ms"] = e0.elapsed_time(e1) / 5
# small-batch latency through the host-buffer entry point (what the micro-batcher of the shim sees at low QPS)
import time as _t
for Rs in (1, 16, 256, 4096):
    sub = np.ascontiguousarray(wl.reqs[:Rs])
    for _ in range(20): pk.pick(sub)
    lat = []
    for _ in range(300):
        t0 = _t.perf_counter(); pk.pick(sub); lat.append(_t.perf_counter() - t0)
    lat = np.asarray(lat) * 1e6
    out[f"host_pick_latency_us_R{Rs}"] = {"p50": float(np.percentile(lat, 50)), "p99": float(np.percentile(lat, 99))}
print(json.dumps(out))
