#!/usr/bin/env python3
"""GPU box: host-side cost per call of the operations in bench.py's N>1 step (what limits the step rate once the kernel is short)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
import __graft_entry__ as graft
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
pkg = graft.load_package()
wl = pkg.workload.make_workload(5, R=256)
pk = pkg.BatchedPicker(wl.chain, max_pods=wl.P, max_blocks=wl.B, max_batch=wl.R, index_slots=wl.index_slots)
pk.publish(wl.pods); pk.index_insert(wl.index_hashes, wl.index_pods)
dev = torch.device("cuda", 0)
d_reqs = torch.from_numpy(wl.reqs.view(np.int64)).to(dev)
d_pick = torch.empty(wl.R, dtype=torch.int32, device=dev); d_score = torch.empty(wl.R, dtype=torch.float64, device=dev)
d_all = torch.empty(wl.R, dtype=torch.int32, device=dev)
compute = torch.cuda.Stream(device=dev); comm = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(comm)
ev = torch.cuda.Event()
N = 2000
def timeit(name, fn):
    for _ in range(50): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(N): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize()
    print(f"{name}: {1e6*(t1-t0)/N:.1f} us host per call")
p = (d_reqs.data_ptr(), d_pick.data_ptr(), d_score.data_ptr())
timeit("pick_device (profiling off)", lambda: pk.pick_device(p[0], wl.R, None, p[1], p[2], compute.cuda_stream))
pk.profile(True)
timeit("pick_device (profiling on: hipExtLaunchKernel + events)", lambda: pk.pick_device(p[0], wl.R, None, p[1], p[2], compute.cuda_stream))
pk.profile_drain(); pk.profile(False)
timeit("event record + wait_event", lambda: (ev.record(compute), comm.wait_event(ev)))
timeit("all_gather_into_tensor (1 rank)", lambda: dist.all_gather_into_tensor(d_all, d_pick))
dist.destroy_process_group(); pk.close()
