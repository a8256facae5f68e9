"""Measurement tool: device and host latency of torch.distributed.all_gather_into_tensor (RCCL) at this world size, by message size.
Run under torchrun (or alone: world size 1)."""
import os, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29517")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
lr = int(os.environ.get("LOCAL_RANK", "0")); torch.cuda.set_device(lr)
dist.init_process_group("nccl", device_id=torch.device("cuda", lr))
W = dist.get_world_size()
for nbytes in (4096, 131072, 1 << 20, 2 << 20):
    src = torch.zeros(nbytes, dtype=torch.uint8, device="cuda"); out = torch.empty(W * nbytes, dtype=torch.uint8, device="cuda")
    for _ in range(5): dist.all_gather_into_tensor(out, src)
    torch.cuda.synchronize()
    host, dev, e2e = [], [], []
    for _ in range(50):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); t0 = time.perf_counter(); a.record()
        dist.all_gather_into_tensor(out, src)
        t1 = time.perf_counter(); b.record(); torch.cuda.synchronize(); t2 = time.perf_counter()
        host.append((t1 - t0) * 1e6); dev.append(a.elapsed_time(b) * 1e3); e2e.append((t2 - t0) * 1e6)
    if dist.get_rank() == 0:
        med = lambda x: sorted(x)[len(x) // 2]
        print(f"world {W} bytes/rank {nbytes}: host enqueue {med(host):.1f} us, device (events on the current stream) {med(dev):.1f} us, call -> synchronized {med(e2e):.1f} us", flush=True)
dist.destroy_process_group()
