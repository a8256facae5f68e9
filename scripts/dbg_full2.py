import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
from test_gpu_fuzz import KV, PF
pods = pkg.workload.make_pods(7, 200, 128)
for slots, ppk, nres, minus in [(128, 2, 2, 0), (128, 2, 1, 0), (128, 2, 0, 0), (128, 2, 2, 1), (128, 2, 2, 2), (128, 2, 2, 3), (256, 4, 2, 0), (256, 4, 2, 2)]:
    rng = np.random.default_rng(slots * 10 + ppk)
    n_keys = slots // 2 - minus
    keys = rng.integers(1, 2**63, n_keys, dtype=np.uint64)
    if nres >= 1: keys[0] = np.uint64(0)
    if nres >= 2: keys[1] = np.uint64(0xFFFFFFFFFFFFFFFF)
    ih = np.repeat(keys, ppk); rng.shuffle(ih)
    ip = rng.integers(0, 200, ih.size).astype(np.uint32)
    res = []
    for rep in range(5):
        with pkg.BatchedPicker([(KV, 1), (PF, 3)], max_pods=200, max_blocks=4, max_batch=8, index_slots=slots) as pk:
            pk.publish(pods)
            try:
                pk.index_insert(ih, ip); err = ""
            except Exception as e:
                err = "FULL"
            res.append((err, pk.index_size(), pk.index_dropped()))
    print(f"slots {slots} pairs/key {ppk} reserved {nres} keys {n_keys} (limit {slots // 2}): {res}", flush=True)
