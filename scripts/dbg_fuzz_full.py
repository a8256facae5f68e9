import sys, os
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import __graft_entry__ as g
pkg, orc = g.load_package(), g.load_oracle()
import test_gpu_fuzz as t
for seed in (5506, 808, 1241):
    chain, pods, ih, ip, slots, reqs, mask, P, B, R = t._case(pkg, 1000 + seed)
    keys = sorted(set(ih.tolist()))
    res = []
    for rep in range(6):
        with pkg.BatchedPicker(chain, max_pods=P, max_blocks=B, max_batch=R, index_slots=slots) as pk:
            pk.publish(pods)
            try:
                pk.index_insert(ih, ip); err = None
            except Exception as e:
                err = str(e)[:40]
            res.append((err, pk.index_size(), pk.index_dropped()))
    # one key at a time: does every key fit?
    with pkg.BatchedPicker(chain, max_pods=P, max_blocks=B, max_batch=R, index_slots=slots) as pk:
        pk.publish(pods)
        bad = 0
        for i in range(ih.size):
            try:
                pk.index_insert(ih[i:i + 1], ip[i:i + 1])
            except Exception as e:
                bad += 1
        one = (bad, pk.index_size(), pk.index_dropped())
    act = int(((pods["flags"] & 1) == 0).sum())
    print(f"seed {seed}: keys {len(keys)} slots {slots} limit {slots // 2} pairs {ih.size} active pods {act}/{P} | bulk x6: {res} | pair by pair: {one}", flush=True)
