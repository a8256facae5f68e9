#!/usr/bin/env python3
"""Debug aid: a mixed batch of returning and new requests against the oracle, row by row (which rows differ, and how)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
import torch
pkg, orc = g.load_package(), g.load_oracle()
R = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
f = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
slots = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 23
wl = pkg.workload.make_workload(5, R=R)
fresh = pkg.workload.make_requests(wl, 0x5EED0005 ^ (0x9E3779B1 & 0x7FFFFFFF))
dev = torch.device("cuda", 0)
st = torch.cuda.Stream()
d_pick = torch.empty(R, dtype=torch.int32, device=dev); d_score = torch.empty(R, dtype=torch.float64, device=dev)
with pkg.BatchedPicker(wl.chain, max_pods=wl.P, max_blocks=wl.B, max_batch=R, index_slots=slots) as pk:
    pk.publish(wl.pods); pk.index_insert(wl.index_hashes, wl.index_pods)
    d0 = torch.from_numpy(wl.reqs.view(np.int64)).to(dev)
    pk.pick_learn_device(d0.data_ptr(), R, None, d_pick.data_ptr(), d_score.data_ptr(), st.cuda_stream); torch.cuda.synchronize()
    oix = orc.OracleIndex(); oix.insert(wl.index_hashes, wl.index_pods)
    op0, _, _ = orc.pick_batch(wl.chain, wl.pods, oix, wl.reqs, wl.B, threads=os.cpu_count())
    print("learned batch equal:", np.array_equal(d_pick.cpu().numpy(), op0), "selfcheck", pk.index_selfcheck())
    oix.insert_picks(wl.reqs, wl.B, op0)
    rows = pkg.workload.returning_rows(fresh, wl.reqs, f, 0x5EED0005)
    back = (rows == wl.reqs).all(axis=1)
    d1 = torch.from_numpy(rows.view(np.int64)).to(dev)
    q0 = pk.quad_stats()
    pk.pick_device(d1.data_ptr(), R, None, d_pick.data_ptr(), d_score.data_ptr(), st.cuda_stream); torch.cuda.synchronize()
    q1 = pk.quad_stats()
    op, osc, _ = orc.pick_batch(wl.chain, wl.pods, oix, rows, wl.B, threads=os.cpu_count())
    gp, gs = d_pick.cpu().numpy(), d_score.cpu().numpy()
    bad = np.nonzero((gp != op) | (gs.view(np.uint64) != osc.view(np.uint64)))[0]
    print(f"R={R} f={f}: quad launches {q1[0]-q0[0]} deferred {q1[1]-q0[1]}; differing rows {bad.size} (returning among them: {int(back[bad].sum())})")
    for r in bad[:12]:
        grp = [int(np.count_nonzero(back[(r // 4) * 4:(r // 4) * 4 + 4]))]
        print(f"  row {r} (block {r//4}, row-in-wave {r%4}, returning rows in its block {grp}) returning={bool(back[r])} gpu pick {gp[r]} score {gs[r]!r} | oracle {op[r]} {osc[r]!r}; learned pick of that row in batch 0: {op0[r]}")
