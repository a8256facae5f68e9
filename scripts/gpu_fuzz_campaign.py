"""Development aid: the randomized GPU tests of tests/test_gpu_fuzz.py over seeds BEYOND the ranges the suite runs (240 pick cases, 40
index-maintenance sequences), for a time budget:  python scripts/gpu_fuzz_campaign.py [seconds] [first pick seed] [first index seed].
Prints the first failures (seed + assertion) and a summary line."""
import os, sys, time, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg, orc = g.load_package(), g.load_oracle()
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_gpu_fuzz as t
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
s_pick = int(sys.argv[2]) if len(sys.argv) > 2 else 240
s_ix = int(sys.argv[3]) if len(sys.argv) > 3 else 40
t0 = time.time()
n_pick = n_ix = 0
fails = []
while time.time() - t0 < budget and len(fails) < 12:
    for fn, seed, kind in ((t.test_fuzz_pick, s_pick + n_pick, "pick"), (t.test_fuzz_index_maintenance, s_ix + n_ix, "index")):
        try:
            fn(pkg, orc, seed)
        except Exception as e:          # (AssertionError included)
            fails.append((kind, seed, repr(e)[:300]))
            traceback.print_exc(limit=2)
        if kind == "pick":
            n_pick += 1
        else:
            n_ix += 1
print(f"fuzz campaign: {n_pick} pick cases (seeds {s_pick}..{s_pick + n_pick - 1}), {n_ix} index sequences (seeds {s_ix}..{s_ix + n_ix - 1}) in {time.time() - t0:.0f} s; "
      f"{len(fails)} failures {fails}")
sys.exit(1 if fails else 0)
