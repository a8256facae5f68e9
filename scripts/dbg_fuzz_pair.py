import sys, os
sys.path.insert(0, "."); sys.path.insert(0, "tests"); sys.path.insert(0, "scripts")
import numpy as np
import __graft_entry__ as g
pkg, orc = g.load_package(), g.load_oracle()
import test_gpu_fuzz as t
import dbg_fuzz_ix as d
for a in sys.argv[1:]:
    ps, ix = a.split(":")
    for rep in range(int(os.environ.get("REPS", "3"))):
        for s in ps.split(","):
            try:
                t.test_fuzz_pick(pkg, orc, int(s))
            except Exception as e:
                print("pick", s, repr(e)[:80])
        ok = d.run(int(ix))
        print("=== after pick seeds", ps, "index seed", ix, "rep", rep, "->", "ok" if ok else "FAILED", flush=True)
