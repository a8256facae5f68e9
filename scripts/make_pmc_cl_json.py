#!/usr/bin/env python3
"""rocprofv3 --pmc passes of `bench.py --closed-loop` (scripts/gpu.sh pmc_cl) -> profiles/pmc_traffic_closed_loop.json, stamped with the
kernel-source hash: HBM bytes per closed-loop STEP = the LEARN pick + the update kernels (budget, insert, list sort) + the eviction
divided by age_every.  Exact bytes as in make_pmc_json.py: 32 * RDREQ_32B + 64 * RDREQ_64B + 128 * RDREQ_128B + WRITE_SIZE.
usage: make_pmc_cl_json.py <pmc dir> <requests> <age_every> <out json>"""
import collections, csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

root, requests, age_every, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
GROUPS = {"pick": ("pick_quad_kernel", "pick_fast_kernel"),
          "index_update": ("index_budget_kernel", "index_insert_picks_kernel", "index_lists_sort_kernel"),
          "ageing": ("index_evict_kernel",)}
acc = collections.defaultdict(lambda: collections.defaultdict(float))       # (kernel, counter) -> dispatch -> value
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        for names in GROUPS.values():
            for k in names:
                if k in row["Kernel_Name"]:
                    acc[(k, row["Counter_Name"])][(f, row["Dispatch_Id"])] += float(row["Counter_Value"])


def per_dispatch(kernel):
    """bytes per dispatch of one kernel; the FIRST third of its dispatches (warm-up, verification) is dropped"""
    m = {}
    for (k, c), v in acc.items():
        if k != kernel:
            continue
        vals = [x for _, x in sorted(v.items(), key=lambda kv: (kv[0][0], int(kv[0][1])))]
        vals = vals[len(vals) // 3:] or vals
        m[c] = sum(vals) / len(vals)
    if not m:
        return None
    n32, n128, nall = m.get("TCC_EA0_RDREQ_32B_sum", 0.0), m.get("TCC_EA0_RDREQ_128B_sum", 0.0), m.get("TCC_EA0_RDREQ_sum", 0.0)
    n64 = m.get("TCC_EA0_RDREQ_64B_sum", 0.0) or max(0.0, nall - n32 - n128)
    rd = 32.0 * n32 + 64.0 * n64 + 128.0 * n128
    wr = m.get("WRITE_SIZE", 0.0) * 1024.0
    return {"read_bytes": rd, "write_bytes": wr, "bytes": rd + wr, "rdreq_128B": n128, "rdreq_64B": n64}


doc = {"kernel_src_sha16": bench.kernel_source_hash(), "requests": requests, "age_every": age_every, "kernels": {}}
total = 0.0
for g, names in GROUPS.items():
    gb = 0.0
    for k in names:
        d = per_dispatch(k)
        if d:
            doc["kernels"][k] = d
            gb += d["bytes"]
    key = "ageing_per_step" if g == "ageing" else g
    doc[key] = gb / age_every if g == "ageing" else gb
    total += doc[key]
doc["hbm_bytes_per_step"] = total
doc["note"] = ("separate rocprofv3 --pmc passes of bench.py --closed-loop; mean per dispatch (first third of each kernel's dispatches dropped); the eviction runs every "
               f"{age_every} steps: its bytes are divided by that")
json.dump(doc, open(out, "w"), indent=1)
print(json.dumps(doc))
