#!/usr/bin/env python3
"""Turn the rocprofv3 --pmc passes of a bench run into the stamped JSON files bench.py reads (profiles/pmc_traffic*.json,
profiles/pmc_issue.json).  HBM bytes are EXACT: 32 * RDREQ_32B + 64 * RDREQ_64B + 128 * RDREQ_128B (the L2's memory-side request
counters by size; FETCH_SIZE alone tallies the 128-byte requests of coalesced streams at 64 bytes: profiles/r02_fetchcal.txt) plus
WRITE_SIZE.   usage: make_pmc_json.py <pmc dir> <kernel substring> <workload name> <out json> [--issue <out issue json> <requests>]"""
import csv, glob, json, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (kernel_source_hash)

root, pat, workload, out = sys.argv[1:5]
acc = collections.defaultdict(lambda: collections.defaultdict(float))   # (pattern index, counter) -> dispatch -> value
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        for pi, one in enumerate(pat.split("+")):      # "a+b": a launch is one dispatch of kernel a AND one of kernel b (means are added)
            if one in row["Kernel_Name"]:
                acc[(pi, row["Counter_Name"])][(f, row["Dispatch_Id"])] += float(row["Counter_Value"])
                break
mean = collections.defaultdict(float)
for (pi, c), v in acc.items():
    mean[c] += sum(v.values()) / len(v)
mean = dict(mean)
kh = bench.kernel_source_hash()
n32, n64, n128, nall = (mean.get(k, 0.0) for k in ("TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum", "TCC_EA0_RDREQ_sum"))
if n64 == 0.0 and nall:      # (no 64B counter in this pass set: the rest of the requests)
    n64 = nall - n32 - n128
read_b = 32.0 * n32 + 64.0 * n64 + 128.0 * n128
write_b = mean.get("WRITE_SIZE", 0.0) * 1024.0
doc = {"kernel_src_sha16": kh, "workload": workload, "kernel": pat,
       "hbm_bytes_per_launch": read_b + write_b, "read_bytes": read_b, "write_bytes": write_b,
       "rdreq_32B": n32, "rdreq_64B": n64, "rdreq_128B": n128, "rdreq_all": nall,
       "fetch_size_kb_as_reported": mean.get("FETCH_SIZE"), "write_size_kb": mean.get("WRITE_SIZE"),
       "tcc_hit_rate": (mean["TCC_HIT_sum"] / (mean["TCC_HIT_sum"] + mean["TCC_MISS_sum"])) if "TCC_HIT_sum" in mean and (mean["TCC_HIT_sum"] + mean["TCC_MISS_sum"]) else None,
       "note": "separate rocprofv3 --pmc passes, mean per dispatch of the named kernel; read bytes = 32*RDREQ_32B + 64*RDREQ_64B + 128*RDREQ_128B"}
json.dump(doc, open(out, "w"), indent=1)
print(json.dumps(doc))
if "--issue" in sys.argv:
    i = sys.argv.index("--issue")
    iout, reqs = sys.argv[i + 1], float(sys.argv[i + 2])
    d = {"kernel_src_sha16": kh, "kernel": pat, "requests_per_launch": reqs, "sclk_hz": 2.0e9,
         "valu_per_decision": mean.get("SQ_INSTS_VALU", 0) / reqs, "salu_per_decision": mean.get("SQ_INSTS_SALU", 0) / reqs,
         "branch_per_decision": mean.get("SQ_INSTS_BRANCH", 0) / reqs, "lds_per_decision": mean.get("SQ_INSTS_LDS", 0) / reqs,
         "vmem_rd_per_decision": mean.get("SQ_INSTS_VMEM_RD", 0) / reqs, "smem_per_decision": mean.get("SQ_INSTS_SMEM", 0) / reqs,
         "wave_cycles_quad": mean.get("SQ_WAVE_CYCLES"), "active_inst_any_quad": mean.get("SQ_ACTIVE_INST_ANY"), "wait_any_quad": mean.get("SQ_WAIT_ANY"),
         "wait_inst_any_quad": mean.get("SQ_WAIT_INST_ANY"),
         # the vector memory pipe: address-unit busy cycles (average over the CUs) against the kernel's length in L2 clocks, tag look-ups, L2 requests
         "ta_busy_cycles_avg": mean.get("TA_BUSY_avr"), "kernel_cycles_tcc_busy_avg": mean.get("TCC_BUSY_avr"),
         "tcp_accesses_per_decision": (mean.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0) / reqs) or None,
         "l2_requests_per_decision": (mean.get("TCC_REQ_sum", 0) / reqs) or None,
         "note": "SQ counters, mean per dispatch; *_quad in 4-cycle units summed over wavefronts; sclk 2.0 GHz measured for 40-us kernels (profiles/r02_clockcal.txt)"}
    json.dump(d, open(iout, "w"), indent=1)
    print(json.dumps(d))
