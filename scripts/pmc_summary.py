#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: mean per dispatch of every counter for kernels matching a substring.
`--by-kernel` as a third argument keeps the kernels apart (one row per kernel name and counter)."""
import csv, glob, sys, collections, re
root, pat = sys.argv[1], sys.argv[2]
by_kernel = len(sys.argv) > 3 and sys.argv[3] == "--by-kernel"
acc = collections.defaultdict(lambda: collections.defaultdict(float))   # (kernel, counter) -> dispatch -> value (summed over XCD/SE rows)
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if pat in row["Kernel_Name"]:
            k = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "") if by_kernel else ""
            acc[(k, row["Counter_Name"])][(f, row["Dispatch_Id"])] += float(row["Counter_Value"])
print(("kernel," if by_kernel else "") + "counter,mean_per_dispatch,dispatches")
for (k, c) in sorted(acc):
    v = list(acc[(k, c)].values())
    print((f"{k}," if by_kernel else "") + f"{c},{sum(v)/len(v):.6g},{len(v)}")
