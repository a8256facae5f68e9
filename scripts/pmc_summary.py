#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: mean per dispatch of every counter for kernels matching a substring."""
import csv, glob, sys, collections
root, pat = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float))   # counter -> dispatch -> value (summed over XCD/SE rows)
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if pat in row["Kernel_Name"]:
            acc[row["Counter_Name"]][(f, row["Dispatch_Id"])] += float(row["Counter_Value"])
print("counter,mean_per_dispatch,dispatches")
for c in sorted(acc):
    v = list(acc[c].values())
    print(f"{c},{sum(v)/len(v):.6g},{len(v)}")
