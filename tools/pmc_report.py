#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes written by tools/pmc_sets.sh: per kernel-name-substring, mean counter values."""
import collections, csv, glob, sys
root, pat = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
for f in sorted(glob.glob(f"{root}/set*/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(f"{k:34s} n={len(v):3d} mean={sum(v)/len(v):.5g}")
