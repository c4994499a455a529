#!/usr/bin/env python3
"""How much of the time of the weight-gradient kernels in a rocprofv3 kernel trace overlaps other kernels (development aid):
usage tools/overlap_check.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda e: e[0])
tot = ov = 0
n = 0
for i, (s, e, name) in enumerate(ev):
    if "wgrad" not in name:
        continue
    n += 1
    tot += e - s
    for j in range(max(0, i - 8), min(len(ev), i + 8)):
        if j == i:
            continue
        s2, e2, _ = ev[j]
        ov += max(0, min(e, e2) - max(s, s2))
span = ev[-1][1] - ev[0][0]
busy = sum(e - s for s, e, _ in ev)
print(f"{n} wgrad launches, {tot/1e6:.2f} ms, of which {ov/1e6:.2f} ms overlap another kernel; trace span {span/1e6:.1f} ms, sum of kernel times {busy/1e6:.1f} ms")
