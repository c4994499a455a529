#!/usr/bin/env python3
"""Per-kernel difference of the two kernel-stats files tools/ab_prof.sh writes: usage tools/ab_diff.py <outdir> [steps]"""
import csv, re, sys
d, steps = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 11.0
def load(f):
    out = {}
    for r in csv.DictReader(open(f)):
        out[re.sub(r'\(.*', '', r['Name'])] = (float(r['TotalDurationNs']) / steps / 1e6, float(r['AverageNs']) / 1e3, int(r['Calls']))
    return out
a, b = load(f'{d}/libttts_prev.csv'), load(f'{d}/libttts_hip.csv')
print(f"total {sum(v[0] for v in a.values()):.3f} -> {sum(v[0] for v in b.values()):.3f} ms/step; launches {sum(v[2] for v in a.values())/steps:.0f} -> {sum(v[2] for v in b.values())/steps:.0f}")
for k in sorted(set(a) | set(b), key=lambda k: -max(a.get(k, (0,))[0], b.get(k, (0,))[0])):
    x, y = a.get(k, (0, 0, 0)), b.get(k, (0, 0, 0))
    if abs(x[0] - y[0]) > 0.006:
        print(f"{x[0]:7.3f} -> {y[0]:7.3f}  ({x[1]:7.1f} -> {y[1]:7.1f} us, {x[2]/steps:.0f} -> {y[2]/steps:.0f} calls)  {k[:80]}")
