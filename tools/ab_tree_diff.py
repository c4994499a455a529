#!/usr/bin/env python3
"""Per-kernel difference of the two kernel-stats files tools/ab_tree.sh writes: usage tools/ab_tree_diff.py <a.csv> <b.csv> [steps_a] [steps_b]"""
import csv, re, sys
sa = float(sys.argv[3]) if len(sys.argv) > 3 else 11.0
sb = float(sys.argv[4]) if len(sys.argv) > 4 else sa
def load(f, steps):
    out = {}
    for r in csv.DictReader(open(f)):
        k = re.sub(r'\(.*', '', r['Name'])
        x = out.get(k, (0.0, 0.0, 0))
        out[k] = (x[0] + float(r['TotalDurationNs']) / steps / 1e6, float(r['AverageNs']) / 1e3, x[2] + int(r['Calls']))
    return out
a, b = load(sys.argv[1], sa), load(sys.argv[2], sb)
print(f"total {sum(v[0] for v in a.values()):.3f} -> {sum(v[0] for v in b.values()):.3f} ms/step; launches {sum(v[2] for v in a.values())/sa:.0f} -> {sum(v[2] for v in b.values())/sb:.0f}")
for k in sorted(set(a) | set(b), key=lambda k: -max(a.get(k, (0,))[0], b.get(k, (0,))[0])):
    x, y = a.get(k, (0, 0, 0)), b.get(k, (0, 0, 0))
    if abs(x[0] - y[0]) > 0.004:
        print(f"{x[0]:7.3f} -> {y[0]:7.3f}  ({x[1]:7.1f} -> {y[1]:7.1f} us, {x[2]/sa:.1f} -> {y[2]/sb:.1f} calls)  {k[:90]}")
