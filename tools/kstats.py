#!/usr/bin/env python3
"""Per-step view of a rocprofv3 --kernel-trace --stats kernel_stats.csv: usage  tools/kstats.py <csv> <steps> [top]"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]); top = int(sys.argv[3]) if len(sys.argv) > 3 else 45
tot = sum(float(r['TotalDurationNs']) for r in rows)
cls = {}
for r in rows:
    n = r['Name']
    key = ('gemm fwd/dgrad' if re.search(r'gemm_(bf16x6|h3|h3i|h3_wide|f32)_kernel', n) else 'wgrad' if 'wgrad' in n else 'attention' if 'attn_' in n
           else 'other ttts' if 'ttts::' in n else 'non-ttts')
    cls[key] = cls.get(key, 0) + float(r['TotalDurationNs'])
for r in rows[:top]:
    n = re.sub(r'\(.*', '', r['Name'])[:90]
    print(f"{float(r['TotalDurationNs'])/steps/1e6:7.3f} ms/step {int(r['Calls'])/steps:7.1f} calls/step avg {float(r['AverageNs'])/1e3:8.1f} us  {n}")
print(f"total {tot/steps/1e6:.2f} ms/step, {sum(int(r['Calls']) for r in rows)/steps:.0f} launches/step")
for k, v in sorted(cls.items(), key=lambda kv: -kv[1]):
    print(f"  {k:16s} {v/steps/1e6:7.3f} ms/step")
