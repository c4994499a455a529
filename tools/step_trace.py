#!/usr/bin/env python3
"""One step of a rocprofv3 kernel trace as a table, in launch order: kernel, grid, duration, gap to the previous kernel's end.
usage: python3 tools/step_trace.py <kernel_trace.csv> [which step from the end, default 2]
(collect with `rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py --steps 6 --warmup 5 --no-cpu-baseline
--no-probe --sustain 0 --no-alignments-figure`; a step begins at weight_split_batched_kernel)"""
import csv, re, sys

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "weight_split_batched_kernel" in r["Kernel_Name"]]
a, b = starts[-back - 1], starts[-back]
prev_end = None
tot = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"^void ttts::|^ttts::|\(.*$", "", r["Kernel_Name"])
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    grid = "x".join(str(int(r[k]) // int(r[w])) for k, w in (("Grid_Size_X", "Workgroup_Size_X"), ("Grid_Size_Y", "Workgroup_Size_Y"), ("Grid_Size_Z", "Workgroup_Size_Z")))
    print(f"{(s - int(rows[a]['Start_Timestamp'])) / 1e3:9.1f} us  {(e - s) / 1e3:7.1f} us  gap {gap:6.1f}  {grid:>14s}  {name}")
    prev_end = max(e, prev_end or e)
    tot += e - s
print(f"# {b - a} launches, kernel time {tot / 1e6:.3f} ms, span {(int(rows[b]['Start_Timestamp']) - int(rows[a]['Start_Timestamp'])) / 1e6:.3f} ms")
