#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of the bench command into per-kernel HBM bytes per launch.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/traf_fetch -- python bench.py ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/traf_write -- python bench.py ...
    python tools/collect_traffic.py gpurun_out/traf_fetch gpurun_out/traf_write profiles/r01_traffic.json

Units and corrections as MI355X_MICROARCH.md (HBM section): both counters are in KiB; on gfx950 FETCH_SIZE reports
exactly half of the bytes of a wide coalesced streaming read, so it is doubled; WRITE_SIZE is exact."""
import collections, csv, glob, json, re, sys

def per_kernel(root, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{root}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc

def short(name):
    m = re.search(r"ttts::((?:gemm|wgrad)_\w+_kernel)<([^>]*)>", name)
    if not m:
        return None
    args = m.group(2).replace(" ", "")
    if m.group(1) == "gemm_h3_wide_kernel":
        return m.group(1)                              # clipped / unclipped loaders are one line
    if m.group(1) in ("gemm_bf16x6_kernel", "gemm_h3_kernel"):
        args = ",".join(args.split(",")[:4])          # clipped / unclipped loaders of one tile are one line
    if m.group(1) == "gemm_f32_kernel":
        parts = args.split(",")
        args = ",".join(parts[:4] + [parts[4], "*"]) if parts[4] == "true" else args
    return f"{m.group(1)}<{args}>"

fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
out = {}
for k in fetch:
    s = short(k)
    if s is None or k not in write:
        continue
    f = sum(fetch[k]) / len(fetch[k]) * 1024 * 2      # gfx950: FETCH_SIZE under-reports wide reads by 2x
    w = sum(write[k]) / len(write[k]) * 1024
    e = out.setdefault(s, {"launches": 0, "fetch": 0.0, "write": 0.0})
    e["fetch"] += f * len(fetch[k]); e["write"] += w * len(write[k]); e["launches"] += len(fetch[k])
res = {}
for s, e in out.items():
    res[s] = {"launches_profiled": e["launches"], "fetch_bytes_per_launch": e["fetch"] / e["launches"],
              "write_bytes_per_launch": e["write"] / e["launches"],
              "hbm_bytes_per_launch": (e["fetch"] + e["write"]) / e["launches"],
              "note": "FETCH_SIZE x2 (gfx950 wide-read correction), WRITE_SIZE exact; separate --pmc passes of bench.py"}
if len(sys.argv) > 4:
    res["_meta"] = sys.argv[4]
json.dump(res, open(sys.argv[3], "w"), indent=1)
print(json.dumps(res, indent=1))
