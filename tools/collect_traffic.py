#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of the bench command into per-kernel HBM bytes per launch.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/traf_fetch -- python bench.py ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/traf_write -- python bench.py ...
    python tools/collect_traffic.py gpurun_out/traf_fetch gpurun_out/traf_write profiles/r04_traffic.json "<meta>" [max write bytes]
(run both passes with `--sustain 0` so that they execute the same steps)

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
    # Each counter is normalised by ITS OWN launch count: the two --pmc passes are separate runs of the bench command and
    # need not contain the same number of steps (round 3 divided the WRITE sum by the FETCH pass's launch count; with a
    # `sustained` stretch of different length in each pass that inflated every WRITE figure by 1.9x).
    e = out.setdefault(s, {"fetch": 0.0, "n_fetch": 0, "write": 0.0, "n_write": 0})
    e["fetch"] += sum(fetch[k]) * 1024 * 2            # gfx950: FETCH_SIZE under-reports wide reads by 2x
    e["n_fetch"] += len(fetch[k])
    e["write"] += sum(write[k]) * 1024
    e["n_write"] += len(write[k])
res = {}
for s, e in out.items():
    f, w = e["fetch"] / e["n_fetch"], e["write"] / e["n_write"]
    res[s] = {"launches_profiled": e["n_fetch"], "launches_profiled_write_pass": e["n_write"],
              "fetch_bytes_per_launch": f, "write_bytes_per_launch": w, "hbm_bytes_per_launch": f + w,
              "note": "FETCH_SIZE x2 (gfx950 wide-read correction), WRITE_SIZE exact; separate --pmc passes of bench.py, each "
                      "counter divided by the launches of its own pass"}
    if e["n_fetch"] != e["n_write"]:
        print(f"warning: {s}: {e['n_fetch']} launches in the FETCH pass, {e['n_write']} in the WRITE pass "
              "(run both passes with --sustain 0 so that they execute the same steps)", file=sys.stderr)
# Plausibility gate (the figure goes into the driver's bench line): a GEMM cannot write much more than its outputs.  The
# dominant kernel's launches at the default workload are M = 55 680 with N <= 1024: at most 228 MB of C per launch, so a
# per-launch WRITE average above 1.1 x that is a collection error, not a kernel property.
limit = float(sys.argv[5]) if len(sys.argv) > 5 else 1.1 * 55680 * 1024 * 4
for s, e in res.items():
    if s.startswith("gemm_") and e["write_bytes_per_launch"] > limit:
        raise SystemExit(f"{s}: {e['write_bytes_per_launch'] / 1e6:.1f} MB written per launch exceeds {limit / 1e6:.1f} MB "
                         "(largest possible output x 1.1): refusing to write the file")
if len(sys.argv) > 4:
    res["_meta"] = sys.argv[4]
json.dump(res, open(sys.argv[3], "w"), indent=1)
print(json.dumps(res, indent=1))
