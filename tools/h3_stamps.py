#!/usr/bin/env python3
"""Phase timeline of the 256x256 fp16x3 GEMM tile loop (development aid; needs a library built with -DTTTS_EXP_STAMPS,
see tools/build_variant.sh): s_memtime stamps of every wave at the phase boundaries of each tile it processes.
usage: TTTS_LIB=<variant.so> tools/h3_stamps.py [M N K]"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from transformertts_amd import _lib, ops
from transformertts_amd.ops import _p, _stream

lib = _lib.load()
dev = torch.device("cuda:0")
M, N, K = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (55680, 1024, 256)
x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * K ** -0.5; b = torch.randn(N, device=dev)
y = torch.empty(M, N, device=dev)
xa = torch.zeros(ops.AMAX_SLOTS, device=dev)
lib.ttts_amax_partials(_p(x), x.numel(), _p(xa), _stream())
p3 = ops._planes(w, 4, N, K).clone()
run = lambda: lib.ttts_linear_fwd_h3(_p(x), _p(p3), _p(b), None, _p(y), M, N, K, 0, 0.0, 0, None, 0, 0, _p(xa), None, _stream())
for _ in range(5):
    run()
torch.cuda.synchronize()
raw = ctypes.CDLL(_lib.LIB_PATH)
raw.ttts_dbg_clear_stamps()
run()
torch.cuda.synchronize()
n = 512 * 8 * 6 * 8
buf = (ctypes.c_ulonglong * n)()
raw.ttts_dbg_read_stamps(buf, ctypes.c_size_t(n))
st = np.frombuffer(buf, dtype=np.uint64).reshape(512, 8, 6, 8).astype(np.int64)[:256]
t0 = st[:, :, 0, 3][st[:, :, 0, 3] > 0].min()
names = ["tile start", "loads requested", "tile 0 staged", "first barrier", "main loop end", "epilogue issued", "end barrier"]
print(f"M={M} N={N} K={K}; stamps relative to the earliest kernel entry, mean over workgroups (wave 0 / wave 4), s_memtime ticks")
for it in range(6):
    live = st[:, 0, it, 3] > 0
    if not live.any():
        break
    line = [f"tile {it} ({int(live.sum())} wgs)"]
    for ph in range(7):
        a = (st[live, 0, it, ph] - t0).mean(); c = (st[live, 4, it, ph] - t0).mean()
        line.append(f"{names[ph]} {a:8.0f}/{c:8.0f}")
    print(" | ".join(line))
    nw = 8 if (st[live][:, 4, it, 0] > 0).any() else 4
    d = st[live][:, :nw, it, :]
    seg = [("prologue issue", 0, 2), ("wait+barrier", 2, 3), ("main", 3, 4), ("epilogue", 4, 5), ("end barrier", 5, 6)]
    if not (d[:, :, 0] > 0).any():            # the one-wave-per-SIMD kernel has no per-tile prologue: phases 3 .. 6 only
        seg = seg[2:]
    print("     durations: " + "  ".join(f"{nm} {(d[:, :, b] - d[:, :, a]).mean():7.0f}" for nm, a, b in seg))
print("kernel span (ticks):", st[:, :, :, 6].max() - t0, " first-tile entry skew (max - min):", st[:, 0, 0, 3].max() - t0)
