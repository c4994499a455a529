#!/usr/bin/env python3
"""Where a wave of wgrad_dma_kernel spends its ticks (development aid; needs a library built with -DTTTS_WG_STAMPS:
hipcc ... -DTTTS_WG_STAMPS csrc/wgrad_dma.hip, link with the other objects, TTTS_LIB=<so> python tools/wgrad_stamps.py [M N K [idle ms between launches]])."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from transformertts_amd import _lib, ops
from transformertts_amd.ops import _p, _stream
lib = _lib.load(); dev = torch.device("cuda:0")
M, N, K = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (55680, 1024, 256)
gap_ms = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0       # idle time between launches (0: back to back)
x = torch.randn(M, K, device=dev); dy = torch.randn(M, N, device=dev) * 1e-6
dw = torch.empty(N, K, device=dev); db = torch.empty(N, device=dev)
ws = torch.empty(lib.ttts_wgrad_workspace_bytes(M, N, K, 1) // 4, device=dev)
am, xm = ops._amax(dy), ops._amax(x)
f = lambda: lib.ttts_linear_bwd_weight_h3(_p(dy), _p(x), _p(dw), _p(db), _p(ws), ws.numel() * 4, M, N, K, 0, 0, 0, _p(am), _p(xm), None, _stream())
for _ in range(5): f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
if gap_ms > 0:
    import time
    tot_us = 0.0
    for _ in range(30):
        time.sleep(gap_ms * 1e-3)
        e0.record(); f(); e1.record(); torch.cuda.synchronize()
        tot_us += e0.elapsed_time(e1) * 1e3
    print(f"{M}x{N}x{K}: {tot_us / 30:.1f} us per launch (kernel + reduction), {gap_ms} ms idle before each")
else:
    e0.record()
    for _ in range(30): f()
    e1.record(); torch.cuda.synchronize()
    print(f"{M}x{N}x{K}: {e0.elapsed_time(e1) / 30 * 1e3:.1f} us per launch (kernel + reduction), back to back")
raw = ctypes.CDLL(_lib.LIB_PATH)
if hasattr(raw, "ttts_dbg_wg_read_stamps"):
    n = 2048 * 8 * 8
    buf = (ctypes.c_ulonglong * n)()
    raw.ttts_dbg_wg_read_stamps(buf, ctypes.c_size_t(n))
    st = np.frombuffer(buf, dtype=np.uint64).reshape(2048, 8, 8).astype(np.float64)
    st = st[st[:, 0, 3] > 0]
    tot = st[:, :, 3].mean(); steps = st[:, :, 4].mean()
    ghz = st[:, :, 3].sum() / st[:, :, 7].sum() / 10.0
    print(f"{len(st)} workgroups; whole kernel {tot:.0f} ticks per wave (max {st[:, :, 3].max():.0f}) = {tot / ghz / 1e3:.1f} us at {ghz:.2f} GHz (s_memtime / s_memrealtime); {steps:.1f} steps")
    for i, nm in ((5, "prologue"), (0, "request + wait for rows"), (1, "products + conversion + fragment loads"), (2, "barrier"), (6, "epilogue")):
        v = st[:, :, i].mean()
        print(f"  {nm:40s} {v:9.0f} ticks = {v / tot * 100:5.1f} %" + (f"   per step {v / steps:7.1f}" if i in (0, 1, 2) else ""))
    print("  by wave (wait, body, barrier):", [[int(st[:, w, i].mean()) for i in (0, 1, 2)] for w in range(8)])
