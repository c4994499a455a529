#!/usr/bin/env python3
"""Time linear_fwd on a few shapes (development aid for A/B builds selected with TTTS_LIB)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformertts_amd import _lib, ops
from transformertts_amd.ops import _p, _stream
lib = _lib.load(); dev = torch.device("cuda:0")
def t(M, N, K, n=20):
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev); y = torch.empty(M, N, device=dev)
    if os.environ.get("X6", "1") == "1":
        pl = ops._planes(w, 0, N, K)
        f = lambda: lib.ttts_linear_fwd_x6(_p(x), _p(pl), _p(b), None, _p(y), M, N, K, 0, 0.0, 0, None, 0, 0, _stream())
    else:
        f = lambda: lib.ttts_linear_fwd(_p(x), _p(w), _p(b), None, _p(y), M, N, K, 0, 0.0, 0, None, 0, 0, _stream())
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    return us, 2.0 * M * N * K / us / 1e6
out = []
for (M, N, K) in [(55680, 256, 256), (55680, 1024, 256), (55680, 256, 1024), (55680, 1024, 1024)]:
    us, tf = t(M, N, K)
    out.append(f"{us:7.1f}us {tf:6.1f}TF")
print(os.path.basename(os.environ.get("TTTS_LIB", "default")).ljust(28), " | ".join(out))
