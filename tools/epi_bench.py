#!/usr/bin/env python3
"""Cost of the GEMM epilogue options on the FFN shapes (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformertts_amd import _lib, ops
from transformertts_amd.ops import _p, _stream
lib = _lib.load(); dev = torch.device("cuda:0")
def t(M, N, K, act, p, res, n=20):
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev); y = torch.empty(M, N, device=dev)
    r = torch.randn(M, N, device=dev) if res else None
    pl = ops._planes(w, 0, N, K)
    f = lambda: lib.ttts_linear_fwd_x6(_p(x), _p(pl), _p(b), _p(r), _p(y), M, N, K, act, p, 7, None, 0, 0, _stream())
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (M, N, K) in [(55680, 1024, 256), (55680, 256, 1024), (55680, 256, 256), (55680, 768, 256)]:
    print(f"M={M} N={N} K={K}: plain {t(M,N,K,0,0.0,False):7.1f}us | relu {t(M,N,K,1,0.0,False):7.1f} | relu+drop {t(M,N,K,1,0.1,False):7.1f} | drop+res {t(M,N,K,0,0.1,True):7.1f} | res {t(M,N,K,0,0.0,True):7.1f}")
