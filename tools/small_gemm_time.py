#!/usr/bin/env python3
"""fp16x3 forward GEMM (ttts_linear_fwd_h3) at encoder-sized row counts: us per launch against K -- the slope is the cost of one
32-deep k-tile of a workgroup (a latency chain on the 64 x 64 tile).  usage: tools/small_gemm_time.py [M [N]]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformertts_amd import _lib, ops
from transformertts_amd.ops import _p, _stream
lib = _lib.load(); dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 6400
Ns = [int(sys.argv[2])] if len(sys.argv) > 2 else [256, 512, 1024]
for N in Ns:
    out = []
    for K in (32, 64, 128, 256, 512, 1024):
        x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * K ** -0.5; b = torch.randn(N, device=dev)
        y = torch.empty(M, N, device=dev)
        pl = ops._planes(w, 4, N, K); xa = ops._amax(x)
        f = lambda: lib.ttts_linear_fwd_h3(_p(x), _p(pl), _p(b), None, _p(y), M, N, K, 0, 0.0, 0, None, 0, 0, _p(xa), None, _stream())
        for _ in range(5): assert f() == 0
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(20): f()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): g.replay()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 100 * 1e3
        ref = x.double() @ w.double().t() + b.double()
        err = float((y.double() - ref).norm() / ref.norm())
        out.append(f"K={K}: {us:5.1f} us ({err:.1e})")
    print(f"M={M} N={N}: " + " | ".join(out))
