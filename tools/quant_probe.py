#!/usr/bin/env python3
"""Tile quantization of the persistent GEMM kernels: us per launch of ttts_linear_fwd_h3 against the row count around M = 55 680
(217.5 row tiles of 256: 870 tiles of 256 x 256 at N = 1024 = 3.4 rounds over 256 CUs).  usage: tools/quant_probe.py [N K]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformertts_amd import _lib, ops
from transformertts_amd.ops import _p, _stream
lib = _lib.load(); dev = torch.device("cuda:0")
shapes = [(int(sys.argv[1]), int(sys.argv[2]))] if len(sys.argv) > 2 else [(1024, 256), (768, 256), (256, 1024), (256, 256)]
for N, K in shapes:
    out = []
    for M in (32768, 49152, 52224, 55680, 58368, 65536):
        x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * K ** -0.5; b = torch.randn(N, device=dev)
        y = torch.empty(M, N, device=dev)
        pl = ops._planes(w, 4, N, K); xa = ops._amax(x)
        f = lambda: lib.ttts_linear_fwd_h3(_p(x), _p(pl), _p(b), None, _p(y), M, N, K, 0, 0.0, 0, None, 0, 0, _p(xa), None, _stream())
        for _ in range(3): assert f() == 0
        torch.cuda.synchronize()
        import time
        tot = 0.0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(20):                       # idle gaps: the unthrottled clock, as inside the step
            time.sleep(0.002)
            e0.record(); f(); e1.record(); torch.cuda.synchronize()
            tot += e0.elapsed_time(e1) * 1e3
        out.append(f"M={M}: {tot / 20:6.1f} us ({M / 256 * ((N + 255) // 256) / 256:.2f} rounds)")
    print(f"N={N} K={K}: " + " | ".join(out))
