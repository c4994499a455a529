#!/usr/bin/env python3
"""fp16x3 forward GEMM on the encoder-side (M = 6400) shapes per tile choice."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformertts_amd import _lib, ops
from transformertts_amd.ops import _p, _stream
lib = _lib.load(); dev = torch.device("cuda:0")
def timeit(fn, reps=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for M, N, K in ((6400, 256, 256), (6400, 768, 256), (6400, 1024, 256), (6400, 256, 1024), (6400, 512, 256), (13920, 256, 256), (13920, 1024, 256)):
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * K ** -0.5; b = torch.randn(N, device=dev)
    y = torch.empty(M, N, device=dev); pl = ops._planes(w, 4, N, K).clone()
    us = min(timeit(lambda: lib.ttts_linear_fwd_h3(_p(x), _p(pl), _p(b), None, _p(y), M, N, K, 0, 0.0, 0, None, 0, 0, _stream())) for _ in range(3))
    print(f"M={M} N={N:5d} K={K:5d}: {us:7.1f} us  {2.0*M*N*K/us/1e6:6.1f} TF", flush=True)
