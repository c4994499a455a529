#!/usr/bin/env python3
"""Quick correctness check of the fp16x3 forward GEMM against fp64 on the step's shapes (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformertts_amd import _lib, ops
from transformertts_amd.ops import _p, _stream
lib = _lib.load(); dev = torch.device("cuda:0")
torch.manual_seed(0)
for M, N, K in ((55680, 1024, 256), (55680, 256, 1024), (55680, 768, 256), (6400, 1024, 256), (1000, 256, 256), (27840, 2048, 512), (300, 512, 64)):
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * K ** -0.5; b = torch.randn(N, device=dev)
    y = torch.full((M, N), float("nan"), device=dev)
    xa = torch.zeros(ops.AMAX_SLOTS, device=dev)
    lib.ttts_amax_partials(_p(x), x.numel(), _p(xa), _stream())
    ya = torch.zeros(ops.AMAX_SLOTS, device=dev)
    pl = ops._planes(w, 4, N, K).clone()
    _lib.check(lib.ttts_linear_fwd_h3(_p(x), _p(pl), _p(b), None, _p(y), M, N, K, 1, 0.0, 0, None, 0, 0, _p(xa), _p(ya), _stream()), "fwd")
    rows = torch.randint(0, M, (512,), device=dev)
    ref = torch.relu(x[rows].double() @ w.double().t() + b.double())
    err = ((y[rows].double() - ref).norm() / ref.norm()).item()
    print(f"M={M} N={N} K={K}: rel err {err:.2e}  nan {int(torch.isnan(y).sum())}  amax {ya.max().item():.4f} vs {y.abs().max().item():.4f}", flush=True)
