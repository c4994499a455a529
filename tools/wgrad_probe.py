#!/usr/bin/env python3
"""Launch the weight-gradient GEMM on one shape (target of rocprofv3 --pmc passes): M N K form(h3|x6) [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformertts_amd import _lib, ops
from transformertts_amd.ops import _p, _stream
M, N, K = (int(a) for a in sys.argv[1:4]); form = sys.argv[4]; reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
lib = _lib.load(); dev = torch.device("cuda:0")
x = torch.randn(M, K, device=dev); dy = torch.randn(M, N, device=dev) * 1e-6
dw = torch.empty(N, K, device=dev); db = torch.empty(N, device=dev)
ws = torch.empty(lib.ttts_wgrad_workspace_bytes(M, N, K, 1) // 4, device=dev)
am, xm = ops._amax(dy), ops._amax(x)
for _ in range(reps):
    if form == "h3":
        lib.ttts_linear_bwd_weight_h3(_p(dy), _p(x), _p(dw), _p(db), _p(ws), ws.numel() * 4, M, N, K, 0, 0, 0, _p(am), _p(xm), None, _stream())
    else:
        lib.ttts_linear_bwd_weight_x6(_p(dy), _p(x), _p(dw), _p(db), _p(ws), ws.numel() * 4, M, N, K, 0, 0, 0, None, _stream())
torch.cuda.synchronize()
