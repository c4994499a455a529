#!/usr/bin/env python3
"""Launch one GEMM shape a few times (for rocprofv3 --pmc runs).  usage: gemm_probe.py M N K [reps] [x6=1]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformertts_amd import _lib, ops
from transformertts_amd.ops import _p, _stream
M, N, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
x6 = (sys.argv[5] if len(sys.argv) > 5 else "1") == "1"
lib = _lib.load()
dev = torch.device("cuda:0")
x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev); y = torch.empty(M, N, device=dev)
pl = ops._planes(w, 0, N, K)
for _ in range(reps):
    if x6:
        lib.ttts_linear_fwd_x6(_p(x), _p(pl), _p(b), None, _p(y), M, N, K, 0, 0.0, 0, None, 0, 0, _stream())
    else:
        lib.ttts_linear_fwd(_p(x), _p(w), _p(b), None, _p(y), M, N, K, 0, 0.0, 0, None, 0, 0, _stream())
torch.cuda.synchronize()
