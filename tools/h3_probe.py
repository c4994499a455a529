#!/usr/bin/env python3
"""Launch the fp16x3 forward GEMM on one shape a few times (target of the rocprofv3 --pmc passes): M N K [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformertts_amd import _lib, ops
from transformertts_amd.ops import _p, _stream
M, N, K = (int(a) for a in sys.argv[1:4]); reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
lib = _lib.load(); dev = torch.device("cuda:0")
x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * K ** -0.5; b = torch.randn(N, device=dev)
y = torch.empty(M, N, device=dev); pl = ops._planes(w, 4, N, K)
xa = torch.zeros(ops.AMAX_SLOTS, device=dev)
lib.ttts_amax_partials(_p(x), x.numel(), _p(xa), _stream())
for _ in range(reps):
    lib.ttts_linear_fwd_h3(_p(x), _p(pl), _p(b), None, _p(y), M, N, K, 0, 0.0, 0, None, 0, 0, _p(xa), None, _stream())
torch.cuda.synchronize()
