#!/usr/bin/env python3
"""One shape on gemm_h3i_kernel, a few launches (for rocprofv3 --pmc passes: tools/pmc_h3i.sh).
usage: h3i_probe.py M N K reps [image|raw]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformertts_amd import _lib, ops
from transformertts_amd.ops import _p, _stream
lib = _lib.load(); dev = torch.device("cuda:0")
M, N, K, reps = (int(a) for a in sys.argv[1:5])
mode = sys.argv[5] if len(sys.argv) > 5 else "image"
x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * K ** -0.5; b = torch.randn(N, device=dev)
y = torch.empty(M, N, device=dev)
pl = ops._planes(w, 8, N, K).clone()
if mode == "image":
    img = torch.empty(M * K * 2, dtype=torch.int16, device=dev); inv = torch.empty(M, device=dev)
    lib.ttts_act_image(_p(x), _p(img), _p(inv), M, K, _stream())
    run = lambda: lib.ttts_linear_fwd_h3i(_p(img), _p(inv), _p(pl), _p(b), None, _p(y), M, N, K, 0, 0.0, 0, None, None, _stream())
else:
    xa = torch.zeros(ops.AMAX_SLOTS, device=dev)
    lib.ttts_amax_partials(_p(x), x.numel(), _p(xa), _stream())
    run = lambda: lib.ttts_linear_fwd_h3d(_p(x), _p(pl), _p(b), None, _p(y), M, N, K, 0, 0.0, 0, None, _p(xa), None, _stream())
for _ in range(reps):
    assert run() == 0
torch.cuda.synchronize()
