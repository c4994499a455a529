#!/usr/bin/env python3
"""Time the fp16x3 weight gradient (kernel + split reduction) of the library in TTTS_LIB on a few shapes: [M N K ...] triples,
default the step's three 256-tile linears.  One line per shape; used for same-box A/B of build variants (tools/build_variant.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformertts_amd import _lib, ops
from transformertts_amd.ops import _p, _stream
lib = _lib.load(); dev = torch.device("cuda:0")
args = [int(a) for a in sys.argv[1:]] or [55680, 1024, 256, 55680, 256, 1024, 55680, 768, 256]
out = []
for M, N, K in zip(args[0::3], args[1::3], args[2::3]):
    x = torch.randn(M, K, device=dev); dy = torch.randn(M, N, device=dev) * 1e-6
    dw = torch.empty(N, K, device=dev); db = torch.empty(N, device=dev)
    ws = torch.empty(lib.ttts_wgrad_workspace_bytes(M, N, K, 1) // 4, device=dev)
    am, xm = ops._amax(dy), ops._amax(x)
    f = lambda: lib.ttts_linear_bwd_weight_h3(_p(dy), _p(x), _p(dw), _p(db), _p(ws), ws.numel() * 4, M, N, K, 0, 0, 0, _p(am), _p(xm), None, _stream())
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): f()
    e1.record(); torch.cuda.synchronize()
    out.append(f"{M}x{N}x{K}: {e0.elapsed_time(e1) / 50 * 1e3:6.1f} us")
print(os.environ.get("TTTS_LIB", "default").split("/")[-1].ljust(14), " | ".join(out))
