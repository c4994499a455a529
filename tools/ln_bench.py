#!/usr/bin/env python3
"""LayerNorm forward / backward timing at the step's shape (55 680 x 256) and the scaled one (27 840 x 512)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformertts_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, reps=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for M, d in ((55680, 256), (27840, 512), (6400, 256)):
    x = torch.randn(M, d, device=dev, requires_grad=True); g = torch.randn(d, device=dev, requires_grad=True); b = torch.randn(d, device=dev, requires_grad=True)
    dy = torch.randn(M, d, device=dev)
    with torch.no_grad():
        f = min(timeit(lambda: ops.layer_norm(x, g, b)) for _ in range(3))
    y = ops.layer_norm(x, g, b)
    bw = min(timeit(lambda: torch.autograd.grad(y, (x, g, b), dy, retain_graph=True)) for _ in range(3))
    print(f"M={M} d={d}: fwd {f:6.1f} us ({2*M*d*4/f/1e6:.2f} TB/s)  bwd {bw:6.1f} us ({3*M*d*4/bw/1e6:.2f} TB/s incl. reductions)")
