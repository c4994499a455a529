#!/usr/bin/env python3
"""Timing of the fp16x3 forward / data-gradient GEMMs with the epilogues the training step uses (relu + dropout + published
maxima, residual + dropout, relu gate, residual), interleaved rounds, HIP events.  A/B two builds with TTTS_LIB=<lib>."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformertts_amd import _lib, ops
from transformertts_amd.ops import _p, _stream

lib = _lib.load()
dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 55680


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def amax_of(x):
    out = torch.zeros(ops.AMAX_SLOTS, device=dev)
    _lib.check(lib.ttts_amax_partials(_p(x), x.numel(), _p(out), _stream()), "amax")
    return out


d, f = 256, 1024
x = torch.randn(M, d, device=dev); h = torch.relu(torch.randn(M, f, device=dev)); skip = torch.randn(M, d, device=dev)
w1 = torch.randn(f, d, device=dev) * d ** -0.5; w2 = torch.randn(d, f, device=dev) * f ** -0.5; wq = torch.randn(3 * d, d, device=dev) * d ** -0.5
b1 = torch.randn(f, device=dev); b2 = torch.randn(d, device=dev); bq = torch.randn(3 * d, device=dev)
dy = torch.randn(M, d, device=dev) * 1e-5; dh = torch.randn(M, f, device=dev) * 1e-5
yf = torch.empty(M, f, device=dev); yd = torch.empty(M, d, device=dev); yq = torch.empty(M, 3 * d, device=dev)
xa, ha, dya, dha = amax_of(x), amax_of(h), amax_of(dy), amax_of(dh)
am = torch.zeros(ops.AMAX_SLOTS, device=dev)
p1, p2, pq = ops._planes(w1, 4, f, d).clone(), ops._planes(w2, 4, d, f).clone(), ops._planes(wq, 4, 3 * d, d).clone()
p1t, p2t = ops._planes(w1, 5, d, f).clone(), ops._planes(w2, 5, f, d).clone()
cases = [
    ("ffn1 fwd  relu+drop+amax  N=1024 K=256", 2.0 * M * f * d,
     lambda: lib.ttts_linear_fwd_h3(_p(x), _p(p1), _p(b1), None, _p(yf), M, f, d, 1, 0.1, 77, None, 0, 0, _p(xa), _p(am), _stream())),
    ("ffn2 fwd  res+drop        N=256 K=1024", 2.0 * M * f * d,
     lambda: lib.ttts_linear_fwd_h3(_p(h), _p(p2), _p(b2), _p(skip), _p(yd), M, d, f, 0, 0.1, 78, None, 0, 0, _p(ha), None, _stream())),
    ("inproj fwd bias+amax      N=768 K=256", 2.0 * M * 3 * d * d,
     lambda: lib.ttts_linear_fwd_h3(_p(x), _p(pq), _p(bq), None, _p(yq), M, 3 * d, d, 0, 0.0, 0, None, 0, 0, _p(xa), _p(am), _stream())),
    ("ffn2 dgrad gate+amax      N=1024 K=256", 2.0 * M * f * d,
     lambda: lib.ttts_linear_bwd_data_h3(_p(dy), _p(p2t), None, _p(yf), M, d, f, _p(h), 1.0 / 0.9, _p(dya), _p(am), _stream())),
    ("ffn1 dgrad residual       N=256 K=1024", 2.0 * M * f * d,
     lambda: lib.ttts_linear_bwd_data_h3(_p(dh), _p(p1t), _p(skip), _p(yd), M, f, d, None, 1.0, _p(dha), None, _stream())),
]
res = {}
for rnd in range(3):
    for name, fl, fn in cases:
        res.setdefault(name, []).append(timeit(fn))
for name, fl, fn in cases:
    print(f"{name:42s} {min(res[name]):7.1f} us {fl / min(res[name]) / 1e6:6.1f} TF", flush=True)
