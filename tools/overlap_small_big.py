#!/usr/bin/env python3
"""Do the small encoder-side GEMMs (M = 6400) hide beside the large decoder-side ones (M = 55680) when the two chains run
on two streams?  Times 4 big + 24 small fp16x3 linear launches back to back on one stream against the same launches split
over two streams (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformertts_amd import _lib, ops
from transformertts_amd.ops import _p

lib = _lib.load()
dev = torch.device("cuda:0")


def mk(M, N, K):
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * K ** -0.5; b = torch.randn(N, device=dev)
    y = torch.empty(M, N, device=dev); pl = ops._planes(w, 4, N, K)
    xa = torch.zeros(ops.AMAX_SLOTS, device=dev)
    lib.ttts_amax_partials(_p(x), x.numel(), _p(xa), torch.cuda.current_stream().cuda_stream)
    return lambda st: lib.ttts_linear_fwd_h3(_p(x), _p(pl), _p(b), None, _p(y), M, N, K, 0, 0.0, 0, None, 0, 0, _p(xa), None, st)


big = [mk(55680, 768, 256), mk(55680, 256, 256), mk(55680, 1024, 256), mk(55680, 256, 1024)]
small = [mk(6400, 768, 256), mk(6400, 256, 256), mk(6400, 1024, 256), mk(6400, 256, 1024)] * 6
torch.cuda.synchronize()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def run(two):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s1.synchronize(); s2.synchronize()
    e0.record(s1)
    if two:
        s2.wait_event(e0)
    for f in big:
        f(s1.cuda_stream)
    for f in small:
        f((s2 if two else s1).cuda_stream)
    if two:
        s1.wait_stream(s2)
    e1.record(s1)
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3


for _ in range(3):
    run(False); run(True)
for i in range(4):
    print(f"one stream {run(False):8.1f} us   two streams {run(True):8.1f} us")
