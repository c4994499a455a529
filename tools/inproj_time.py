#!/usr/bin/env python3
"""In-projection shapes of the step: the head-image GEMM (ttts_linear_fwd_h3d_img) against the fp32-output kernels the dispatch
would pick (ttts_linear_fwd_h3 / _h3d), back-to-back launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformertts_amd import _lib, ops
from transformertts_amd.ops import _p, _stream
lib = _lib.load(); dev = torch.device("cuda:0")
def timed(f, reps=30):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for M, N, K, what in ((55680, 768, 256, "decoder self in-proj"), (55680, 256, 256, "cross q-proj"), (6400, 768, 256, "encoder in-proj"),
                      (6400, 512, 256, "cross kv-proj"), (27840, 1536, 512, "scaled decoder in-proj")):
    x, w, b = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev) * K ** -0.5, torch.randn(N, device=dev)
    xa = ops._amax(x)
    p4, p8 = ops._planes(w, 4, N, K).clone(), ops._planes(w, 8, N, K).clone()
    y = torch.empty(M, N, device=dev); inv = torch.empty(N // 64, M, device=dev); am = torch.zeros(3, 1024, device=dev)
    t_img = timed(lambda: lib.ttts_linear_fwd_h3d_img(_p(x), _p(p8), _p(b), _p(y), _p(inv), M, N, K, _p(xa), _p(am), 0, _stream()))
    t_h3 = timed(lambda: lib.ttts_linear_fwd_h3(_p(x), _p(p4), _p(b), None, _p(y), M, N, K, 0, 0.0, 0, None, 0, 0, _p(xa), _p(am), _stream()))
    t_h3d = timed(lambda: lib.ttts_linear_fwd_h3d(_p(x), _p(p8), _p(b), None, _p(y), M, N, K, 0, 0.0, 0, None, _p(xa), _p(am), _stream()))
    print(f"{what:24s} M={M:6d} N={N:5d} K={K:4d}: image {t_img:6.1f} us   fp32 gemm_h3 {t_h3:6.1f} us   fp32 by DMA (h3d) {t_h3d:6.1f} us")
