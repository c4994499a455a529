#!/usr/bin/env python3
"""Per-shape timing of the forward GEMM forms (bf16x6, fp16x3) on the step's shapes, interleaved rounds in one process
(HIP events on the launch stream), random operands.  usage: tools/gemm_forms_bench.py [h3]   (h3: skip the bf16x6 form)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformertts_amd import _lib, ops
from transformertts_amd.ops import _p, _stream

lib = _lib.load()
dev = torch.device("cuda:0")
ONLY_H3 = len(sys.argv) > 1 and sys.argv[1] == "h3"
SHAPES = [("ffn1 256->1024", 55680, 1024, 256), ("ffn2 1024->256", 55680, 256, 1024), ("inproj 256->768", 55680, 768, 256),
          ("outproj 256->256", 55680, 256, 256), ("enc ffn1", 6400, 1024, 256), ("enc ffn2", 6400, 256, 1024), ("enc outproj", 6400, 256, 256), ("enc inproj", 6400, 768, 256), ("scaled ffn1 512->2048", 27840, 2048, 512),
          ("scaled ffn2 2048->512", 27840, 512, 2048)]
CONV = [("postnet conv 256->256 k5", 64, 870, 256, 256), ("enc conv 256->256 k5", 64, 100, 256, 256)]


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


for name, M, N, K in SHAPES:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * K ** -0.5; b = torch.randn(N, device=dev)
    y = torch.empty(M, N, device=dev)
    xa = amax_of(x)
    p6, p3 = ops._planes(w, 0, N, K).clone(), ops._planes(w, 4, N, K).clone()
    fl = 2.0 * M * N * K
    res = {}
    forms = [("h3", lambda: lib.ttts_linear_fwd_h3(_p(x), _p(p3), _p(b), None, _p(y), M, N, K, 0, 0.0, 0, None, 0, 0, _p(xa), None, _stream()))]
    if not ONLY_H3:
        forms.insert(0, ("x6", lambda: lib.ttts_linear_fwd_x6(_p(x), _p(p6), _p(b), None, _p(y), M, N, K, 0, 0.0, 0, None, 0, 0, _stream())))
    for rnd in range(3):
        for form, fn in forms:
            res.setdefault(form, []).append(timeit(fn))
    print(f"{name:28s} M={M} N={N} K={K}: " + "  ".join(f"{f} {min(v):7.1f} us {fl / min(v) / 1e6:6.1f} TF" for f, v in res.items()), flush=True)
for name, B, T, cin, cout in CONV:
    x = torch.randn(B, T, cin, device=dev); w = torch.randn(cout, cin, 5, device=dev) * (5 * cin) ** -0.5; b = torch.randn(cout, device=dev)
    y = torch.empty(B, T, cout, device=dev)
    xa = amax_of(x)
    p6, p3 = ops._planes(w, 2, cout, 5 * cin, cin, 5).clone(), ops._planes(w, 6, cout, 5 * cin, cin, 5).clone()
    fl = 2.0 * B * T * cout * cin * 5
    res = {}
    forms = [("h3", lambda: lib.ttts_conv1d_fwd_h3(_p(x), _p(p3), _p(b), _p(y), B, T, cin, cout, 5, _p(xa), None, _stream()))]
    if not ONLY_H3:
        forms.insert(0, ("x6", lambda: lib.ttts_conv1d_fwd_x6(_p(x), _p(p6), _p(b), _p(y), B, T, cin, cout, 5, _stream())))
    for rnd in range(3):
        for form, fn in forms:
            res.setdefault(form, []).append(timeit(fn))
    print(f"{name:28s} B={B} T={T}: " + "  ".join(f"{f} {min(v):7.1f} us {fl / min(v) / 1e6:6.1f} TF" for f, v in res.items()), flush=True)
