#!/usr/bin/env python3
"""Weight-gradient kernels (fp32 MFMA | bf16x6 | fp16x3, each including its split-K reduction) on the step's shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformertts_amd import _lib
from transformertts_amd.ops import _p, _stream
lib = _lib.load(); dev = torch.device("cuda:0")


def timeit(fn, n=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def lin(M, N, K):
    dy = torch.randn(M, N, device=dev); x = torch.randn(M, K, device=dev)
    dw = torch.empty(N, K, device=dev); db = torch.empty(N, device=dev)
    ws = torch.empty(lib.ttts_wgrad_workspace_bytes(M, N, K, 1) // 4, device=dev)
    out = []
    am, xm = torch.empty(1024, device=dev), torch.empty(1024, device=dev)
    lib.ttts_amax_partials(_p(dy), dy.numel(), _p(am), _stream())
    lib.ttts_amax_partials(_p(x), x.numel(), _p(xm), _stream())
    for name in ("ttts_linear_bwd_weight", "ttts_linear_bwd_weight_x6", "ttts_linear_bwd_weight_h3"):
        f = getattr(lib, name)
        tail = (_p(am), _p(xm), None, _stream()) if name.endswith("h3") else (None, _stream())
        us = timeit(lambda: f(_p(dy), _p(x), _p(dw), _p(db), _p(ws), ws.numel() * 4, M, N, K, 0, 0, 0, *tail))
        out.append(f"{us:7.1f}us {2.0 * M * N * K / us / 1e6:6.1f}TF")
    print(f"linear M={M} N={N} K={K}:".ljust(36), " | ".join(out))


def conv(B, T, cin, cout):
    M = B * T
    dy = torch.randn(B, T, cout, device=dev); x = torch.randn(B, T, cin, device=dev)
    dw = torch.empty(cout, cin, 5, device=dev); db = torch.empty(cout, device=dev)
    ws = torch.empty(lib.ttts_wgrad_workspace_bytes(M, cout, cin, 5) // 4, device=dev)
    out = []
    am, xm = torch.empty(1024, device=dev), torch.empty(1024, device=dev)
    lib.ttts_amax_partials(_p(dy), dy.numel(), _p(am), _stream())
    lib.ttts_amax_partials(_p(x), x.numel(), _p(xm), _stream())
    for name in ("ttts_conv1d_bwd_weight", "ttts_conv1d_bwd_weight_x6", "ttts_conv1d_bwd_weight_h3"):
        f = getattr(lib, name)
        tail = (_p(am), _p(xm), None, _stream()) if name.endswith("h3") else (None, _stream())
        us = timeit(lambda: f(_p(dy), _p(x), _p(dw), _p(db), _p(ws), ws.numel() * 4, B, T, cin, cout, 5, 0, *tail))
        out.append(f"{us:7.1f}us {2.0 * M * cin * cout * 5 / us / 1e6:6.1f}TF")
    print(f"conv B={B} T={T} {cin}->{cout}:".ljust(36), " | ".join(out))


for (M, N, K) in [(55680, 256, 256), (55680, 768, 256), (55680, 1024, 256), (55680, 256, 1024), (55680, 512, 256),
                  (55680, 256, 80), (55680, 80, 256), (6400, 768, 256), (6400, 1024, 256), (6400, 256, 1024)]:
    lin(M, N, K)
for (B, T, cin, cout) in [(64, 870, 256, 256), (64, 870, 80, 256), (64, 870, 256, 80), (64, 100, 256, 256), (32, 870, 512, 512)]:
    conv(B, T, cin, cout)
