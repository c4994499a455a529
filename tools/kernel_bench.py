#!/usr/bin/env python3
"""Per-kernel micro-benchmarks on the shapes of the B=64 training step (development aid).
HIP-event timing on torch's current stream, random data, 20 launches after 3 warm-ups."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from transformertts_amd import _lib, ops
from transformertts_amd.ops import _p, _stream, _off, _ws

dev = torch.device("cuda:0")
lib = _lib.load()


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3   # us


def gemm_fwd(M, N, K, act=0, res=False):
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * K ** -0.5; b = torch.randn(N, device=dev)
    y = torch.empty(M, N, device=dev); r = torch.randn(M, N, device=dev) if res else None
    us = timeit(lambda: lib.ttts_linear_fwd(_p(x), _p(w), _p(b), _p(r), _p(y), M, N, K, act, 0.0, 0, None, 0, 0, _stream()))
    print(f"linear_fwd   M={M:6d} N={N:5d} K={K:5d}: {us:8.1f} us  {2.0*M*N*K/us/1e6:7.1f} TF/s")


def gemm_fwd_x6(M, N, K):
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * K ** -0.5; b = torch.randn(N, device=dev)
    y = torch.empty(M, N, device=dev)
    pl = ops._planes(w, 0, N, K)
    us = timeit(lambda: lib.ttts_linear_fwd_x6(_p(x), _p(pl), _p(b), None, _p(y), M, N, K, 0, 0.0, 0, None, 0, 0, _stream()))
    ref = (x[:256].double() @ w.double().t() + b.double())
    err = ((y[:256].double() - ref).norm() / ref.norm()).item()
    print(f"linear_fwd_x6 M={M:6d} N={N:5d} K={K:5d}: {us:8.1f} us  {2.0*M*N*K/us/1e6:7.1f} TF/s   rel err vs fp64 {err:.2e}")


def conv_x6(B, T, cin, cout):
    x = torch.randn(B, T, cin, device=dev); w = torch.randn(cout, cin, 5, device=dev) * (5 * cin) ** -0.5; b = torch.randn(cout, device=dev)
    y = torch.empty(B, T, cout, device=dev); dx = torch.empty_like(x)
    pf = ops._planes(w, 2, cout, 5 * cin, cin, 5); pb = ops._planes(w, 3, cin, 5 * cout, cout, 5)
    fl = 2.0 * B * T * cin * cout * 5
    us = timeit(lambda: lib.ttts_conv1d_fwd_x6(_p(x), _p(pf), _p(b), _p(y), B, T, cin, cout, 5, _stream()))
    print(f"conv_fwd_x6   {cin:4d}->{cout:4d} M={B*T}: {us:8.1f} us {fl/us/1e6:7.1f} TF/s")
    us = timeit(lambda: lib.ttts_conv1d_bwd_data_x6(_p(y), _p(pb), _p(dx), B, T, cin, cout, 5, _stream()))
    print(f"conv_dgrad_x6 {cin:4d}->{cout:4d} M={B*T}: {us:8.1f} us {fl/us/1e6:7.1f} TF/s")


def gemm_dgrad(M, N, K):
    dy = torch.randn(M, N, device=dev); w = torch.randn(N, K, device=dev); dx = torch.empty(M, K, device=dev)
    us = timeit(lambda: lib.ttts_linear_bwd_data(_p(dy), _p(w), None, _p(dx), M, N, K, None, 1.0, _stream()))
    print(f"linear_dgrad M={M:6d} N={N:5d} K={K:5d}: {us:8.1f} us  {2.0*M*N*K/us/1e6:7.1f} TF/s")


def gemm_wgrad(M, N, K):
    dy = torch.randn(M, N, device=dev); x = torch.randn(M, K, device=dev); dw = torch.empty(N, K, device=dev)
    db = torch.empty(N, device=dev)
    ws = _ws(lib.ttts_wgrad_workspace_bytes(M, N, K, 1), dev)
    us = timeit(lambda: lib.ttts_linear_bwd_weight(_p(dy), _p(x), _p(dw), _p(db), _p(ws), ws.numel() * 4, M, N, K, 0, 0, 0, _stream()))
    print(f"linear_wgrad M={M:6d} N={N:5d} K={K:5d}: {us:8.1f} us  {2.0*M*N*K/us/1e6:7.1f} TF/s (incl. reduce + bias)")


def conv(B, T, cin, cout):
    x = torch.randn(B, T, cin, device=dev); w = torch.randn(cout, cin, 5, device=dev); b = torch.randn(cout, device=dev)
    wf = torch.empty(cout * cin * 5, device=dev); wb = torch.empty(cout * cin * 5, device=dev)
    lib.ttts_conv1d_pack_weight(_p(w), _p(wf), _p(wb), cout, cin, 5, _stream())
    y = torch.empty(B, T, cout, device=dev); dx = torch.empty_like(x); dw = torch.empty_like(w); db = torch.empty(cout, device=dev)
    fl = 2.0 * B * T * cin * cout * 5
    us = timeit(lambda: lib.ttts_conv1d_fwd(_p(x), _p(wf), _p(b), _p(y), B, T, cin, cout, 5, _stream()))
    print(f"conv_fwd   {cin:4d}->{cout:4d} M={B*T}: {us:8.1f} us {fl/us/1e6:7.1f} TF/s")
    us = timeit(lambda: lib.ttts_conv1d_bwd_data(_p(y), _p(wb), _p(dx), B, T, cin, cout, 5, _stream()))
    print(f"conv_dgrad {cin:4d}->{cout:4d} M={B*T}: {us:8.1f} us {fl/us/1e6:7.1f} TF/s")
    ws = _ws(lib.ttts_wgrad_workspace_bytes(B * T, cout, cin, 5), dev)
    us = timeit(lambda: lib.ttts_conv1d_bwd_weight(_p(y), _p(x), _p(dw), _p(db), _p(ws), ws.numel() * 4, B, T, cin, cout, 5, 0, _stream()))
    print(f"conv_wgrad {cin:4d}->{cout:4d} M={B*T}: {us:8.1f} us {fl/us/1e6:7.1f} TF/s (incl. reduce + bias)")


def attn(B, H, Tq, Tk, causal, cross, p=0.1):
    d = H * 64
    lens = torch.full((B,), Tk, dtype=torch.int64, device=dev)
    if cross:
        q = torch.randn(B, Tq, d, device=dev); kv = torch.randn(B, Tk, 2 * d, device=dev)
        qa, ka, va, ldq, ldk = _off(q, 0), _off(kv, 0), _off(kv, d), d, 2 * d
        dqb, dkvb = torch.empty_like(q), torch.empty_like(kv)
        dqa, dka, dva, lddq, lddk = _off(dqb, 0), _off(dkvb, 0), _off(dkvb, d), d, 2 * d
    else:
        qkv = torch.randn(B, Tq, 3 * d, device=dev)
        qa, ka, va, ldq, ldk = _off(qkv, 0), _off(qkv, d), _off(qkv, 2 * d), 3 * d, 3 * d
        dqkv = torch.empty_like(qkv)
        dqa, dka, dva, lddq, lddk = _off(dqkv, 0), _off(dqkv, d), _off(dqkv, 2 * d), 3 * d, 3 * d
    o = torch.empty(B, Tq, d, device=dev); lse = torch.empty(B, H, Tq, device=dev); delta = torch.empty_like(lse)
    a = torch.empty(B, H, Tq, Tk, device=dev) if cross else None
    do = torch.randn(B, Tq, d, device=dev)
    fl = 4.0 * B * H * Tq * Tk * 64 * (0.5 if causal else 1.0)
    us = timeit(lambda: lib.ttts_attention_fwd(qa, ka, va, _p(o), _p(lse), _p(a), _p(lens), B, H, Tq, Tk, ldq, ldk, ldk, d,
                                               int(causal), p, 7, None, _stream()))
    print(f"attn_fwd B={B} H={H} Tq={Tq} Tk={Tk} causal={causal} cross={cross}: {us:8.1f} us {fl/us/1e6:6.1f} TF/s (algorithmic)")
    us = timeit(lambda: lib.ttts_attention_bwd(qa, ka, va, _p(o), _p(do), _p(lse), _p(delta), dqa, dka, dva, _p(lens), B, H, Tq,
                                               Tk, ldq, ldk, ldk, d, lddq, lddk, lddk, int(causal), p, 7, None, _stream()))
    print(f"attn_bwd (dq + dkv)                                       : {us:8.1f} us {2.5*fl/us/1e6:6.1f} TF/s (2.5x fwd flops)")


if __name__ == "__main__":
    Mm, Mp = 64 * 870, 64 * 100
    for (M, N, K) in [(Mm, 256, 256), (Mm, 768, 256), (Mm, 1024, 256), (Mm, 256, 1024), (Mm, 256, 80), (Mm, 80, 256),
                      (Mp, 256, 256), (Mp, 768, 256), (Mp, 1024, 256), (Mp, 256, 1024), (Mp, 512, 256)]:
        gemm_fwd(M, N, K)
    for (M, N, K) in [(Mm, 256, 256), (Mm, 768, 256), (Mm, 1024, 256), (Mm, 256, 1024), (Mm, 256, 80), (Mm, 80, 256),
                      (Mp, 256, 256), (Mp, 1024, 256), (Mp, 256, 1024)]:
        gemm_fwd_x6(M, N, K)
    conv_x6(64, 870, 256, 256)
    conv_x6(64, 870, 80, 256)
    conv_x6(64, 870, 256, 80)
    conv_x6(64, 100, 256, 256)
    for (M, N, K) in [(Mm, 256, 256), (Mm, 768, 256), (Mm, 1024, 256), (Mm, 256, 1024), (Mm, 80, 256)]:
        gemm_dgrad(M, N, K)
    for (M, N, K) in [(Mm, 256, 256), (Mm, 768, 256), (Mm, 1024, 256), (Mm, 256, 1024), (Mm, 80, 256), (Mm, 256, 80), (Mp, 1024, 256)]:
        gemm_wgrad(M, N, K)
    conv(64, 870, 256, 256)
    conv(64, 870, 80, 256)
    conv(64, 870, 256, 80)
    conv(64, 100, 256, 256)
    attn(64, 4, 870, 870, True, False)
    attn(64, 4, 870, 100, False, True)
    attn(64, 4, 100, 100, False, False)
