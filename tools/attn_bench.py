#!/usr/bin/env python3
"""Attention kernels, fp32 MFMA vs split-precision forms, on the step's shapes (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformertts_amd import _lib
from transformertts_amd.ops import _p, _stream, _off
lib = _lib.load(); dev = torch.device("cuda:0")


def timeit(fn, n=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def run(name, B, H, Tq, Tk, causal, weights, p=0.1):
    d = H * 64
    q = torch.randn(B, Tq, d, device=dev); kv = torch.randn(B, Tk, 2 * d, device=dev)
    o = torch.empty(B, Tq, d, device=dev); do = torch.randn(B, Tq, d, device=dev)
    lse = torch.empty(B, H, Tq, device=dev); delta = torch.empty_like(lse)
    dq = torch.empty_like(q); dkv = torch.empty_like(kv)
    attn = torch.empty(B, H, Tq, Tk, device=dev) if weights else None
    lens = torch.full((B,), Tk, dtype=torch.int64, device=dev)
    out = []
    res = {}
    for suffix in ("", "_x6"):
        f = getattr(lib, "ttts_attention_fwd" + suffix, None)
        if f is None:
            continue
        us = timeit(lambda: f(_p(q), _off(kv, 0), _off(kv, d), _p(o), _p(lse), _p(attn), _p(lens), B, H, Tq, Tk, d, 2 * d, 2 * d, d,
                              causal, p, 7, None, _stream()))
        res[suffix] = (o.clone(), lse.clone(), attn.clone() if weights else None)
        out.append(f"fwd{suffix} {us:7.1f}us")
        g = getattr(lib, "ttts_attention_bwd" + suffix, None)
        if g is not None:
            us = timeit(lambda: g(_p(q), _off(kv, 0), _off(kv, d), _p(o), _p(do), _p(lse), _p(delta), _p(dq), _off(dkv, 0), _off(dkv, d),
                                  _p(lens), B, H, Tq, Tk, d, 2 * d, 2 * d, d, d, 2 * d, 2 * d, causal, p, 7, None, _stream()))
            res["b" + suffix] = (dq.clone(), dkv.clone())
            out.append(f"bwd{suffix} {us:7.1f}us")
    def rel(a, b): return float((a - b).norm() / b.norm())
    if "" in res and "_x6" in res:
        out.append(f"o diff {rel(res['_x6'][0], res[''][0]):.1e} lse {rel(res['_x6'][1], res[''][1]):.1e}")
        if weights: out.append(f"attn diff {rel(res['_x6'][2], res[''][2]):.1e}")
    if "b" in res and "b_x6" in res:
        out.append(f"dq diff {rel(res['b_x6'][0], res['b'][0]):.1e} dkv {rel(res['b_x6'][1], res['b'][1]):.1e}")
    print(name.ljust(12), " | ".join(out))


run("dec self", 64, 4, 870, 870, 1, False)
run("enc self", 64, 4, 100, 100, 0, False)
run("cross", 64, 4, 870, 100, 0, True)
run("cross nw", 64, 4, 870, 100, 0, False)
