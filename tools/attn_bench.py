#!/usr/bin/env python3
"""Self- / cross-attention forward and backward on the step's shapes through the ops layer (fp16x3 kernels), per-kernel
times from HIP events around the forward and the backward call, and a check against fp64 torch (development aid).
A/B two builds with TTTS_LIB=<lib>."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformertts_amd import ops
dev = torch.device("cuda:0")


def ev():
    return torch.cuda.Event(enable_timing=True)


def run(name, B, H, Tq, Tk, causal, p=0.1, reps=8):
    d = H * 64
    torch.manual_seed(0)
    lens = torch.full((B,), Tk, dtype=torch.int64, device=dev)
    if Tq == Tk:
        x = torch.randn(B, Tq, 3 * d, device=dev, requires_grad=True)
        x._ttts_amax = None
        fwd = lambda: ops.self_attention(x, lens, H, bool(causal), p, 7)
    else:
        q = torch.randn(B, Tq, d, device=dev, requires_grad=True)
        kv = torch.randn(B, Tk, 2 * d, device=dev, requires_grad=True)
        fwd = lambda: ops.cross_attention(q, kv, lens, H, p, 7, False)[0]
    do = torch.randn(B, Tq, d, device=dev)
    tf, tb = [], []
    for i in range(reps + 2):
        e0, e1, e2 = ev(), ev(), ev()
        e0.record()
        o = fwd()
        e1.record()
        o.backward(do)
        e2.record()
        torch.cuda.synchronize()
        if i >= 2:
            tf.append(e0.elapsed_time(e1) * 1e3); tb.append(e1.elapsed_time(e2) * 1e3)
    # fp64 check of the forward without dropout on a few batch entries
    if Tq == Tk:
        with torch.no_grad():
            o0 = ops.self_attention(x.detach(), lens, H, bool(causal), 0.0, 0)
            xb = x.detach()[:2].double().view(2, Tq, 3, H, 64)
            qq, kk, vv = (xb[:, :, i].transpose(1, 2) for i in range(3))
            s = qq @ kk.transpose(-1, -2) / 8.0
            if causal:
                s = s + torch.full((Tq, Tq), float("-inf"), device=dev, dtype=torch.float64).triu(1)
            ref = (torch.softmax(s, -1) @ vv).transpose(1, 2).reshape(2, Tq, d)
            err = float((o0[:2].double() - ref).norm() / ref.norm())
    else:
        err = float("nan")
    print(f"{name:10s} fwd {min(tf):7.1f} us  bwd {min(tb):7.1f} us   fwd rel err vs fp64 {err:.1e}", flush=True)


run("dec self", 64, 4, 870, 870, 1)
run("enc self", 64, 4, 100, 100, 0)
run("cross", 64, 4, 870, 100, 0)
