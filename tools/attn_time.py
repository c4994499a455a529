#!/usr/bin/env python3
"""HIP-event timing of the attention kernels through the ops layer on the step's shapes (development aid):
causal self-attention forward, forward + backward; cross-attention (Tk = 100) forward, forward + backward."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformertts_amd import ops
dev = torch.device("cuda:0")
B, H, T, TP = 64, 4, 870, 100
P = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
d = H * 64


def timeit(fn, reps=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


torch.manual_seed(0)
qkv = torch.randn(B, T, 3 * d, device=dev, requires_grad=True)
do = torch.randn(B, T, d, device=dev) * 1e-5
lens = torch.full((B,), T, dtype=torch.int64, device=dev)
q = torch.randn(B, T, d, device=dev, requires_grad=True)
kv = torch.randn(B, TP, 2 * d, device=dev, requires_grad=True)
plens = torch.full((B,), TP, dtype=torch.int64, device=dev)


def self_fwd():
    with torch.no_grad():
        ops.self_attention(qkv, lens, H, True, P, 7)


def self_fb():
    o = ops.self_attention(qkv, lens, H, True, P, 7)
    o.backward(do)
    qkv.grad = None


def cross_fwd():
    with torch.no_grad():
        ops.cross_attention(q, kv, plens, H, P, 9, need_weights=False)


def cross_fb():
    o = ops.cross_attention(q, kv, plens, H, P, 9, need_weights=False)
    o = o[0] if isinstance(o, tuple) else o
    o.backward(do)
    q.grad = None; kv.grad = None


for rnd in range(2):
    a, b, c, e = timeit(self_fwd), timeit(self_fb), timeit(cross_fwd), timeit(cross_fb)
    print(f"round {rnd}: causal self fwd {a:6.1f} us   fwd+bwd {b:6.1f} us (bwd {b - a:6.1f})   cross fwd {c:6.1f} us   fwd+bwd {e:6.1f} us (bwd {e - c:6.1f})",
          flush=True)
