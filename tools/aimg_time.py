#!/usr/bin/env python3
"""Per-launch times of the head-image attention kernels at the BASELINE shapes (causal self-attention 64 x 4 x 870, cross
870 x 100 with weights, encoder 100 x 100): HIP events around back-to-back launches; run under
`rocprofv3 --kernel-trace --stats` for the per-kernel split of the backward.  usage: python3 tools/aimg_time.py [p_drop] [reps]   (under the profiler: `rocprofv3 --kernel-trace --stats -- python3 tools/aimg_time.py ...` -- the interpreter itself behind `--`, never the script through its env shebang: that is an exec hop behind the profiler's preloaded library, which this pool forbids)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformertts_amd import _lib, ops
from transformertts_amd.ops import _p, _off, _stream

lib = _lib.load()
dev = torch.device("cuda:0")
p_drop = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
B, H = 64, 4
d = H * 64


def himg(x2d):
    M, N = x2d.shape
    img, inv = torch.empty_like(x2d), torch.empty(N // 64, M, device=dev)
    _lib.check(lib.ttts_head_image(_p(x2d), N, _p(img), N, _p(inv), M, N, _stream()), "head_image")
    return img, inv


def timed(f):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for name, causal, Tq, Tk, attn_out in (("decoder self (causal)", 1, 870, 870, False), ("cross, weights written", 0, 870, 100, True),
                                        ("cross, no weights", 0, 870, 100, False), ("encoder self", 0, 100, 100, False)):
    q, kv, do = torch.randn(B * Tq, d, device=dev), torch.randn(B * Tk, 2 * d, device=dev), torch.randn(B, Tq, d, device=dev) * 1e-5
    qi, qinv = himg(q)
    kvi, kvinv = himg(kv)
    va = ops._amax(kv[:, d:].contiguous())
    lens = torch.full((B,), Tk, dtype=torch.int64, device=dev)
    o = torch.empty(B, Tq, d, device=dev); stat = torch.empty(6, B, H, Tq, device=dev)
    attn = torch.empty(B, H, Tq, Tk, device=dev) if attn_out else None
    dq, dkv, delta = torch.empty(B, Tq, d, device=dev), torch.empty(B, Tk, 2 * d, device=dev), torch.empty(B, H, Tq, device=dev)
    doa = ops._amax(do)
    nsp = 1 if causal else ops._dkv_query_splits(B * H * -(-Tk // 128), Tq)
    part = torch.empty(nsp, B, Tk, 2 * d, device=dev) if nsp > 1 else None
    HK = H * B * Tk
    fwd = lambda: _lib.check(lib.ttts_attention_fwd_img(_p(qi), _off(kvi, 0), _off(kvi, d), _p(qinv), _off(kvinv, 0), _off(kvinv, HK), _p(o),
                                                        _p(stat[0]), _p(attn), _p(lens), B, H, Tq, Tk, d, 2 * d, 2 * d, d, causal, 0.125, p_drop, 7,
                                                        None, _p(va), None, _p(stat[1:]), 0, 0, 0, _stream()), "fwd")
    bwd = lambda: _lib.check(lib.ttts_attention_bwd_img(_p(qi), _off(kvi, 0), _off(kvi, d), _p(qinv), _off(kvinv, 0), _off(kvinv, HK), _p(o),
                                                        _p(do), _p(stat[1:]), _p(delta), _p(dq), _off(dkv, 0), _off(dkv, d), _p(lens), B, H, Tq,
                                                        Tk, d, 2 * d, 2 * d, d, d, 2 * d, 2 * d, causal, 0.125, p_drop, 7, None, _p(doa), None,
                                                        None, _p(part), nsp, 0, 0, 0, _stream()), "bwd")
    tf = timed(fwd)
    tb = timed(bwd)
    print(f"{name:26s} fwd {tf:7.1f} us   bwd (dq + dkv) {tb:7.1f} us   [p_drop {p_drop}]")
