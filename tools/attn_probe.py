#!/usr/bin/env python3
"""Launch the self-attention forward+backward a few times (for rocprofv3 --pmc runs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformertts_amd import _lib
from transformertts_amd.ops import _p, _stream, _off
lib = _lib.load(); dev = torch.device("cuda:0")
B, H, T = 64, 4, 870
d = H * 64
qkv = torch.randn(B, T, 3 * d, device=dev); dqkv = torch.empty_like(qkv)
o = torch.empty(B, T, d, device=dev); do = torch.randn(B, T, d, device=dev)
lse = torch.empty(B, H, T, device=dev); delta = torch.empty_like(lse)
lens = torch.full((B,), T, dtype=torch.int64, device=dev)
mode = os.environ.get("MODE", "h3")          # h3 | x6 | f32
do = do * 1e-5
am = torch.empty(1024, device=dev)
lib.ttts_amax_partials(_p(do), do.numel(), _p(am), _stream())
fwd = {"h3": lib.ttts_attention_fwd_h3, "x6": lib.ttts_attention_fwd_x6, "f32": lib.ttts_attention_fwd}[mode]
if mode == "h3":
    def bwd(*args):
        return lib.ttts_attention_bwd_h3(*args[:-1], _p(am), None, None, args[-1])
else:
    bwd = lib.ttts_attention_bwd_x6 if mode == "x6" else lib.ttts_attention_bwd
for _ in range(3):
    fwd(_off(qkv, 0), _off(qkv, d), _off(qkv, 2 * d), _p(o), _p(lse), None, _p(lens), B, H, T, T, 3 * d, 3 * d, 3 * d, d, 1, 0.1, 7, None, _stream())
    bwd(_off(qkv, 0), _off(qkv, d), _off(qkv, 2 * d), _p(o), _p(do), _p(lse), _p(delta), _off(dqkv, 0), _off(dqkv, d),
                           _off(dqkv, 2 * d), _p(lens), B, H, T, T, 3 * d, 3 * d, 3 * d, d, 3 * d, 3 * d, 3 * d, 1, 0.1, 7, None, _stream())
torch.cuda.synchronize()
