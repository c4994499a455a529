#!/usr/bin/env python3
"""A few launches of the head-image attention kernels at the BASELINE self-attention shape (64 x 4 x 870 causal, dropout 0.1) for
rocprofv3 --pmc passes (tools/pmc_aimg.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformertts_amd import _lib, ops
from transformertts_amd.ops import _p, _off, _stream
lib = _lib.load(); dev = torch.device("cuda:0")
B, H, T = 64, 4, 870
d = H * 64
qkv = torch.randn(B * T, 3 * d, device=dev)
img, inv = torch.empty_like(qkv), torch.empty(3 * H, B * T, device=dev)
_lib.check(lib.ttts_head_image(_p(qkv), 3 * d, _p(img), 3 * d, _p(inv), B * T, 3 * d, _stream()), "head_image")
va = ops._amax(qkv[:, 2 * d:].contiguous())
lens = torch.full((B,), T, dtype=torch.int64, device=dev)
o = torch.empty(B, T, d, device=dev); stat = torch.empty(6, B, H, T, device=dev)
do = torch.randn(B, T, d, device=dev) * 1e-5; doa = ops._amax(do)
dqkv = torch.empty(B * T, 3 * d, device=dev); delta = torch.empty(B, H, T, device=dev)
HM = H * B * T
for _ in range(3):
    _lib.check(lib.ttts_attention_fwd_img(_off(img, 0), _off(img, d), _off(img, 2 * d), _off(inv, 0), _off(inv, HM), _off(inv, 2 * HM), _p(o),
                                          _p(stat[0]), None, _p(lens), B, H, T, T, 3 * d, 3 * d, 3 * d, d, 1, 0.125, 0.1, 5, None, _p(va), None,
                                          _p(stat[1:]), 0, 0, 0, _stream()), "fwd")
    _lib.check(lib.ttts_attention_bwd_img(_off(img, 0), _off(img, d), _off(img, 2 * d), _off(inv, 0), _off(inv, HM), _off(inv, 2 * HM), _p(o),
                                          _p(do), _p(stat[1:]), _p(delta), _off(dqkv, 0), _off(dqkv, d), _off(dqkv, 2 * d), _p(lens), B, H, T, T,
                                          3 * d, 3 * d, 3 * d, d, 3 * d, 3 * d, 3 * d, 1, 0.125, 0.1, 5, None, _p(doa), None, None, None, 1,
                                          0, 0, 0, _stream()), "bwd")
torch.cuda.synchronize()
