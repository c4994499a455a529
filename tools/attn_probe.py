#!/usr/bin/env python3
"""Launch the causal self-attention forward + backward (fp16x3 kernels, through the ops layer) a few times on the step's
shape: the target of the rocprofv3 --pmc passes of tools/pmc_attn.sh."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformertts_amd import ops
dev = torch.device("cuda:0")
B, H, T = 64, 4, 870
d = H * 64
qkv = torch.randn(B, T, 3 * d, device=dev, requires_grad=True)
do = torch.randn(B, T, d, device=dev) * 1e-5
lens = torch.full((B,), T, dtype=torch.int64, device=dev)
for _ in range(3):
    o = ops.self_attention(qkv, lens, H, True, 0.1, 7)
    o.backward(do)
torch.cuda.synchronize()
