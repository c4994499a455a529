"""ConvNormBN and LinearNorm with the reference's constructor signatures, init and state-dict keys
(/root/reference/model/module.py:4-53); the arithmetic runs in the gfx950 kernels of `ops`.

`nn.Conv1d` / `nn.BatchNorm1d` / `nn.Linear` objects are kept purely as parameter containers so that
`state_dict()` keys, shapes and default initialisation are the reference's; their own `forward` is
never called.
"""
from __future__ import annotations

import torch.nn as nn
from torch import Tensor

from .. import ops


class ConvNormBN(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, kernel_size: int, padding: int = None,
                 activation: str = 'relu'):
        super().__init__()
        if padding is None:
            padding = (kernel_size - 1) // 2
        if padding != (kernel_size - 1) // 2 or kernel_size % 2 == 0:
            raise ValueError("ConvNormBN: the HIP implicit-GEMM conv supports odd kernels with 'same' padding only")
        self.conv = nn.Conv1d(in_channels, out_channels, kernel_size, padding=padding)
        self.bn = nn.BatchNorm1d(out_channels)
        gain = nn.init.calculate_gain(activation)
        nn.init.xavier_normal_(self.conv.weight, gain=gain)
        nn.init.constant_(self.conv.bias, 0.0)
        nn.init.constant_(self.bn.weight, 1.0)
        nn.init.constant_(self.bn.bias, 0.0)

    def fused(self, x: Tensor, act: int = ops.ACT_NONE, drop_p: float = 0.0, twin_last: bool = False) -> Tensor:
        """conv -> batch-norm -> optional tanh -> optional dropout in one autograd node, (B,T,C) in and out.
        twin_last: see ops.conv_bn (the last layer of a pass over a twin batch)."""
        bn = self.bn
        p = drop_p if self.training else 0.0
        return ops.conv_bn(x, self.conv.weight, self.conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                           bn.num_batches_tracked, self.training, bn.momentum, bn.eps, act, p,
                           ops.seeds.next() if p > 0.0 else 0, twin_last=twin_last)

    def forward(self, x: Tensor) -> Tensor:
        return self.fused(x)


class LinearNorm(nn.Module):
    def __init__(self, in_features: int, out_features: int, bias: bool = True, activation: str = 'relu'):
        super().__init__()
        self.linear = nn.Linear(in_features, out_features, bias=bias)
        gain = nn.init.calculate_gain(activation)
        nn.init.xavier_normal_(self.linear.weight, gain=gain)
        if bias:
            nn.init.constant_(self.linear.bias, 0.0)

    def forward(self, x: Tensor) -> Tensor:
        return ops.linear(x, self.linear.weight, self.linear.bias)
