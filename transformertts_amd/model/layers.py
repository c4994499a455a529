"""Transformer encoder / decoder stacks on the HIP kernels.

Class names, constructor signatures and parameter names follow the reference's
`model/layers.py:7-111` and the torch classes it builds on (`nn.MultiheadAttention`,
`nn.TransformerEncoderLayer`, `nn.TransformerEncoder`, torch/nn/modules/transformer.py), so
`state_dict()` is key-for-key the reference's.  Differences by design (MI355X-first):
  * masks never exist as tensors: the stacks take per-utterance lengths (`*_lens`, int64 on the
    device) and the attention kernels derive key-padding and causal masks from them;
    `*_key_padding_mask` / `tgt_mask` arguments are still accepted and converted (prefix masks only);
  * post-norm (`norm_first=False`, the reference's configuration) and pre-norm (`norm_first=True`, the other branch of
    model/layers.py:41-50); batch-first only, relu FFN;
  * residual adds, biases, relu and dropout live in GEMM epilogues, not in separate ops.
"""
from __future__ import annotations

import copy
from typing import Optional

import torch
import torch.nn as nn
from torch import Tensor

from .. import ops


def _lens_from_kpm(kpm: Optional[Tensor], B: int, T: int, device) -> Tensor:
    if kpm is None:
        return torch.full((B,), T, dtype=torch.int64, device=device)
    return (~kpm.bool()).sum(dim=1).to(torch.int64)


class MultiheadAttention(nn.Module):
    """Parameter layout of nn.MultiheadAttention (packed in-proj, `out_proj` sub-module).  Heads of 64 columns run on the
    head-image kernels, narrower ones on the padded fp32 kernels, wider ones as tensor algebra (ops._attention_wide_heads)."""

    def __init__(self, embed_dim: int, num_heads: int, dropout: float = 0.0):
        super().__init__()
        if num_heads <= 0 or embed_dim % num_heads != 0:
            raise ValueError("MultiheadAttention: embed_dim must be divisible by num_heads")
        self.embed_dim, self.num_heads, self.dropout = embed_dim, num_heads, dropout
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.empty(3 * embed_dim))
        self.out_proj = nn.Linear(embed_dim, embed_dim)
        nn.init.xavier_uniform_(self.in_proj_weight)     # torch MultiheadAttention._reset_parameters
        nn.init.constant_(self.in_proj_bias, 0.0)
        nn.init.constant_(self.out_proj.bias, 0.0)

    def _p(self) -> float:
        return self.dropout if self.training else 0.0

    def self_attention(self, x: Tensor, lens: Tensor, causal: bool, residual: Tensor, out_drop: float) -> Tensor:
        """residual + drop(out_proj(attention(in_proj(x))))"""
        skip = ops.SkipToken() if residual is x else None      # the skip gradient rides in the in-projection's epilogue
        # 64-column heads: q / k / v leave the in-projection as a head image (f16 hi / lo pieces with per-(row, head) scales in the
        # cells fp32 would occupy) and attention stages them by LDS-DMA; narrower heads take the fp32 path through padded copies
        img = 3 if ops.head_image_ok(x, self.in_proj_weight, self.num_heads, 3) else 0
        qkv = ops.linear(x, self.in_proj_weight, self.in_proj_bias, skip_in=skip, publish_amax=not img, head_image_sections=img)
        p = self._p()
        ctx = ops.self_attention(qkv, lens, self.num_heads, causal, p, ops.seeds.next() if p > 0 else 0)
        ctx._ttts_sole_consumer = True      # only the out-projection below reads it (see LinearFn.forward)
        return ops.linear(ctx, self.out_proj.weight, self.out_proj.bias, residual=residual, drop_p=out_drop,
                          seed=ops.seeds.next() if out_drop > 0 else 0, skip_out=skip)

    def cross_attention(self, x: Tensor, mem: Tensor, mem_lens: Tensor, residual: Tensor, out_drop: float,
                        need_weights: bool = True):
        d = self.embed_dim
        skip = ops.SkipToken() if residual is x else None
        wq, wkv = ops.param_rows(self.in_proj_weight, 0, d), ops.param_rows(self.in_proj_weight, d, 3 * d)
        img = ops.head_image_ok(x, wq, self.num_heads, 1) and ops.head_image_ok(mem, wkv, self.num_heads, 2)
        q = ops.linear(x, wq, ops.param_rows(self.in_proj_bias, 0, d), skip_in=skip, publish_amax=not img,
                       head_image_sections=1 if img else 0)
        kv = ops.linear(mem, wkv, ops.param_rows(self.in_proj_bias, d, 3 * d), publish_amax=not img,
                        head_image_sections=2 if img else 0)
        p = self._p()
        ctx, attn = ops.cross_attention(q, kv, mem_lens, self.num_heads, p, ops.seeds.next() if p > 0 else 0, need_weights)
        ctx._ttts_sole_consumer = True
        if not need_weights:
            attn = None
        out = ops.linear(ctx, self.out_proj.weight, self.out_proj.bias, residual=residual, drop_p=out_drop,
                         seed=ops.seeds.next() if out_drop > 0 else 0, skip_out=skip)
        return out, attn


def _ffn_block(layer, x: Tensor, out_dropout: nn.Dropout, residual: Optional[Tensor] = None) -> Tensor:
    """residual + drop_out(W2 . drop(relu(W1 x))), residual = x unless given (pre-norm: x is the normalised copy): relu+dropout
    ride in the first GEMM's epilogue, dropout+residual in the second's (torch `_ff_block`,
    torch/nn/modules/transformer.py:980-982,1197-1199)."""
    p = layer.dropout.p if layer.training else 0.0
    po = out_dropout.p if layer.training else 0.0
    skip = ops.SkipToken() if residual is None else None      # the skip gradient rides in the first GEMM's epilogue
    h = ops.linear(x, layer.linear1.weight, layer.linear1.bias, act=ops.ACT_RELU, drop_p=p,
                   seed=ops.seeds.next() if p > 0 else 0, skip_in=skip, publish_amax=True)
    return ops.linear(h, layer.linear2.weight, layer.linear2.bias, residual=x if residual is None else residual, drop_p=po,
                      seed=ops.seeds.next() if po > 0 else 0, sole_consumer=True, skip_out=skip)   # h feeds nothing else


class TransformerEncoderLayer(nn.Module):
    def __init__(self, d_model: int, nhead: int, dim_feedforward: int = 2048, dropout: float = 0.1,
                 activation: str = 'relu', batch_first: bool = True, norm_first: bool = False):
        super().__init__()
        if activation != 'relu' or not batch_first:
            raise ValueError("TransformerEncoderLayer: relu activation and batch_first only (the reference's configuration)")
        self.norm_first = norm_first
        self.self_attn = MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model, eps=1e-5)
        self.norm2 = nn.LayerNorm(d_model, eps=1e-5)
        self.dropout1 = nn.Dropout(dropout)
        self.dropout2 = nn.Dropout(dropout)

    def forward(self, src: Tensor, src_lens: Tensor) -> Tensor:
        p1 = self.dropout1.p if self.training else 0.0
        if self.norm_first:      # torch/nn/modules/transformer.py:944-950: x + SA(LN1(x)), then x + FF(LN2(x))
            x1 = ops.layer_norm(src, self.norm1.weight, self.norm1.bias, self.norm1.eps)
            s = self.self_attn.self_attention(x1, src_lens, False, residual=src, out_drop=p1)
            x2 = ops.layer_norm(s, self.norm2.weight, self.norm2.bias, self.norm2.eps)
            return _ffn_block(self, x2, self.dropout2, residual=s)
        s = self.self_attn.self_attention(src, src_lens, False, residual=src, out_drop=p1)
        x = ops.layer_norm(s, self.norm1.weight, self.norm1.bias, self.norm1.eps, sole_consumer=True)
        x = ops.layer_norm(_ffn_block(self, x, self.dropout2), self.norm2.weight, self.norm2.bias, self.norm2.eps,
                           sole_consumer=True)
        return x


class TransformerEncoder(nn.Module):
    def __init__(self, encoder_layer: TransformerEncoderLayer, num_layers: int, norm=None):
        super().__init__()
        self.layers = nn.ModuleList([copy.deepcopy(encoder_layer) for _ in range(num_layers)])
        self.num_layers = num_layers
        self.norm = norm

    def forward(self, src: Tensor, mask=None, src_key_padding_mask: Optional[Tensor] = None,
                src_lens: Optional[Tensor] = None) -> Tensor:
        if mask is not None:
            raise ValueError("TransformerEncoder: arbitrary attention masks are not supported, pass lengths")
        if src_lens is None:
            src_lens = _lens_from_kpm(src_key_padding_mask, src.size(0), src.size(1), src.device)
        x = src
        for layer in self.layers:
            x = layer(x, src_lens)
        if self.norm is not None:
            x = ops.layer_norm(x, self.norm.weight, self.norm.bias, self.norm.eps)
        return x


class TransformerDecoderLayer(nn.Module):
    '''
    Decoder layer which returns the per-head cross-attention weights (reference model/layers.py:7-74).
    '''

    def __init__(self, d_model: int, nhead: int, dim_feedforward: int = 2048, dropout: float = 0.1,
                 norm_first: bool = False, batch_first: bool = True):
        super().__init__()
        if not batch_first:
            raise ValueError("TransformerDecoderLayer: batch_first only (the reference's configuration)")
        self.self_attn = MultiheadAttention(d_model, nhead, dropout=dropout)
        self.multihead_attn = MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm_first = norm_first
        self.norm1 = nn.LayerNorm(d_model, eps=1e-5)
        self.norm2 = nn.LayerNorm(d_model, eps=1e-5)
        self.norm3 = nn.LayerNorm(d_model, eps=1e-5)
        self.dropout1 = nn.Dropout(dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.dropout3 = nn.Dropout(dropout)

    def forward(self, tgt: Tensor, memory: Tensor, tgt_mask: Optional[Tensor] = None,
                memory_mask: Optional[Tensor] = None, tgt_key_padding_mask: Optional[Tensor] = None,
                memory_key_padding_mask: Optional[Tensor] = None, tgt_is_causal: bool = True,
                memory_is_causal: bool = False, tgt_lens: Optional[Tensor] = None,
                memory_lens: Optional[Tensor] = None, need_alignments: bool = True):
        if memory_mask is not None or memory_is_causal:
            raise ValueError("TransformerDecoderLayer: memory masks other than key padding are not supported")
        B = tgt.size(0)
        if tgt_lens is None:
            tgt_lens = _lens_from_kpm(tgt_key_padding_mask, B, tgt.size(1), tgt.device)
        if memory_lens is None:
            memory_lens = _lens_from_kpm(memory_key_padding_mask, B, memory.size(1), tgt.device)
        causal = bool(tgt_is_causal) or tgt_mask is not None
        tr = self.training
        if self.norm_first:      # reference model/layers.py:41-45
            x1 = ops.layer_norm(tgt, self.norm1.weight, self.norm1.bias, self.norm1.eps)
            s = self.self_attn.self_attention(x1, tgt_lens, causal, residual=tgt, out_drop=self.dropout1.p if tr else 0.0)
            x2 = ops.layer_norm(s, self.norm2.weight, self.norm2.bias, self.norm2.eps)
            s2, alignments = self.multihead_attn.cross_attention(x2, memory, memory_lens, residual=s,
                                                                 out_drop=self.dropout2.p if tr else 0.0,
                                                                 need_weights=need_alignments)
            x3 = ops.layer_norm(s2, self.norm3.weight, self.norm3.bias, self.norm3.eps)
            return _ffn_block(self, x3, self.dropout3, residual=s2), alignments
        s = self.self_attn.self_attention(tgt, tgt_lens, causal, residual=tgt, out_drop=self.dropout1.p if tr else 0.0)
        x = ops.layer_norm(s, self.norm1.weight, self.norm1.bias, self.norm1.eps, sole_consumer=True)
        s, alignments = self.multihead_attn.cross_attention(x, memory, memory_lens, residual=x,
                                                            out_drop=self.dropout2.p if tr else 0.0,
                                                            need_weights=need_alignments)
        x = ops.layer_norm(s, self.norm2.weight, self.norm2.bias, self.norm2.eps, sole_consumer=True)
        x = ops.layer_norm(_ffn_block(self, x, self.dropout3), self.norm3.weight, self.norm3.bias, self.norm3.eps,
                           sole_consumer=True)
        return x, alignments


class TransformerDecoder(nn.Module):
    '''
    Decoder stack which returns the alignments of every layer (reference model/layers.py:77-111).
    '''

    def __init__(self, decoder_layer, num_layers, norm=None):
        super().__init__()
        self.layers = nn.ModuleList([copy.deepcopy(decoder_layer) for _ in range(num_layers)])
        self.num_layers = num_layers
        self.norm = norm

    def forward(self, tgt: Tensor, memory: Tensor, tgt_mask: Optional[Tensor] = None,
                memory_mask: Optional[Tensor] = None, tgt_key_padding_mask: Optional[Tensor] = None,
                memory_key_padding_mask: Optional[Tensor] = None, tgt_is_causal: Optional[bool] = None,
                memory_is_causal: Optional[bool] = None, tgt_lens: Optional[Tensor] = None,
                memory_lens: Optional[Tensor] = None, need_alignments: bool = True):
        B = tgt.size(0)
        if tgt_lens is None:
            tgt_lens = _lens_from_kpm(tgt_key_padding_mask, B, tgt.size(1), tgt.device)
        if memory_lens is None:
            memory_lens = _lens_from_kpm(memory_key_padding_mask, B, memory.size(1), tgt.device)
        alignments = []
        memories = ops.fanout(memory, len(self.layers))     # one handle per layer: their gradients meet in one launch
        for layer, memory in zip(self.layers, memories):
            tgt, alignment = layer(tgt, memory, tgt_mask=tgt_mask, memory_mask=memory_mask,
                                   tgt_is_causal=True if tgt_is_causal is None else tgt_is_causal,
                                   memory_is_causal=bool(memory_is_causal), tgt_lens=tgt_lens,
                                   memory_lens=memory_lens, need_alignments=need_alignments)
            alignments.append(alignment)
        if self.norm is not None:
            tgt = ops.layer_norm(tgt, self.norm.weight, self.norm.bias, self.norm.eps)
        return tgt, alignments
