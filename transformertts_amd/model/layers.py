"""Transformer encoder / decoder stacks on the HIP kernels.

Class names, constructor signatures and parameter names follow the reference's
`model/layers.py:7-111` and the torch classes it builds on (`nn.MultiheadAttention`,
`nn.TransformerEncoderLayer`, `nn.TransformerEncoder`, torch/nn/modules/transformer.py), so
`state_dict()` is key-for-key the reference's.  Differences by design (MI355X-first):
  * masks never exist as tensors: the stacks take per-utterance lengths (`*_lens`, int64 on the
    device) and the attention kernels derive key-padding and causal masks from them;
    `*_key_padding_mask` / `tgt_mask` / `memory_mask` / `mask` arguments are still accepted: a key-padding mask that is a
    prefix mask becomes lengths and a `tgt_mask` that is the causal mask becomes the kernels' causal flag (one host check per
    call, only when a mask TENSOR is passed -- the model passes lengths); any other mask keeps its meaning and that attention
    runs as tensor algebra on library GEMMs (`ops.masked_attention`: correct, not tuned);
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


_NEG = torch.finfo(torch.float32).min


def _dead_keys(kpm: Tensor) -> Tensor:
    """torch `key_padding_mask` (B, T) -> bool, True = key ignored (a float mask: its -inf entries; anything else but 0 has no
    length / hole reading and is refused)"""
    if kpm.dtype == torch.bool:
        return kpm
    if kpm.is_floating_point():
        dead = torch.isneginf(kpm)
        if bool(((kpm != 0) & ~dead).any()):
            raise ValueError("key_padding_mask: float masks may hold 0 and -inf only")
        return dead
    return kpm != 0


def _resolve_kpm(kpm: Optional[Tensor], B: int, T: int, device):
    """-> (lengths, dead): a prefix mask (live keys first) is what the kernels derive from lengths -> (lens, None); a mask with
    holes stays a tensor -> (live-key counts, dead (B, T) bool) and the attention that uses it runs in `ops.masked_attention`."""
    if kpm is None:
        return torch.full((B,), T, dtype=torch.int64, device=device), None
    if kpm.dim() != 2 or kpm.shape[0] != B or kpm.shape[1] != T:
        raise ValueError(f"key_padding_mask: expected ({B}, {T}), got {tuple(kpm.shape)}")
    dead = _dead_keys(kpm)
    lens = (~dead).sum(dim=1).to(torch.int64)
    prefix = torch.arange(T, device=dead.device)[None, :] >= lens[:, None]
    return lens, (None if bool(torch.equal(dead, prefix)) else dead)


def _lens_from_kpm(kpm: Optional[Tensor], B: int, T: int, device) -> Tensor:
    lens, dead = _resolve_kpm(kpm, B, T, device)
    if dead is not None:
        raise ValueError("key_padding_mask with holes: pass it to the layer / stack (it cannot be expressed as lengths)")
    return lens


def _is_causal_mask(mask: Tensor, Tq: int, Tk: int) -> bool:
    """is `mask` exactly the mask torch's generate_square_subsequent_mask / the reference's model/model.py:251-255 build?"""
    if mask.dim() != 2 or Tq != Tk or mask.shape[0] != Tq or mask.shape[1] != Tk:
        return False
    ref = torch.triu(torch.ones(Tq, Tk, dtype=torch.bool, device=mask.device), diagonal=1)
    if mask.dtype == torch.bool:
        return bool(torch.equal(mask, ref))
    return bool(torch.equal(torch.isneginf(mask), ref)) and bool((mask.masked_fill(ref, 0) == 0).all())


def _additive_mask(mask: Tensor, B: int, H: int, Tq: int, Tk: int) -> Tensor:
    """torch `attn_mask` ((Tq, Tk) or (B * H, Tq, Tk); bool True = not allowed, float = added to the scores) -> finite fp32,
    broadcastable to (B, H, Tq, Tk)"""
    if mask.dim() == 2 and mask.shape[0] == Tq and mask.shape[1] == Tk:
        m = mask[None, None]
    elif mask.dim() == 3 and mask.shape[0] == B * H and mask.shape[1] == Tq and mask.shape[2] == Tk:
        m = mask.reshape(B, H, Tq, Tk)
    else:
        raise ValueError(f"attention mask: expected ({Tq}, {Tk}) or ({B * H}, {Tq}, {Tk}), got {tuple(mask.shape)}")
    if m.dtype == torch.bool:
        return torch.zeros(m.shape, dtype=torch.float32, device=m.device).masked_fill(m, _NEG)
    if bool((torch.isnan(m) | torch.isposinf(m)).any()):
        raise ValueError("attention mask: float masks may hold finite values and -inf only (NaN / +inf would reach the softmax)")
    return m.to(torch.float32).clamp_min(_NEG)


def _lens_and_kpm(lens: Optional[Tensor], kpm: Optional[Tensor], B: int, T: int, device, what: str):
    """-> (lens, dead) of one side of an attention.  Lengths alone are the model's own call; a key-padding mask alone is torch's;
    BOTH: the mask must be the prefix mask of those very lengths (anything else would silently lose its holes -- refused)."""
    if lens is None:
        return _resolve_kpm(kpm, B, T, device)
    if kpm is not None:
        got, dead = _resolve_kpm(kpm, B, T, device)
        if dead is not None or not bool(torch.equal(got.to(lens.device), lens.to(torch.int64).clamp(0, T))):
            raise ValueError(f"{what}: both lengths and a key-padding mask were given and the mask is not the prefix mask of those "
                             "lengths; pass one of them")
    return lens, None


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

    def self_attention(self, x: Tensor, lens: Tensor, causal: bool, residual: Tensor, out_drop: float,
                       dead: Optional[Tensor] = None, add_mask: Optional[Tensor] = None) -> Tensor:
        """residual + drop(out_proj(attention(in_proj(x)))); `dead` / `add_mask`: masks the kernels do not derive from lengths
        (`_resolve_kpm`, `_additive_mask`) -- that attention runs in `ops.masked_attention` between the same two GEMMs"""
        skip = ops.SkipToken() if residual is x else None      # the skip gradient rides in the in-projection's epilogue
        if dead is not None or add_mask is not None:
            d = self.embed_dim
            qkv = ops.linear(x, self.in_proj_weight, self.in_proj_bias, skip_in=skip, publish_amax=True)
            ctx, _ = ops.masked_attention(qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:], lens, self.num_heads, causal,
                                          self._p(), dead, add_mask)
            return ops.linear(ctx, self.out_proj.weight, self.out_proj.bias, residual=residual, drop_p=out_drop,
                              seed=ops.seeds.next() if out_drop > 0 else 0, skip_out=skip)
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
                        need_weights: bool = True, dead: Optional[Tensor] = None, add_mask: Optional[Tensor] = None, kv=None):
        """`kv`: None, or this layer's (k | v) HeadImage from the stack's one K/V projection of the memory
        (`ops.cross_kv_projection`, TransformerDecoder.forward): the layer then projects only its queries"""
        d = self.embed_dim
        skip = ops.SkipToken() if residual is x else None
        wq, wkv = ops.param_rows(self.in_proj_weight, 0, d), ops.param_rows(self.in_proj_weight, d, 3 * d)
        if kv is not None:
            if dead is not None or add_mask is not None:
                raise ValueError("cross_attention: a pre-projected K/V image takes length masks only")
            q = ops.linear(x, wq, ops.param_rows(self.in_proj_bias, 0, d), skip_in=skip, head_image_sections=1)
            p = self._p()
            ctx, attn = ops.cross_attention(q, kv, mem_lens, self.num_heads, p, ops.seeds.next() if p > 0 else 0, need_weights)
            ctx._ttts_sole_consumer = True
            out = ops.linear(ctx, self.out_proj.weight, self.out_proj.bias, residual=residual, drop_p=out_drop,
                             seed=ops.seeds.next() if out_drop > 0 else 0, skip_out=skip)
            return out, (attn if need_weights else None)
        if dead is not None or add_mask is not None:
            q = ops.linear(x, wq, ops.param_rows(self.in_proj_bias, 0, d), skip_in=skip, publish_amax=True)
            kv = ops.linear(mem, wkv, ops.param_rows(self.in_proj_bias, d, 3 * d), publish_amax=True)
            ctx, attn = ops.masked_attention(q, kv[..., :d], kv[..., d:], mem_lens, self.num_heads, False, self._p(), dead, add_mask)
            out = ops.linear(ctx, self.out_proj.weight, self.out_proj.bias, residual=residual, drop_p=out_drop,
                             seed=ops.seeds.next() if out_drop > 0 else 0, skip_out=skip)
            return out, (attn if need_weights else None)
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

    def forward(self, src: Tensor, src_lens: Tensor, dead: Optional[Tensor] = None, add_mask: Optional[Tensor] = None) -> Tensor:
        p1 = self.dropout1.p if self.training else 0.0
        if self.norm_first:      # torch/nn/modules/transformer.py:944-950: x + SA(LN1(x)), then x + FF(LN2(x))
            x1 = ops.layer_norm(src, self.norm1.weight, self.norm1.bias, self.norm1.eps)
            s = self.self_attn.self_attention(x1, src_lens, False, residual=src, out_drop=p1, dead=dead, add_mask=add_mask)
            x2 = ops.layer_norm(s, self.norm2.weight, self.norm2.bias, self.norm2.eps)
            return _ffn_block(self, x2, self.dropout2, residual=s)
        s = self.self_attn.self_attention(src, src_lens, False, residual=src, out_drop=p1, dead=dead, add_mask=add_mask)
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
        B, T = src.size(0), src.size(1)
        src_lens, dead = _lens_and_kpm(src_lens, src_key_padding_mask, B, T, src.device, "TransformerEncoder")
        add_mask = None if mask is None else _additive_mask(mask, B, self.layers[0].self_attn.num_heads, T, T)
        x = src
        for layer in self.layers:
            x = layer(x, src_lens, dead, add_mask)
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
                memory_lens: Optional[Tensor] = None, need_alignments: bool = True, memory_kv=None):
        if memory_is_causal and memory_mask is None:
            raise ValueError("TransformerDecoderLayer: memory_is_causal is a hint about memory_mask and needs one (as torch)")
        B, Tq, Tk, H = tgt.size(0), tgt.size(1), memory.size(1), self.self_attn.num_heads
        tgt_lens, tgt_dead = _lens_and_kpm(tgt_lens, tgt_key_padding_mask, B, Tq, tgt.device, "TransformerDecoderLayer (tgt)")
        memory_lens, mem_dead = _lens_and_kpm(memory_lens, memory_key_padding_mask, B, Tk, tgt.device, "TransformerDecoderLayer (memory)")
        # tgt_mask: the causal mask (what the reference's model passes, model/model.py:251-255) is the kernels' causal flag; any
        # other mask is applied as given.  Without a mask tensor `tgt_is_causal` (default True, model/layers.py:36) decides.
        tgt_add = None
        if tgt_mask is None:
            causal = bool(tgt_is_causal)
        elif _is_causal_mask(tgt_mask, Tq, Tq):
            causal = True
        else:
            causal, tgt_add = False, _additive_mask(tgt_mask, B, H, Tq, Tq)
        mem_add = None if memory_mask is None else _additive_mask(memory_mask, B, H, Tq, Tk)
        tr = self.training
        sa = dict(dead=tgt_dead, add_mask=tgt_add)
        ca = dict(dead=mem_dead, add_mask=mem_add)
        if self.norm_first:      # reference model/layers.py:41-45
            x1 = ops.layer_norm(tgt, self.norm1.weight, self.norm1.bias, self.norm1.eps)
            s = self.self_attn.self_attention(x1, tgt_lens, causal, residual=tgt, out_drop=self.dropout1.p if tr else 0.0, **sa)
            x2 = ops.layer_norm(s, self.norm2.weight, self.norm2.bias, self.norm2.eps)
            s2, alignments = self.multihead_attn.cross_attention(x2, memory, memory_lens, residual=s,
                                                                 out_drop=self.dropout2.p if tr else 0.0,
                                                                 need_weights=need_alignments, kv=memory_kv, **ca)
            x3 = ops.layer_norm(s2, self.norm3.weight, self.norm3.bias, self.norm3.eps)
            return _ffn_block(self, x3, self.dropout3, residual=s2), alignments
        s = self.self_attn.self_attention(tgt, tgt_lens, causal, residual=tgt, out_drop=self.dropout1.p if tr else 0.0, **sa)
        x = ops.layer_norm(s, self.norm1.weight, self.norm1.bias, self.norm1.eps, sole_consumer=True)
        s, alignments = self.multihead_attn.cross_attention(x, memory, memory_lens, residual=x,
                                                            out_drop=self.dropout2.p if tr else 0.0,
                                                            need_weights=need_alignments, kv=memory_kv, **ca)
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
        # prefix key-padding masks become lengths once, here; masks with holes travel on to the layers as tensors
        if tgt_key_padding_mask is not None:
            lens, dead = _lens_and_kpm(tgt_lens, tgt_key_padding_mask, B, tgt.size(1), tgt.device, "TransformerDecoder (tgt)")
            if dead is None:
                tgt_lens, tgt_key_padding_mask = lens, None
        if memory_key_padding_mask is not None:
            lens, dead = _lens_and_kpm(memory_lens, memory_key_padding_mask, B, memory.size(1), tgt.device, "TransformerDecoder (memory)")
            if dead is None:
                memory_lens, memory_key_padding_mask = lens, None
        if tgt_mask is not None and _is_causal_mask(tgt_mask, tgt.size(1), tgt.size(1)):
            tgt_mask, tgt_is_causal = None, True            # (checked once for the stack, not once per layer)
        alignments = []
        # The layers' cross-attentions all project the SAME memory (reference model/layers.py:54-74, once per layer): with plain
        # length masks and 64-column heads that is ONE GEMM of N = layers x 2 d_model columns (ops.cross_kv_projection) whose
        # backward also sums the layers' memory gradients; otherwise one handle per layer, their gradients meeting in one launch
        attns = [layer.multihead_attn for layer in self.layers]
        H = attns[0].num_heads
        fused = (memory_mask is None and memory_key_padding_mask is None and memory.dim() == 3 and
                 ops.cross_kv_ok(memory, attns, H) and
                 ops.head_image_ok(tgt, attns[0].in_proj_weight.detach()[:memory.size(-1)], H, 1))
        if fused:
            if memory_lens is None:
                memory_lens = torch.full((B,), memory.size(1), dtype=torch.int64, device=tgt.device)
            kvs = ops.cross_kv_projection(memory, attns, self)
            memories = [memory] * len(self.layers)          # (shapes only: the layers do not read it)
        else:
            kvs = [None] * len(self.layers)
            memories = ops.fanout(memory, len(self.layers))     # one handle per layer: their gradients meet in one launch
        for layer, memory, kv in zip(self.layers, memories, kvs):
            tgt, alignment = layer(tgt, memory, tgt_mask=tgt_mask, memory_mask=memory_mask,
                                   tgt_key_padding_mask=tgt_key_padding_mask,
                                   memory_key_padding_mask=memory_key_padding_mask,
                                   tgt_is_causal=True if tgt_is_causal is None else tgt_is_causal,
                                   memory_is_causal=bool(memory_is_causal), tgt_lens=tgt_lens,
                                   memory_lens=memory_lens, need_alignments=need_alignments, memory_kv=kv)
            alignments.append(alignment)
        if self.norm is not None:
            tgt = ops.layer_norm(tgt, self.norm.weight, self.norm.bias, self.norm.eps)
        return tgt, alignments
