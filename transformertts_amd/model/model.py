"""TransformerTTS on hand-written gfx950 kernels -- same constructor kwargs, `forward` signature, output
dict and `state_dict()` keys as the reference's `model/model.py:138-394`, so it drops in behind
`lightning_module.py::training_step()` unchanged.

What differs from the reference, by design:
  * `forward` never materialises masks and never synchronises with the host (`_get_mask`'s two `.item()` calls,
    model/model.py:236,245): lengths go straight to the attention kernels;
  * the go-frame shift (model/model.py:278-279) is folded into the decoder pre-net's first GEMM loader;
  * conv -> batch-norm -> tanh -> dropout of every ConvNormBN block is one autograd node on (B,T,C) without
    permutes; biases / relu / dropout / residual adds live in GEMM epilogues.
The `nn.Dropout` / `nn.Tanh` entries are kept in the ModuleLists (they fix the state-dict indices 0,2,4 /
0,3,6,... and carry `p`), but are applied inside the fused kernels.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
from torch import Tensor

from .. import ops
from .layers import (TransformerDecoder, TransformerDecoderLayer, TransformerEncoder, TransformerEncoderLayer)
from .module import ConvNormBN, LinearNorm


class EncoderPreNet(nn.Module):
    """N x [ConvNormBN -> Dropout] (no activation) + LinearNorm; (B,T,H) -> (B,T,H).  reference :13-45"""

    def __init__(self, n_layers: int, in_channels: int, out_channels: int, kernel_size: int, dropout: float = 0.5):
        super().__init__()
        self.layers = nn.ModuleList()
        for i in range(n_layers):
            in_dim = in_channels if i == 0 else out_channels
            self.layers.append(ConvNormBN(in_dim, out_channels, kernel_size))
            self.layers.append(nn.Dropout(dropout))
        self.linear = LinearNorm(out_channels, out_channels)

    def forward(self, x: Tensor) -> Tensor:
        mods = list(self.layers)
        for conv, drop in zip(mods[0::2], mods[1::2]):
            x = conv.fused(x, ops.ACT_NONE, drop.p)
        return self.linear(x)


class DecoderPreNet(nn.Module):
    """Drop(relu(L1 x)) -> Drop(relu(L2 .)), p fixed at the ctor default 0.5.  reference :48-67"""

    def __init__(self, n_mels: int, d_model: int, dropout: float = 0.5):
        super().__init__()
        self.linear1 = LinearNorm(n_mels, d_model, activation='relu')
        self.linear2 = LinearNorm(d_model, d_model, activation='relu')
        self.dropout1 = nn.Dropout(dropout)
        self.dropout2 = nn.Dropout(dropout)

    def forward(self, x: Tensor, shift_right: bool = False) -> Tensor:
        """`shift_right=True` reads frame t-1 for output frame t (zeros at t = 0): the go-frame shift."""
        p1 = self.dropout1.p if self.training else 0.0
        p2 = self.dropout2.p if self.training else 0.0
        l1, l2 = self.linear1.linear, self.linear2.linear
        x = ops.linear(x, l1.weight, l1.bias, act=ops.ACT_RELU, drop_p=p1, seed=ops.seeds.next() if p1 > 0 else 0,
                       row_shift=-1 if shift_right else 0, T=x.size(1), publish_amax=True)
        return ops.linear(x, l2.weight, l2.bias, act=ops.ACT_RELU, drop_p=p2, seed=ops.seeds.next() if p2 > 0 else 0,
                          sole_consumer=True)


class PositionalEncoding(nn.Module):
    """x + alpha * pe[:T], then dropout; one learnable alpha.  reference :70-97"""

    def __init__(self, d_model: int, device: str, dropout: float = 0.1, max_len: int = 5000):
        super().__init__()
        position = torch.arange(0, max_len, dtype=torch.float32).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float32) * (-math.log(10000.0)) / d_model)
        pe = torch.zeros(max_len, d_model, dtype=torch.float32)
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
        self.register_buffer('pe', pe.to(device))
        self.alpha = nn.Parameter(torch.ones(1), requires_grad=True)
        self.dropout = nn.Dropout(dropout)

    def forward(self, x: Tensor) -> Tensor:
        p = self.dropout.p if self.training else 0.0
        return ops.posenc(x, self.pe, self.alpha, p, ops.seeds.next() if p > 0 else 0)


class PostNet(nn.Module):
    """ConvNormBN+tanh+drop, (n-2) x same, ConvNormBN+drop; (B,T,n_mels) -> (B,T,n_mels).  reference :100-135"""

    def __init__(self, n_layers: int, in_channels: int, out_channels: int, kernel_size: int, dropout: float = 0.5):
        super().__init__()
        self.layers = nn.ModuleList()
        self.layers.append(ConvNormBN(in_channels, out_channels, kernel_size, activation='tanh'))
        self.layers.append(nn.Tanh())
        self.layers.append(nn.Dropout(dropout))
        for _ in range(n_layers - 2):
            self.layers.append(ConvNormBN(out_channels, out_channels, kernel_size, activation='tanh'))
            self.layers.append(nn.Tanh())
            self.layers.append(nn.Dropout(dropout))
        self.layers.append(ConvNormBN(out_channels, in_channels, kernel_size, activation='tanh'))
        self.layers.append(nn.Dropout(dropout))

    def forward(self, x: Tensor) -> Tensor:
        mods = list(self.layers)
        i = 0
        while i < len(mods):
            conv = mods[i]
            act, p = ops.ACT_NONE, 0.0
            i += 1
            if i < len(mods) and isinstance(mods[i], nn.Tanh):
                act = ops.ACT_TANH
                i += 1
            if i < len(mods) and isinstance(mods[i], nn.Dropout):
                p = mods[i].p
                i += 1
            x = conv.fused(x, act, p, twin_last=i >= len(mods))      # (a twin batch ends here: ops.PostnetTwin)
        return x


class TransformerTTS(nn.Module):
    def __init__(
        self,
        encoder_prenet_n_layers: int,
        encoder_prenet_in_channel: int,
        encoder_prenet_out_channel: int,
        encoder_prenet_kernel_size: int,
        encoder_prenet_dropout: float,
        encoder_n_layers: int,
        encoder_n_head: int,
        encoder_d_ffn: int,
        encoder_dropout: float,
        decoder_n_layers: int,
        decoder_n_head: int,
        decoder_d_ffn: int,
        decoder_dropout: float,
        postnet_n_layers: int,
        postnet_kernel_size: int,
        postnet_dropout: float,
        d_model: int,
        n_phon: int = 100,
        n_mels: int = 80,
        device: str = 'cuda',
    ):
        super().__init__()
        self.device = device
        self.n_mels = n_mels
        self.emb = nn.Embedding(n_phon, d_model)
        self.enc_prenet = EncoderPreNet(encoder_prenet_n_layers, encoder_prenet_in_channel,
                                        encoder_prenet_out_channel, encoder_prenet_kernel_size,
                                        encoder_prenet_dropout)
        self.dec_prenet = DecoderPreNet(n_mels, d_model)
        self.pe = PositionalEncoding(d_model, device)
        encoder_layer = TransformerEncoderLayer(d_model=d_model, nhead=encoder_n_head,
                                                dim_feedforward=encoder_d_ffn, dropout=encoder_dropout,
                                                activation='relu', batch_first=True)
        self.encoder = TransformerEncoder(encoder_layer=encoder_layer, num_layers=encoder_n_layers)
        decoder_layer = TransformerDecoderLayer(d_model=d_model, nhead=decoder_n_head,
                                                dim_feedforward=decoder_d_ffn, dropout=decoder_dropout,
                                                batch_first=True)
        self.decoder = TransformerDecoder(decoder_layer=decoder_layer, num_layers=decoder_n_layers)
        self.postnet = PostNet(postnet_n_layers, n_mels, d_model, postnet_kernel_size, postnet_dropout)
        self.linear1 = LinearNorm(d_model, n_mels)
        self.linear2 = LinearNorm(d_model, 1)  # stop token

    def _get_mask(self, phoneme_lens: Tensor = None, mel_lens: Tensor = None):
        """Boolean masks as the reference builds them (model/model.py:229-257).  Kept for API parity; the
        forward pass does not call it (it costs two host synchronisations)."""
        src_kpm = tgt_kpm = tgt_mask = None
        if phoneme_lens is not None:
            n = int(phoneme_lens.max().item())
            src_kpm = torch.arange(n, device=phoneme_lens.device).unsqueeze(0) >= phoneme_lens.unsqueeze(1)
        if mel_lens is not None:
            n = int(mel_lens.max().item())
            tgt_kpm = torch.arange(n, device=mel_lens.device).unsqueeze(0) >= mel_lens.unsqueeze(1)
            tgt_mask = torch.triu(torch.ones(n, n, device=mel_lens.device), diagonal=1).bool()
        return src_kpm, tgt_kpm, tgt_mask

    def encode(self, phoneme: Tensor, phoneme_lens: Tensor) -> Tensor:
        x = ops.embedding(phoneme, self.emb.weight)
        x = self.pe(self.enc_prenet(x))
        return self.encoder(x, src_lens=phoneme_lens)

    def twin_encode_ok(self, phoneme: Tensor) -> bool:
        """can `encode_twin` serve this model?  (64-column heads: the encoder's attention then runs on head images, the only form
        the twin batch takes; fp16x3 forms selected; training mode -- in eval mode there is one forward)"""
        d = self.emb.weight.shape[1]
        layers = list(self.encoder.layers)
        return (ops.TWIN_ENCODER and self.training and phoneme.is_cuda and len(layers) > 0 and self.encoder.norm is None and
                all(l.self_attn.embed_dim == d and d == l.self_attn.num_heads * 64 and not l.norm_first for l in layers) and
                ops.HEAD_IMAGES and ops.ATTN_FWD_MODE == "h3" and ops.ATTN_BWD_MODE == "h3" and ops._fwd_h3(d, 3 * d) and
                2 * phoneme.numel() * 3 * d * 4 < (1 << 31))

    def twin_postnet_ok(self, melspec: Tensor) -> bool:
        """can both forwards of a training step share ONE post-net pass (ops.PostnetTwin)?  Training mode (in eval mode BatchNorm
        has no batch statistics to keep apart and there is one forward), fp16x3 convolutions throughout."""
        convs = [m for m in self.postnet.layers if hasattr(m, "conv")]
        return (ops.TWIN_POSTNET and self.training and melspec.is_cuda and len(convs) > 0 and
                all(ops._fwd_h3(c.conv.weight.shape[2] * c.conv.weight.shape[1], c.conv.weight.shape[0], c.conv.weight.shape[1])
                    for c in convs) and 2 * melspec.shape[0] * melspec.shape[1] * max(c.conv.weight.shape[0] for c in convs) * 4 < (1 << 31))

    def encode_twin(self, phoneme: Tensor, phoneme_lens: Tensor):
        """-> (memory of a forward WITH grad, memory of a forward without) from ONE pass over a batch of 2 B.  The reference's
        training_step encodes the same phonemes twice -- under no_grad for the scheduled-sampling prediction and with grad
        (lightning_module.py:53-59,77) --, each time with fresh dropout masks and with the pre-net's BatchNorm in training
        mode.  The encoder is row-parallel, so both encodes run as one batch: the grad forward's utterances first, the no-grad
        forward's behind them (`ops._twin`); autograd sees the first half only, BatchNorm statistics (and their two updates of
        the running statistics, no-grad forward first) stay per forward, dropout masks are independent (one stream over 2 B
        utterances).  Half the launches and twice the rows per launch for the whole encoder side of the step."""
        B = phoneme.size(0)
        ids_g, _ = ops.twin_pair(torch.cat([phoneme, phoneme], dim=0))
        lens_g, _ = ops.twin_pair(torch.cat([phoneme_lens, phoneme_lens], dim=0).to(torch.int64))
        mem = self.encode(ids_g, lens_g)
        full = ops._twin(mem)
        if full is None:
            raise RuntimeError("encode_twin: the encoder dropped the twin batch")
        mem_ng = full[B:].detach()
        am = getattr(mem, "_ttts_amax", None)
        if am is not None:
            mem_ng._ttts_amax = am           # (published over the whole buffer: a bound for either half)
        return mem, mem_ng

    def forward(self, phoneme: Tensor, melspec: Tensor, phoneme_lens: Tensor, melspec_lens: Tensor,
                need_alignments: bool = True, need_stop: bool = True, memory: Tensor = None,
                postnet_twin: "ops.PostnetTwin" = None) -> dict:
        """
        `need_alignments=False` (an extension; the reference always returns them) skips writing the per-head
        cross-attention maps -- 267 MB per forward at batch 64 -- for callers that only want the mels, e.g. the
        no-grad first forward of `training_step`.  `alignments` is then a list of None.
        `memory` (an extension): the encoder output of `phoneme`, when the caller has it already (`encode_twin`: the two forwards
        of a training step encoded as one batch); None: encoded here.
        `postnet_twin` (an extension; `training_step` alone passes it, the same object to both of its forwards): the no-grad
        forward leaves its prediction there and returns `post_melspec` None WITHOUT running the post-net; the grad forward runs
        the post-net once over both predictions (ops.PostnetTwin: the BatchNorm running statistics receive both forwards'
        updates, in the reference's order).
        `need_stop=False` (no-grad only; same caller): `pred_stop` is None -- the stop head is stateless and its logits have no
        reader there (reference lightning_module.py:53-59 keeps `pred_melspec` alone).  The post-net still runs: its BatchNorm
        running statistics are updated by that forward too.

        Args:
          - phoneme (B, T_phon) int64, melspec (B, T_mel, n_mels) fp32, phoneme_lens / melspec_lens (B,) int64,
            all on the HIP device.
        Returns dict with pred_melspec, post_melspec (B,T_mel,n_mels), pred_stop (B,T_mel) logits and
        alignments: list of per-layer (B, heads, T_mel, T_phon) post-dropout cross-attention weights.
        """
        if phoneme.dim() != 2 or melspec.dim() != 3 or melspec.size(-1) != self.n_mels:
            raise ValueError("TransformerTTS.forward: expected phoneme (B,Tp) and melspec (B,Tm,n_mels)")
        if phoneme_lens.device != phoneme.device or melspec_lens.device != melspec.device:
            raise ValueError("TransformerTTS.forward: lengths must live on the same device as the batch")
        phoneme_lens = phoneme_lens.to(torch.int64)
        melspec_lens = melspec_lens.to(torch.int64)
        if memory is None:
            memory = self.encode(phoneme, phoneme_lens)
        elif memory.dim() != 3 or memory.size(0) != phoneme.size(0) or memory.size(1) != phoneme.size(1):
            raise ValueError("TransformerTTS.forward: `memory` must be the encoder output of these phonemes")
        tgt = self.pe(self.dec_prenet(melspec, shift_right=True))
        tgt_out, alignments = self.decoder(tgt, memory, tgt_is_causal=True, memory_is_causal=False,
                                           tgt_lens=melspec_lens, memory_lens=phoneme_lens,
                                           need_alignments=need_alignments)
        first_of_pair = postnet_twin is not None and postnet_twin.full is None
        pred_melspec, pred_stop = ops.heads(tgt_out, self.linear1.linear.weight, self.linear1.linear.bias,
                                            self.linear2.linear.weight, self.linear2.linear.bias,
                                            need_stop=need_stop or torch.is_grad_enabled(), box=postnet_twin)
        if first_of_pair:          # its prediction waits in the box for the second forward's post-net pass
            return {'pred_melspec': pred_melspec, 'post_melspec': None, 'pred_stop': pred_stop, 'alignments': alignments}
        # three consumers of the prediction (the loss, the post-net, its residual): one handle each
        pred_melspec, pred_in, pred_res = ops.fanout(pred_melspec, 3)
        post_melspec = ops.AddFn.apply(self.postnet(pred_in), pred_res)
        return {
            'pred_melspec': pred_melspec,
            'post_melspec': post_melspec,
            'pred_stop': pred_stop,
            'alignments': alignments,
        }

    @torch.no_grad()
    def inference(self, phoneme: Tensor, phoneme_lens: Tensor, max_len: int = 1500, stop_threshold: float = 0.5,
                  use_kv_cache: bool = True) -> dict:
        """Greedy autoregressive decoding with the reference's semantics (model/model.py:323-394): eval mode, the
        encoder runs WITHOUT a padding mask, cross-attention masks padded phonemes, stop when sigmoid(stop) >= threshold
        for every item, post-net once at the end.

        The reference re-runs pre-net + the whole decoder over all frames so far at every step (O(T^3) overall).
        `use_kv_cache=True` (default) computes the same values incrementally: each step
        projects only the newest frame, appends its self-attention K/V to per-layer caches and attends with one query
        row; the cross-attention K/V of the encoder memory are projected once.  `use_kv_cache=False` keeps the
        reference's recomputation (used as a cross-check in the tests)."""
        B = phoneme.size(0)
        self.eval()
        dev = phoneme.device
        full = torch.full((B,), phoneme.size(1), dtype=torch.int64, device=dev)
        memory = self.encode(phoneme, full)
        phoneme_lens = phoneme_lens.to(torch.int64)
        ys = torch.zeros(B, max_len, self.n_mels, device=dev)     # ys[:, t] = frame fed at step t (frame 0 = go)
        stops = []
        n = 0
        heads = (self.linear1.linear.weight, self.linear1.linear.bias, self.linear2.linear.weight, self.linear2.linear.bias)
        if use_kv_cache and memory.size(-1) // self.decoder.layers[0].self_attn.num_heads != 64:
            use_kv_cache = False          # the incremental kernels read 64-wide heads in place; others take the recompute loop
        if use_kv_cache:
            d = memory.size(-1)
            layers = list(self.decoder.layers)
            H = layers[0].self_attn.num_heads
            # encoder memory K/V per layer, once; self-attention K/V caches (B, max_len, 2d) filled one row per step
            mem_kv = [ops.linear(memory, ops.param_rows(l.multihead_attn.in_proj_weight, d, 3 * d),
                                 ops.param_rows(l.multihead_attn.in_proj_bias, d, 3 * d), publish_amax=True) for l in layers]
            cache = [torch.zeros(B, max_len, 2 * d, device=dev) for _ in layers]
            # running maxima of everything the self-attention in-projections have produced so far (the K/V caches hold a
            # subset of it): the same zeroed slots are handed to every step's GEMM, whose atomic maxima accumulate
            run_am = [ops._amax_slots(dev, True) for _ in layers]
            Tp = memory.size(1)
            for t in range(1, max_len):
                x = self.dec_prenet(ys[:, t - 1:t].contiguous())
                x = ops.posenc(x, self.pe.pe[t - 1:t], self.pe.alpha, 0.0, 0)
                lens_t = torch.full((B,), t, dtype=torch.int64, device=dev)
                for l, kvc, mkv, ram in zip(layers, cache, mem_kv, run_am):
                    sa = l.self_attn
                    qkv = ops.linear(x, sa.in_proj_weight, sa.in_proj_bias, publish_amax=ram)  # (B,1,3d)
                    kvc[:, t - 1] = qkv[:, 0, d:]
                    ctx, _, _ = ops._attn_fwd(ops._off(qkv, 0), ops._off(kvc, 0), ops._off(kvc, d), 3 * d, 2 * d, 2 * d, B, H,
                                              1, max_len, lens_t, False, 0.0, 0, False, ram, ram, ram)
                    x = ops.layer_norm(ops.linear(ctx, sa.out_proj.weight, sa.out_proj.bias, residual=x),
                                       l.norm1.weight, l.norm1.bias, l.norm1.eps)
                    ca = l.multihead_attn
                    q = ops.linear(x, ops.param_rows(ca.in_proj_weight, 0, d), ops.param_rows(ca.in_proj_bias, 0, d),
                                   publish_amax=True)
                    mkv_am = ops._amax(mkv)
                    ctx, _, _ = ops._attn_fwd(ops._off(q, 0), ops._off(mkv, 0), ops._off(mkv, d), d, 2 * d, 2 * d, B, H, 1, Tp,
                                              phoneme_lens, False, 0.0, 0, False, ops._amax(q), mkv_am, mkv_am)
                    x = ops.layer_norm(ops.linear(ctx, ca.out_proj.weight, ca.out_proj.bias, residual=x),
                                       l.norm2.weight, l.norm2.bias, l.norm2.eps)
                    hdn = ops.linear(x, l.linear1.weight, l.linear1.bias, act=ops.ACT_RELU, publish_amax=True)
                    x = ops.layer_norm(ops.linear(hdn, l.linear2.weight, l.linear2.bias, residual=x),
                                       l.norm3.weight, l.norm3.bias, l.norm3.eps)
                mel, stop = ops.heads(x, *heads)
                ys[:, t] = mel[:, 0]
                stops.append(stop)
                n = t
                if bool((torch.sigmoid(stop) >= stop_threshold).all()):
                    break
        else:
            for t in range(1, max_len):
                cur = ys[:, :t].contiguous()
                tgt = self.pe(self.dec_prenet(cur))
                lens_t = torch.full((B,), t, dtype=torch.int64, device=dev)
                out, _ = self.decoder(tgt, memory, tgt_is_causal=True, tgt_lens=lens_t,
                                      memory_lens=phoneme_lens, need_alignments=False)
                last = out[:, -1:, :].contiguous()
                mel, stop = ops.heads(last, *heads)
                ys[:, t] = mel[:, 0]
                stops.append(stop)
                n = t
                if bool((torch.sigmoid(stop) >= stop_threshold).all()):
                    break
        pred_melspec = ys[:, 1:n + 1].contiguous()
        post_melspec = ops.AddFn.apply(self.postnet(pred_melspec), pred_melspec)
        return {'pred_melspec': pred_melspec, 'post_melspec': post_melspec, 'pred_stop': torch.stack(stops, dim=1)}
