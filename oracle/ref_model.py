"""CPU restatement of the teacher-forced Transformer-TTS path (test infrastructure).

Functional, explicit stock-torch code over a flat state dict (keys as in
`oracle.spec.state_spec`).  Each function cites the reference lines it follows
(paths relative to /root/reference; `torch/...` = the third-party PyTorch the
reference delegates its arithmetic to).  Gradients come from torch autograd on
these same ops.  Pinned against the live reference by tests/golden/*.npz.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


class relu_gates:
    """Test instrument for the ReLU sites of the training forward (FFN blocks, decoder pre-net), visited in call order.

    `with relu_gates() as rec:` records, per site, the pre-activation tensor (`rec.pre`) -- so a test can see which units
    sit within rounding distance of zero.  `with relu_gates(gates=[...])` REPLACES relu(x) by x * gate with the given
    0/1 tensors, one per site: the forward/backward is then evaluated under another implementation's gating decisions.
    A ReLU is the only discontinuous operation of the path; two correct evaluations in different precisions may gate a
    unit whose pre-activation is ~1e-7 differently, and that is the whole difference between their gradients
    (tests/test_hip_model.py::test_gradient_gap_is_relu_gate_flips)."""
    active = None

    def __init__(self, gates=None):
        self.gates, self.pre, self.i = gates, [], 0

    def __enter__(self):
        relu_gates.active = self
        return self

    def __exit__(self, *exc):
        relu_gates.active = None


def _relu(x: Tensor) -> Tensor:
    g = relu_gates.active
    if g is None:
        return F.relu(x)
    if g.gates is None:
        g.pre.append(x.detach())
        return F.relu(x)
    gate = g.gates[g.i].to(dtype=x.dtype).reshape(x.shape)
    g.i += 1
    return x * gate


def _drop(x: Tensor, p: float, on: bool) -> Tensor:
    return F.dropout(x, p, training=True) if (on and p > 0.0) else x


def conv_norm_bn(sd, prefix: str, x: Tensor, training: bool, update_bn: bool) -> Tensor:
    """ConvNormBN.forward, model/module.py:28-33: (B,T,Cin)->conv1d(pad=(k-1)//2)->BatchNorm1d->(B,T,Cout).

    Train mode uses batch statistics over all B*T positions (padding included) and
    updates running stats with momentum 0.1 / unbiased variance (torch BatchNorm1d).
    """
    w = sd[f"{prefix}.conv.weight"]
    k = w.shape[-1]
    y = F.conv1d(x.transpose(1, 2), w, sd[f"{prefix}.conv.bias"], padding=(k - 1) // 2)
    rm, rv = sd[f"{prefix}.bn.running_mean"], sd[f"{prefix}.bn.running_var"]
    if training:
        if update_bn:
            y = F.batch_norm(y, rm, rv, sd[f"{prefix}.bn.weight"], sd[f"{prefix}.bn.bias"],
                             training=True, momentum=0.1, eps=1e-5)
            sd[f"{prefix}.bn.num_batches_tracked"] += 1
        else:
            y = F.batch_norm(y, None, None, sd[f"{prefix}.bn.weight"], sd[f"{prefix}.bn.bias"],
                             training=True, momentum=0.1, eps=1e-5)
    else:
        y = F.batch_norm(y, rm, rv, sd[f"{prefix}.bn.weight"], sd[f"{prefix}.bn.bias"],
                         training=False, eps=1e-5)
    return y.transpose(1, 2)


def positional_encoding(sd, x: Tensor, p: float, drop_on: bool) -> Tensor:
    """PositionalEncoding.forward, model/model.py:91-97: x + alpha * pe[:T]; Dropout(0.1)."""
    x = x + sd["pe.alpha"] * sd["pe.pe"][: x.size(1), :].unsqueeze(0)
    return _drop(x, p, drop_on)


def multi_head_attention(sd, prefix: str, xq: Tensor, xkv: Tensor, n_head: int,
                         key_lens: Tensor, causal: bool, p: float, drop_on: bool):
    """nn.MultiheadAttention as the reference calls it (model/layers.py:68-73; torch
    `F.multi_head_attention_forward`, torch/nn/functional.py:6206+): packed in-proj,
    q scaled by sqrt(1/head_dim) *before* q.k^T (:6578), additive -inf mask from the key
    padding mask (and the causal mask for decoder self-attention, merged at :6566), softmax
    over keys, dropout on the weights (:6592), weights @ v, out-proj.  Returns the per-head,
    un-averaged, post-dropout weights (B,H,Tq,Tk) next to the output.
    """
    B, Tq, d = xq.shape
    Tk = xkv.size(1)
    hd = d // n_head
    w_in, b_in = sd[f"{prefix}.in_proj_weight"], sd[f"{prefix}.in_proj_bias"]
    q = F.linear(xq, w_in[:d], b_in[:d])
    k = F.linear(xkv, w_in[d:2 * d], b_in[d:2 * d])
    v = F.linear(xkv, w_in[2 * d:], b_in[2 * d:])
    q = q.view(B, Tq, n_head, hd).transpose(1, 2) * math.sqrt(1.0 / hd)
    k = k.view(B, Tk, n_head, hd).transpose(1, 2)
    v = v.view(B, Tk, n_head, hd).transpose(1, 2)
    s = q @ k.transpose(-1, -2)                                    # (B,H,Tq,Tk)
    dead = torch.arange(Tk).view(1, 1, 1, Tk) >= key_lens.view(B, 1, 1, 1)
    if causal:
        dead = dead | torch.triu(torch.ones(Tq, Tk, dtype=torch.bool), diagonal=1).view(1, 1, Tq, Tk)
    s = s.masked_fill(dead, float("-inf"))
    a = torch.softmax(s, dim=-1)
    a = _drop(a, p, drop_on)
    o = (a @ v).transpose(1, 2).reshape(B, Tq, d)
    o = F.linear(o, sd[f"{prefix}.out_proj.weight"], sd[f"{prefix}.out_proj.bias"])
    return o, a


def _ffn(sd, prefix: str, x: Tensor, p: float, drop_on: bool) -> Tensor:
    """torch `_ff_block` (torch/nn/modules/transformer.py:980-982): W2 . Drop(relu(W1 x)), then Drop."""
    h = _drop(_relu(F.linear(x, sd[f"{prefix}.linear1.weight"], sd[f"{prefix}.linear1.bias"])), p, drop_on)
    return _drop(F.linear(h, sd[f"{prefix}.linear2.weight"], sd[f"{prefix}.linear2.bias"]), p, drop_on)


def _ln(sd, prefix: str, x: Tensor) -> Tensor:
    return F.layer_norm(x, (x.size(-1),), sd[f"{prefix}.weight"], sd[f"{prefix}.bias"], 1e-5)


def encoder_layer(sd, prefix, x, n_head, lens, p, drop_on):
    """Post-norm nn.TransformerEncoderLayer (torch/nn/modules/transformer.py:951-956)."""
    sa, _ = multi_head_attention(sd, f"{prefix}.self_attn", x, x, n_head, lens, False, p, drop_on)
    x = _ln(sd, f"{prefix}.norm1", x + _drop(sa, p, drop_on))
    x = _ln(sd, f"{prefix}.norm2", x + _ffn(sd, prefix, x, p, drop_on))
    return x


def decoder_layer(sd, prefix, x, mem, n_head, mel_lens, ph_lens, p, drop_on):
    """TransformerDecoderLayer.forward post-norm branch, model/layers.py:46-50."""
    sa, _ = multi_head_attention(sd, f"{prefix}.self_attn", x, x, n_head, mel_lens, True, p, drop_on)
    x = _ln(sd, f"{prefix}.norm1", x + _drop(sa, p, drop_on))
    ca, align = multi_head_attention(sd, f"{prefix}.multihead_attn", x, mem, n_head, ph_lens, False, p, drop_on)
    x = _ln(sd, f"{prefix}.norm2", x + _drop(ca, p, drop_on))
    x = _ln(sd, f"{prefix}.norm3", x + _ffn(sd, prefix, x, p, drop_on))
    return x, align


def oracle_forward(sd: Dict[str, Tensor], cfg: dict, phoneme: Tensor, melspec: Tensor,
                   phoneme_lens: Tensor, melspec_lens: Tensor, training: bool = True,
                   dropout: bool = False, update_bn: bool = True) -> Dict[str, object]:
    """TransformerTTS.forward, model/model.py:260-320 (shapes in SURVEY.md section 3.3)."""
    # go-frame shift (:278-279)
    tgt_in = torch.cat((torch.zeros_like(melspec[:, :1, :]), melspec[:, :-1, :]), dim=1)
    drop_on = dropout and training

    # encoder side (:288-292)
    x = F.embedding(phoneme, sd["emb.weight"])
    for i in range(cfg["encoder_prenet_n_layers"]):          # EncoderPreNet, :38-45 (no activation)
        x = conv_norm_bn(sd, f"enc_prenet.layers.{2 * i}", x, training, update_bn)
        x = _drop(x, cfg["encoder_prenet_dropout"], drop_on)
    x = F.linear(x, sd["enc_prenet.linear.linear.weight"], sd["enc_prenet.linear.linear.bias"])
    x = positional_encoding(sd, x, 0.1, drop_on)
    for i in range(cfg["encoder_n_layers"]):
        x = encoder_layer(sd, f"encoder.layers.{i}", x, cfg["encoder_n_head"], phoneme_lens,
                          cfg["encoder_dropout"], drop_on)
    memory = x

    # decoder side (:297-306); DecoderPreNet :65-66 has fixed dropout 0.5
    y = _drop(_relu(F.linear(tgt_in, sd["dec_prenet.linear1.linear.weight"],
                              sd["dec_prenet.linear1.linear.bias"])), 0.5, drop_on)
    y = _drop(_relu(F.linear(y, sd["dec_prenet.linear2.linear.weight"],
                              sd["dec_prenet.linear2.linear.bias"])), 0.5, drop_on)
    y = positional_encoding(sd, y, 0.1, drop_on)
    aligns: List[Tensor] = []
    for i in range(cfg["decoder_n_layers"]):
        y, a = decoder_layer(sd, f"decoder.layers.{i}", y, memory, cfg["decoder_n_head"],
                             melspec_lens, phoneme_lens, cfg["decoder_dropout"], drop_on)
        aligns.append(a)

    # heads and post-net (:309-313; PostNet :113-126,133-135)
    pred = F.linear(y, sd["linear1.linear.weight"], sd["linear1.linear.bias"])
    z = pred
    n_post = cfg["postnet_n_layers"]
    for i in range(n_post):
        z = conv_norm_bn(sd, f"postnet.layers.{3 * i}", z, training, update_bn)
        if i < n_post - 1:
            z = torch.tanh(z)
        z = _drop(z, cfg["postnet_dropout"], drop_on)
    post = z + pred
    stop = F.linear(y, sd["linear2.linear.weight"], sd["linear2.linear.bias"]).squeeze(-1)
    return {"pred_melspec": pred, "post_melspec": post, "pred_stop": stop, "alignments": aligns}


def oracle_loss(outputs: Dict[str, Tensor], mel: Tensor, lengths: Tensor,
                stop_weight: float = 8.0) -> Dict[str, Tensor]:
    """TransformerTTSLoss.forward, loss.py:15-55: MSE over valid frames (x2, post weighted 0.5)
    + BCE-with-logits on the stop gate (1 at the last valid frame) with pos_weight."""
    pred, post, stop = outputs["pred_melspec"], outputs["post_melspec"], outputs["pred_stop"]
    B, T, C = pred.shape
    pos = torch.arange(T).unsqueeze(0).expand(B, T)
    valid = pos < lengths.unsqueeze(1)
    gate = (pos == (lengths.unsqueeze(1) - 1)).float()
    n = valid.sum() * C
    vm = valid.unsqueeze(-1).float()
    pred_l = (((pred - mel) ** 2) * vm).sum() / n
    post_l = (((post - mel) ** 2) * vm).sum() / n
    bce = F.binary_cross_entropy_with_logits(stop, gate, reduction="none",
                                             pos_weight=torch.tensor(stop_weight))
    stop_l = (bce * valid.float()).sum() / valid.sum()
    return {"total": pred_l + 0.5 * post_l + stop_l, "pred_mel": pred_l, "post_mel": post_l, "stop": stop_l}


def teacher_forcing_ratio(epoch: int, total_epochs: int = 300, mode: str = "cosine",
                          warmup_epochs: int = 10, cycles: int = 1, value: float = 1.0) -> float:
    """get_teacher_forcing_ratio, utils/util.py:54-92."""
    if epoch < warmup_epochs:
        return 1.0
    e = epoch - warmup_epochs
    tot = max(total_epochs - warmup_epochs, 1)
    if mode == "cosine":
        return max(min(0.5 * math.cos(math.pi * e * cycles / tot) + 0.5, 1.0), 0.5)
    if mode == "linear":
        return max(1.0 - e / tot, 0.05)
    if mode == "constant":
        return value
    raise ValueError(f"Unsupported teacher forcing mode: {mode}")


def noam_lambda(d_model: int, warmup_steps: int):
    """get_noam_scheduler, utils/util.py:42-49."""
    def f(step: int) -> float:
        step = max(step, 1)
        return (d_model ** -0.5) * min(step ** -0.5, step * (warmup_steps ** -1.5))
    return f


def scheduled_sampling_mix(pred: Tensor, mel: Tensor, mel_lens: Tensor, p_tf: float,
                           seed_u: Optional[Tensor] = None, l_bar: int = 8) -> Tensor:
    """block_mask + apply_teacher_forcing, utils/util.py:103-120.  `seed_u` is the (B,1,T)
    uniform draw (`torch.rand` at :108); drawn here from the global CPU generator when None."""
    B, T, _ = pred.shape
    if seed_u is None:
        seed_u = torch.rand(B, 1, T)
    seed = (seed_u < (1 - p_tf)).float()
    dil = F.max_pool1d(seed, kernel_size=l_bar, stride=1, padding=l_bar // 2)
    mask = dil.squeeze(1).bool().unsqueeze(-1)[:, :T, :]
    mixed = torch.where(mask, pred.detach(), mel)
    valid = torch.arange(T).unsqueeze(0) < mel_lens.unsqueeze(1)
    return mixed * valid.unsqueeze(-1)


def oracle_training_step(sd, cfg, batch, epoch: int = 0, num_epochs: int = 300,
                         tf_mode: str = "linear", dropout: bool = False,
                         stop_weight: float = 8.0, seed_u: Optional[Tensor] = None):
    """Arithmetic of LightningModule.training_step, lightning_module.py:45-86: no-grad forward
    (train mode: BN stats update) -> scheduled-sampling mix -> forward -> loss.  Returns the loss
    dict and the second forward's outputs; the caller runs `.backward()` on loss['total']."""
    ph, mel, pl, ml = batch["phoneme"], batch["melspec"], batch["phoneme_lens"], batch["melspec_lens"]
    with torch.no_grad():
        pred = oracle_forward(sd, cfg, ph, mel, pl, ml, training=True, dropout=dropout)["pred_melspec"]
    p_tf = teacher_forcing_ratio(epoch + 1, num_epochs, tf_mode, cycles=1)
    mixed = scheduled_sampling_mix(pred, mel, ml, p_tf, seed_u)
    out = oracle_forward(sd, cfg, ph, mixed, pl, ml, training=True, dropout=dropout)
    loss = oracle_loss(out, mel, ml, stop_weight)
    return loss, out, mixed


@torch.no_grad()
def oracle_inference(sd, cfg, phoneme, phoneme_lens, max_len: int = 1500, stop_threshold: float = 0.5):
    """TransformerTTS.inference, model/model.py:323-394: eval mode; the encoder is called WITHOUT a padding mask
    (:346-348), each step re-runs pre-net + PE + the whole decoder over the frames so far (:354-374) with the causal +
    length masks of _get_mask and the memory key-padding mask; stops when sigmoid(stop) >= threshold for all items."""
    B = phoneme.size(0)
    dt = sd["emb.weight"].dtype
    x = F.embedding(phoneme, sd["emb.weight"])
    for i in range(cfg["encoder_prenet_n_layers"]):
        x = conv_norm_bn(sd, f"enc_prenet.layers.{2 * i}", x, False, False)
    x = F.linear(x, sd["enc_prenet.linear.linear.weight"], sd["enc_prenet.linear.linear.bias"])
    x = positional_encoding(sd, x, 0.1, False)
    full = torch.full((B,), phoneme.size(1), dtype=torch.long)
    for i in range(cfg["encoder_n_layers"]):
        x = encoder_layer(sd, f"encoder.layers.{i}", x, cfg["encoder_n_head"], full, cfg["encoder_dropout"], False)
    memory = x
    ys = [torch.zeros(B, 1, cfg["n_mels"], dtype=dt)]
    stops = []
    for t in range(1, max_len):
        y = torch.cat(ys, dim=1)
        y = F.relu(F.linear(y, sd["dec_prenet.linear1.linear.weight"], sd["dec_prenet.linear1.linear.bias"]))
        y = F.relu(F.linear(y, sd["dec_prenet.linear2.linear.weight"], sd["dec_prenet.linear2.linear.bias"]))
        y = positional_encoding(sd, y, 0.1, False)
        lens_t = torch.full((B,), t, dtype=torch.long)
        for i in range(cfg["decoder_n_layers"]):
            y, _ = decoder_layer(sd, f"decoder.layers.{i}", y, memory, cfg["decoder_n_head"], lens_t, phoneme_lens,
                                 cfg["decoder_dropout"], False)
        cur = y[:, -1:, :]
        mel = F.linear(cur, sd["linear1.linear.weight"], sd["linear1.linear.bias"])
        stop = F.linear(cur, sd["linear2.linear.weight"], sd["linear2.linear.bias"]).squeeze(-1)
        ys.append(mel)
        stops.append(stop)
        if bool((torch.sigmoid(stop) >= stop_threshold).all()):
            break
    pred = torch.cat(ys[1:], dim=1)
    z = pred
    n_post = cfg["postnet_n_layers"]
    for i in range(n_post):
        z = conv_norm_bn(sd, f"postnet.layers.{3 * i}", z, False, False)
        if i < n_post - 1:
            z = torch.tanh(z)
    return {"pred_melspec": pred, "post_melspec": z + pred, "pred_stop": torch.stack(stops, dim=1)}
