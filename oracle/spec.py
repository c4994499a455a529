"""State-dict contract and seed-defined weights (test infrastructure).

`state_spec(cfg)` restates, from the constructor arguments alone, which tensors
`TransformerTTS(**cfg).state_dict()` holds in the reference
(/root/reference/model/model.py:139-226; module.py:18-19,45; layers.py:11-27;
torch nn.TransformerEncoderLayer / nn.MultiheadAttention parameter names).
`fill_state(cfg, seed)` regenerates every tensor from a `torch.Generator`, in
sorted-key order, so that weights never have to be stored or shipped: the real
reference (in the build container), the oracle and the HIP path all load the
same values from the same seed.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import torch

# Reference config.yaml:25-42 ("base"), a tiny config for full-tensor goldens,
# and BASELINE.json configs[4] ("scaled": d_model 512, 6+6 layers, 8 heads).
CONFIGS = {
    "base": dict(
        encoder_prenet_n_layers=3, encoder_prenet_in_channel=256,
        encoder_prenet_out_channel=256, encoder_prenet_kernel_size=5,
        encoder_prenet_dropout=0.5, encoder_n_layers=3, encoder_n_head=4,
        encoder_d_ffn=1024, encoder_dropout=0.1, decoder_n_layers=3,
        decoder_n_head=4, decoder_d_ffn=1024, decoder_dropout=0.1,
        postnet_n_layers=5, postnet_kernel_size=5, postnet_dropout=0.5,
        d_model=256, n_phon=100, n_mels=80),
    "tiny": dict(
        encoder_prenet_n_layers=2, encoder_prenet_in_channel=128,
        encoder_prenet_out_channel=128, encoder_prenet_kernel_size=5,
        encoder_prenet_dropout=0.5, encoder_n_layers=1, encoder_n_head=2,
        encoder_d_ffn=256, encoder_dropout=0.1, decoder_n_layers=2,
        decoder_n_head=2, decoder_d_ffn=256, decoder_dropout=0.1,
        postnet_n_layers=3, postnet_kernel_size=5, postnet_dropout=0.5,
        d_model=128, n_phon=30, n_mels=16),
    # SURVEY.md section 8c's tiny-golden plan: d_model 32, 2 heads (head_dim 16), 1+1 layers
    "micro": dict(
        encoder_prenet_n_layers=2, encoder_prenet_in_channel=32,
        encoder_prenet_out_channel=32, encoder_prenet_kernel_size=5,
        encoder_prenet_dropout=0.5, encoder_n_layers=1, encoder_n_head=2,
        encoder_d_ffn=64, encoder_dropout=0.1, decoder_n_layers=1,
        decoder_n_head=2, decoder_d_ffn=64, decoder_dropout=0.1,
        postnet_n_layers=3, postnet_kernel_size=5, postnet_dropout=0.5,
        d_model=32, n_phon=20, n_mels=16),
    # heads wider than the kernels' 64 columns (head_dim 128): the tensor-algebra attention path
    "tiny1h": dict(
        encoder_prenet_n_layers=2, encoder_prenet_in_channel=128,
        encoder_prenet_out_channel=128, encoder_prenet_kernel_size=5,
        encoder_prenet_dropout=0.5, encoder_n_layers=1, encoder_n_head=1,
        encoder_d_ffn=256, encoder_dropout=0.1, decoder_n_layers=2,
        decoder_n_head=1, decoder_d_ffn=256, decoder_dropout=0.1,
        postnet_n_layers=3, postnet_kernel_size=5, postnet_dropout=0.5,
        d_model=128, n_phon=30, n_mels=16),
    "scaled": dict(
        encoder_prenet_n_layers=3, encoder_prenet_in_channel=512,
        encoder_prenet_out_channel=512, encoder_prenet_kernel_size=5,
        encoder_prenet_dropout=0.5, encoder_n_layers=6, encoder_n_head=8,
        encoder_d_ffn=2048, encoder_dropout=0.1, decoder_n_layers=6,
        decoder_n_head=8, decoder_d_ffn=2048, decoder_dropout=0.1,
        postnet_n_layers=5, postnet_kernel_size=5, postnet_dropout=0.5,
        d_model=512, n_phon=100, n_mels=80),
}


def model_config(name: str = "base") -> dict:
    return dict(CONFIGS[name])


def _conv_bn(spec, prefix, cin, cout, k):
    spec[f"{prefix}.conv.weight"] = (cout, cin, k)
    spec[f"{prefix}.conv.bias"] = (cout,)
    spec[f"{prefix}.bn.weight"] = (cout,)
    spec[f"{prefix}.bn.bias"] = (cout,)
    spec[f"{prefix}.bn.running_mean"] = (cout,)
    spec[f"{prefix}.bn.running_var"] = (cout,)
    spec[f"{prefix}.bn.num_batches_tracked"] = ()


def _mha(spec, prefix, d):
    spec[f"{prefix}.in_proj_weight"] = (3 * d, d)
    spec[f"{prefix}.in_proj_bias"] = (3 * d,)
    spec[f"{prefix}.out_proj.weight"] = (d, d)
    spec[f"{prefix}.out_proj.bias"] = (d,)


def _ffn_norms(spec, prefix, d, dff, n_norm):
    spec[f"{prefix}.linear1.weight"] = (dff, d)
    spec[f"{prefix}.linear1.bias"] = (dff,)
    spec[f"{prefix}.linear2.weight"] = (d, dff)
    spec[f"{prefix}.linear2.bias"] = (d,)
    for i in range(1, n_norm + 1):
        spec[f"{prefix}.norm{i}.weight"] = (d,)
        spec[f"{prefix}.norm{i}.bias"] = (d,)


def state_spec(cfg: dict) -> "OrderedDict[str, tuple]":
    """key -> shape for every state-dict tensor (parameters and buffers)."""
    d = cfg["d_model"]
    n_mels = cfg.get("n_mels", 80)
    n_phon = cfg.get("n_phon", 100)
    spec: "OrderedDict[str, tuple]" = OrderedDict()
    spec["emb.weight"] = (n_phon, d)
    # EncoderPreNet: ModuleList [ConvNormBN, Dropout] * n  -> indices 0,2,4,...
    cin, cout, k = (cfg["encoder_prenet_in_channel"], cfg["encoder_prenet_out_channel"],
                    cfg["encoder_prenet_kernel_size"])
    for i in range(cfg["encoder_prenet_n_layers"]):
        _conv_bn(spec, f"enc_prenet.layers.{2 * i}", cin if i == 0 else cout, cout, k)
    spec["enc_prenet.linear.linear.weight"] = (cout, cout)
    spec["enc_prenet.linear.linear.bias"] = (cout,)
    spec["dec_prenet.linear1.linear.weight"] = (d, n_mels)
    spec["dec_prenet.linear1.linear.bias"] = (d,)
    spec["dec_prenet.linear2.linear.weight"] = (d, d)
    spec["dec_prenet.linear2.linear.bias"] = (d,)
    spec["pe.alpha"] = (1,)
    spec["pe.pe"] = (5000, d)
    for i in range(cfg["encoder_n_layers"]):
        p = f"encoder.layers.{i}"
        _mha(spec, f"{p}.self_attn", d)
        _ffn_norms(spec, p, d, cfg["encoder_d_ffn"], 2)
    for i in range(cfg["decoder_n_layers"]):
        p = f"decoder.layers.{i}"
        _mha(spec, f"{p}.self_attn", d)
        _mha(spec, f"{p}.multihead_attn", d)
        _ffn_norms(spec, p, d, cfg["decoder_d_ffn"], 3)
    # PostNet: [ConvNormBN, Tanh, Dropout] * (n-1) + [ConvNormBN, Dropout] -> 0,3,6,...
    n_post, kp = cfg["postnet_n_layers"], cfg["postnet_kernel_size"]
    for i in range(n_post):
        ci = n_mels if i == 0 else d
        co = n_mels if i == n_post - 1 else d
        _conv_bn(spec, f"postnet.layers.{3 * i}", ci, co, kp)
    spec["linear1.linear.weight"] = (n_mels, d)
    spec["linear1.linear.bias"] = (n_mels,)
    spec["linear2.linear.weight"] = (1, d)
    spec["linear2.linear.bias"] = (1,)
    return spec


def sinusoid_table(max_len: int, d_model: int) -> torch.Tensor:
    """pe buffer (reference model/model.py:80-85): sin on even, cos on odd channels."""
    pos = torch.arange(0, max_len, dtype=torch.float32).unsqueeze(1)
    div = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float32) * (-math.log(10000.0)) / d_model)
    pe = torch.zeros(max_len, d_model, dtype=torch.float32)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


def fill_state(cfg: dict, seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    """Seed-defined values for every state-dict tensor, drawn in sorted-key order.

    Matrices get N(0, 1/fan_in)-scaled entries (keeps activations O(1) through the
    stack), biases N(0, 0.05), norm gains U(0.8, 1.2), BN running_var U(0.5, 1.5).
    """
    spec = state_spec(cfg)
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    vals = {}
    for key in sorted(spec):
        shape = spec[key]
        if key == "pe.pe":
            vals[key] = sinusoid_table(shape[0], shape[1])
        elif key.endswith("num_batches_tracked"):
            vals[key] = torch.zeros((), dtype=torch.int64)
        elif key == "pe.alpha":
            vals[key] = torch.full((1,), 1.0) + 0.1 * torch.randn(1, generator=g)
        elif key.endswith("running_var"):
            vals[key] = 0.5 + torch.rand(shape, generator=g)
        elif key.endswith("running_mean"):
            vals[key] = 0.1 * torch.randn(shape, generator=g)
        elif ".bn.weight" in key or (".norm" in key and key.endswith(".weight")):
            vals[key] = 0.8 + 0.4 * torch.rand(shape, generator=g)
        elif key.endswith("bias"):
            vals[key] = 0.05 * torch.randn(shape, generator=g)
        elif key == "emb.weight":
            vals[key] = torch.randn(shape, generator=g)
        else:  # weight matrices / conv kernels
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            vals[key] = torch.randn(shape, generator=g) / math.sqrt(fan_in)
    for key in spec:  # keep the module's registration order in the returned dict
        out[key] = vals[key].contiguous()
    return out
