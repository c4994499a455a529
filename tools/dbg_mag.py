import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from test_hip_model import _build, _oracle64
from conftest import rel_l2
from oracle import synth_batch, oracle_forward, oracle_loss
from transformertts_amd.loss import TransformerTTSLoss
from transformertts_amd import ops
mode = sys.argv[1] if len(sys.argv) > 1 else ""
if "x6f" in mode: ops.ATTN_FWD_MODE = "x6"
if "x6b" in mode: ops.ATTN_BWD_MODE = "x6"
if "gx6" in mode: ops.BWD_MODE = "x6"; ops.WGRAD_MODE = "x6"
cfg, m = _build("base", 12)
batch = synth_batch(2, 60, 300, cfg["n_mels"], cfg["n_phon"], ragged=True, seed=22)
batch["melspec"] = batch["melspec"] * 1e3
args = [batch[k].to("cuda") for k in ("phoneme", "melspec", "phoneme_lens", "melspec_lens")]
m.train()
out = m(*args)
TransformerTTSLoss(8.0).to("cuda")(out, args[1], args[3])["total"].backward()
sd = _oracle64(cfg, 12)
ref = oracle_forward(sd, cfg, batch["phoneme"], batch["melspec"].double(), batch["phoneme_lens"], batch["melspec_lens"], training=True, dropout=False)
oracle_loss(ref, batch["melspec"].double(), batch["melspec_lens"])["total"].backward()
for name in ("decoder.layers.0.self_attn.in_proj_weight", "decoder.layers.0.self_attn.in_proj_bias"):
    g, r = dict(m.named_parameters())[name].grad, sd[name].grad
    d = 256
    for i, part in enumerate("qkv"):
        gg, rr = g[i * d:(i + 1) * d], r[i * d:(i + 1) * d]
        print(mode, name, part, "rel", rel_l2(gg, rr), "norm ref", float(rr.norm()), "norm err", float((gg.double().cpu() - rr.cpu()).norm()), flush=True)
for name in ("decoder.layers.0.self_attn.out_proj.weight", "decoder.layers.0.norm1.weight", "dec_prenet.linear2.linear.weight", "decoder.layers.1.self_attn.in_proj_weight"):
    print(mode, name, rel_l2(dict(m.named_parameters())[name].grad, sd[name].grad), flush=True)
