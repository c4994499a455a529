#!/usr/bin/env python3
"""Saturated softmax probe (development aid): the base model on melspec x 1000 -- capture the packed in-projection output of
decoder layer 0's self-attention, then run that attention alone (forward + backward with a random dO) against an fp64 torch
reference.  Prints how many rows are exactly one-hot and the error of dq / dk / dv."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import math
import torch
from oracle import synth_batch, model_config, fill_state
from transformertts_amd import ops
from transformertts_amd.model import TransformerTTS

cfg = model_config("base")
m = TransformerTTS(**cfg, device="cuda")
m.load_state_dict(fill_state(cfg, 12), strict=True)
m = m.to("cuda")
for mod in m.modules():
    if isinstance(mod, torch.nn.Dropout):
        mod.p = 0.0
    if hasattr(mod, "dropout") and isinstance(mod.dropout, float):
        mod.dropout = 0.0
batch = synth_batch(2, 60, 300, cfg["n_mels"], cfg["n_phon"], ragged=True, seed=22)
batch["melspec"] = batch["melspec"] * 1e3
args = [batch[k].to("cuda") for k in ("phoneme", "melspec", "phoneme_lens", "melspec_lens")]
cap = []
orig = ops.self_attention
def spy(qkv, lens, n_head, causal, drop_p, seed):
    if causal and not cap:
        cap.append((qkv.detach().clone(), lens.clone(), n_head))
    return orig(qkv, lens, n_head, causal, drop_p, seed)
ops.self_attention = spy
m.train()
m(*args)
ops.self_attention = orig
qkv, lens, H = cap[0]
B, T, d3 = qkv.shape
d = d3 // 3
print("qkv max", float(qkv.abs().max()), "shape", tuple(qkv.shape), "lens", lens.tolist())
x = qkv.clone().requires_grad_(True)
o = ops.self_attention(x, lens, H, True, 0.0, 0)
g = torch.randn_like(o)
o.backward(g)
dx = x.grad
# fp64 reference
xq = qkv.double().cpu().requires_grad_(True)
q, k, v = (xq[..., i * d:(i + 1) * d].view(B, T, H, 64).transpose(1, 2) for i in range(3))
s = (q * math.sqrt(1 / 64)) @ k.transpose(-1, -2)
dead = torch.arange(T).view(1, 1, 1, T) >= lens.cpu().view(B, 1, 1, 1)
dead = dead | torch.triu(torch.ones(T, T, dtype=torch.bool), 1).view(1, 1, T, T)
p = torch.softmax(s.masked_fill(dead, float("-inf")), -1)
ref = (p @ v).transpose(1, 2).reshape(B, T, d)
ref.backward(g.double().cpu())
rg = xq.grad
rel = lambda a, b: float((a.double().cpu() - b).norm() / b.norm())
print("o rel", rel(o.detach(), ref.detach()))
for i, nm in enumerate("qkv"):
    print(f"d{nm}: rel {rel(dx[..., i * d:(i + 1) * d], rg[..., i * d:(i + 1) * d]):.3e}  |ref| {float(rg[..., i * d:(i + 1) * d].norm()):.3e}  |hip| {float(dx[..., i * d:(i + 1) * d].norm()):.3e}")
pm = p.max(-1).values
live = (torch.arange(T).view(1, 1, T) < lens.cpu().view(B, 1, 1)).expand(B, H, T)
print("rows:", int(live.sum()), " exactly one-hot in fp64->fp32:", int(((pm.float() == 1.0) & live).sum()), " p_max < 1-1e-6:", int(((pm < 1 - 1e-6) & live).sum()))
pf = torch.softmax(s.float().masked_fill(dead, float("-inf")), -1)
print("fp32 softmax rows with sum of others == 0:", int((((pf.sum(-1) - pf.max(-1).values) == 0) & live).sum()))
one_m = (p.sum(-1) - pm).clamp_min(0)          # mass outside the largest weight (fp64)
lv = one_m[live]
qs = torch.quantile(lv, torch.tensor([0.0, 0.5, 0.9, 0.99, 1.0], dtype=torch.float64))
print("mass outside the largest weight, live rows: quantiles", [f"{float(x):.2e}" for x in qs])
dq_ref = rg[..., :d].view(B, T, H, 64).transpose(1, 2)          # (B,H,T,64)
rown = dq_ref.norm(dim=-1)
print("dq row norms: max", float(rown.max()), "at", (rown == rown.max()).nonzero()[0].tolist(), " live rows max", float(rown[live].max()),
      " dead (padded query) rows max", float(rown[~live].max()) if (~live).any() else 0.0)
dk_ref = rg[..., d:2 * d].view(B, T, H, 64).transpose(1, 2)
dk_hip = dx[..., d:2 * d].double().cpu().view(B, T, H, 64).transpose(1, 2)
err = (dk_hip - dk_ref).norm(dim=-1)
print("dk: worst key rows (b,h,t):", (err == err.max()).nonzero()[0].tolist(), "err", float(err.max()), "ref norm there", float(dk_ref.norm(dim=-1)[tuple((err == err.max()).nonzero()[0].tolist())]))
print("dk err over live keys only:", float((err[live] ** 2).sum().sqrt() / (dk_ref.norm(dim=-1)[live] ** 2).sum().sqrt()))
am = ops._amax(qkv)
ptr = lambda t, off: ops._p(t) + off * 4 if False else None
q_, k_, v_ = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
res = ops._attn_fwd(qkv.data_ptr(), qkv.data_ptr() + d * 4, qkv.data_ptr() + 2 * d * 4, 3 * d, 3 * d, 3 * d, B, H, T, T, lens, True, 0.0, 0, False, am, am, am, None)
stat = res[1] if isinstance(res, tuple) else None
print("fwd returned", type(res), [tuple(r.shape) if hasattr(r, "shape") else r for r in res])
for r in res:
    if hasattr(r, "shape") and r.dim() == 4 and r.shape[0] == 3:
        l2 = r[2][live.cuda()]
        print("l2 of live rows: unique values (first 8)", torch.unique(l2)[:8].tolist(), " == 10:", int((l2 == 10.0).sum()), "of", l2.numel())
