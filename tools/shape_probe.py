#!/usr/bin/env python3
"""Per-shape HIP-event timing of every GEMM-type launch of one training step (development aid)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle.spec import model_config
from oracle.synth import synth_batch
from transformertts_amd import ops, _lib
from transformertts_amd.lightning_module import LightningModule
cfg = model_config("base")
config = {"model": dict(cfg, device="cuda"), "loss": {"stop_weight": 8.0},
          "training": {"num_epochs": 300, "teacher_forcing_mode": "linear", "warmup_steps": 4000, "sync_loss_every_step": False}}
dev = torch.device("cuda:0")
lm = LightningModule(config).to(dev); lm.train()
oc = lm.configure_optimizers(); opt, sch = oc["optimizer"], oc["lr_scheduler"]["scheduler"]
batch = {k: v.to(dev) for k, v in synth_batch(64, 100, 870, 80, 100, seed=1).items()}
lib = _lib.load()
def step(i):
    opt.zero_grad(); loss = lm.training_step(batch, i); loss.backward(); opt.step(); sch.step()
for i in range(3): step(i)
torch.cuda.synchronize()
DIMS = {
    "ttts_linear_fwd_h3": lambda a: ("lin fwd h3", a[5], a[6], a[7]),
    "ttts_linear_fwd_h3d": lambda a: ("lin fwd h3d", a[5], a[6], a[7]),
    "ttts_linear_fwd_h3i": lambda a: ("lin fwd h3i", a[6], a[7], a[8]),
    "ttts_linear_fwd_h3d_img": lambda a: ("lin fwd img", a[5], a[6], a[7]),
    "ttts_linear_bwd_data_h3": lambda a: ("lin dgrad h3", a[4], a[6], a[5]),
    "ttts_linear_bwd_data_h3d": lambda a: ("lin dgrad h3d", a[4], a[6], a[5]),
    "ttts_linear_bwd_data_h3i": lambda a: ("lin dgrad h3i", a[5], a[7], a[6]),
    "ttts_conv1d_fwd_h3": lambda a: ("conv fwd", a[4] * a[5], a[7], a[6] * a[8]),
    "ttts_conv1d_bwd_data_h3": lambda a: ("conv dgrad", a[3] * a[4], a[5], a[6] * a[7]),
    "ttts_linear_bwd_weight_h3": lambda a: ("lin wgrad", a[6], a[7], a[8]),
    "ttts_conv1d_bwd_weight_h3": lambda a: ("conv wgrad", a[6] * a[7], a[9], a[8] * a[10]),
}
rec = []
orig = {}
for n, f in DIMS.items():
    fn = getattr(lib, n); orig[n] = fn
    def wrapped(*a, _fn=fn, _f=f):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); rc = _fn(*a); e1.record()
        rec.append((_f(a), e0, e1)); return rc
    setattr(lib, n, wrapped)
step(3)
torch.cuda.synchronize()
agg = collections.OrderedDict()
for key, e0, e1 in rec:
    g = agg.setdefault(key, [0, 0.0]); g[0] += 1; g[1] += e0.elapsed_time(e1)
tot = 0
for (kind, M, N, K), (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    fl = 2.0 * M * N * K
    print(f"{kind:14s} M={M:6d} N={N:5d} K={K:5d}  x{n:2d}  {ms/n*1e3:8.1f} us  {fl/(ms/n*1e-3)/1e12:6.1f} TF  total {ms:6.3f} ms")
    tot += ms
print("total", tot)
