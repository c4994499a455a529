#!/usr/bin/env python3
"""cProfile of the host side of one training step (development aid): where the Python launch path spends its time."""
import cProfile, pstats, os, sys, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle.spec import model_config
from oracle.synth import synth_batch
from transformertts_amd import ops
from transformertts_amd.lightning_module import LightningModule
from transformertts_amd.parallel import FlatGradBucket
cfg = model_config("base")
config = {"model": dict(cfg, device="cuda"), "loss": {"stop_weight": 8.0},
          "training": {"num_epochs": 300, "teacher_forcing_mode": "linear", "warmup_steps": 4000, "sync_loss_every_step": False}}
dev = torch.device("cuda:0")
lm = LightningModule(config).to(dev); lm.train()
bucket = FlatGradBucket(lm.parameters())
oc = lm.configure_optimizers(); opt, sch = oc["optimizer"], oc["lr_scheduler"]["scheduler"]
batch = {k: v.to(dev) for k, v in synth_batch(64, 100, 870, 80, 100, seed=1).items()}
def step(i):
    bucket.zero(); loss = lm.training_step(batch, i); loss.backward(); bucket.clip_grad_norm_(1.0); opt.step(); sch.step()
for i in range(3): step(i)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for i in range(3): step(3 + i)
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28); print(s.getvalue()[:6000])
