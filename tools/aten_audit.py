#!/usr/bin/env python3
"""Which stock ATen kernels / copies still run inside one eager training step (torch profiler, with Python stacks)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from transformertts_amd.lightning_module import LightningModule
from transformertts_amd.step import TrainStep
from transformertts_amd.workload import model_config, synth_batch
cfg = model_config("base")
config = {"model": dict(cfg, device="cuda"), "loss": {"stop_weight": 8.0},
          "training": {"num_epochs": 300, "teacher_forcing_mode": "linear", "warmup_steps": 4000, "sync_loss_every_step": False,
                       "fused_clip_norm": 1.0}}
lm = LightningModule(config).to("cuda"); lm.train()
oc = lm.configure_optimizers()
batch = {k: v.to("cuda") for k, v in synth_batch(8, 100, 300).items()}
ts = TrainStep(lm, oc["optimizer"], oc["lr_scheduler"]["scheduler"], batch, graph=False)
for _ in range(3):
    ts()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    ts()
    torch.cuda.synchronize()
for ev in prof.key_averages(group_by_stack_n=12):
    if ev.key.startswith("aten::") and ev.device_time_total > 0 and ev.key not in ("aten::empty",):
        print(f"{ev.key:28s} n={ev.count:3d} dev_us={ev.device_time_total:8.1f}")
        for fr in ev.stack[:12]:
            print("      ", fr)
