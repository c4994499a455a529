#!/usr/bin/env python3
"""Count weight-split launches per training step (development aid)."""
import os, sys, time
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
cnt = {"single": 0, "batched": 0, "rebuild": 0}
o1, o2 = lib.ttts_weight_split, lib.ttts_weight_split_batched
class W:
    def __init__(s, f, k): s.f, s.k = f, k
    def __call__(s, *a): cnt[s.k] += 1; return s.f(*a)
lib.ttts_weight_split = W(o1, "single"); lib.ttts_weight_split_batched = W(o2, "batched")
def step(i):
    opt.zero_grad(); loss = lm.training_step(batch, i); loss.backward(); opt.step(); sch.step()
last_table = None
import cProfile, pstats, io
for i in range(6):
    for k in cnt: cnt[k] = 0
    pr = cProfile.Profile()
    torch.cuda.synchronize(); t = time.perf_counter()
    pr.enable(); step(i); pr.disable()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    if i > 0 and (t1 - t) > 0.03:
        s_ = io.StringIO(); pstats.Stats(pr, stream=s_).sort_stats("tottime").print_stats(10); print(s_.getvalue()[:3000])
    tab = ops._plane_table
    print(i, dict(cnt), "entries", len(ops._plane_entries), "table changed", tab is not last_table, f"host {1e3*(t1-t):.1f} ms total {1e3*(t2-t):.1f} ms")
    last_table = tab

# where does a slow step spend its host time?
import cProfile, pstats, io
for i in range(6, 14):
    pr = cProfile.Profile(); torch.cuda.synchronize(); t = time.perf_counter(); pr.enable()
    step(i)
    pr.disable(); t1 = time.perf_counter(); torch.cuda.synchronize()
    if (t1 - t) > 0.03:
        s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(8); print("SLOW STEP", i, f"{1e3*(t1-t):.1f} ms"); print(s.getvalue()[:2500])
    else:
        print("step", i, f"host {1e3*(t1-t):.1f} ms")
