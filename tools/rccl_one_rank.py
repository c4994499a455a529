#!/usr/bin/env python3
"""RCCL on the one GPU a build box has: a ONE-rank process group over the real "nccl" backend (= RCCL on ROCm).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port P tools/rccl_one_rank.py

Exercises what an 8-GPU node runs for the first time otherwise: process-group init with a device id, the broadcast of the
module state, `ReduceOp.AVG` over the flat gradient bucket enqueued right behind a HIP-graph replay on the same stream,
and the optimizer kernels behind the collective.  Checks (printed as one JSON line, exit code 1 on failure):
  * a one-rank mean leaves the bucket bit for bit unchanged;
  * N steps through the data-parallel launch path (graph = zero-grad .. backward, then collective, then clip + Adam)
    end in exactly the parameters / moments / BatchNorm buffers of N steps through the single-process path (graph
    including the optimizer), dropout on;
  * the same with the exchange overlapped: the step captured as TWO graphs cut where backward leaves the decoder, the tail's
    all-reduce started between their replays (step.TrainStep, overlap=True);
  * the same with the second graph's `capture_begin` made to FAIL: TrainStep must say so, fall back to one graph per step
    (whole bucket exchanged after backward) and still end in the same bits (`_capture_guarded`).
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist


def main():
    cfg_name = sys.argv[1] if len(sys.argv) > 1 else "base"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("nccl", device_id=dev)
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1

    from transformertts_amd.lightning_module import LightningModule
    from transformertts_amd.parallel import broadcast_module_state
    from transformertts_amd.step import TrainStep
    from transformertts_amd.workload import model_config, synth_batch

    cfg = model_config(cfg_name)
    config = {"model": dict(cfg, device="cuda"), "loss": {"stop_weight": 8.0},
              "training": {"num_epochs": 300, "teacher_forcing_mode": "linear", "warmup_steps": 50,
                           "sync_loss_every_step": False, "fused_clip_norm": 1.0}}
    B, Tp, Tm = (4, 60, 300) if cfg_name != "tiny" else (3, 12, 40)
    batch = {k: v.to(dev) for k, v in synth_batch(B, Tp, Tm, cfg["n_mels"], cfg["n_phon"], ragged=True, seed=8).items()}
    out = {"backend": dist.get_backend(), "world": dist.get_world_size(), "config": cfg_name, "steps": steps}
    runs = []
    for force, overlap, sabotage in ((False, False, False), (True, False, False), (True, True, False), (True, True, True)):
        torch.manual_seed(42)
        lm = LightningModule(config).to(dev)
        lm.train()
        broadcast_module_state(lm, force=force)            # RCCL broadcast of every parameter and buffer
        oc = lm.configure_optimizers()
        opt, sch = oc["optimizer"], oc["lr_scheduler"]["scheduler"]
        ts = TrainStep(lm, opt, sch, batch, graph=True, seed=77, force_collective=force, overlap=overlap)
        begin = torch.cuda.CUDAGraph.capture_begin
        if sabotage:
            calls = {"n": 0}

            def failing_begin(self, *a, **k):             # the SECOND capture_begin of a step is the tail graph's
                calls["n"] += 1
                if calls["n"] == 2:
                    raise RuntimeError("sabotaged capture_begin (tools/rccl_one_rank.py)")
                return begin(self, *a, **k)
            torch.cuda.CUDAGraph.capture_begin = failing_begin
        try:
            losses = [ts().detach().clone() for _ in range(steps)]
        finally:
            torch.cuda.CUDAGraph.capture_begin = begin
        torch.cuda.synchronize()
        assert ts.graphed
        if sabotage:
            out["fallback"] = ts.capture_fallback
            out["fallback_split_graphs"] = sum(len(sl.tails) for sl in ts._slots.values())
        else:
            # a fallback that fires on the NORMAL path is a finding, not a pass: every un-sabotaged run must have captured what it asked for
            out.setdefault("normal_path_fallbacks", []).append(ts.capture_fallback)
            if overlap:
                out["split_graphs"] = sum(len(sl.tails) for sl in ts._slots.values())
                out["tail_trigger_fired"] = ts.trigger.fired if ts.trigger is not None else 0
        if force and not overlap:
            before = opt.bucket.flat.clone()
            dist.all_reduce(opt.bucket.flat, op=dist.ReduceOp.AVG)
            torch.cuda.synchronize()
            out["one_rank_mean_leaves_bucket_unchanged"] = bool(torch.equal(before, opt.bucket.flat))
        runs.append((losses, opt.flat_params.clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone(),
                     {k: v.clone() for k, v in lm.model.state_dict().items() if "running" in k}))
        # captured graphs go before the process group does (and before the next run re-uses the allocator's pools)
        for sl in ts._slots.values():
            sl.drop_graphs()
        del ts
        torch.cuda.synchronize()
    a = runs[0]
    same = lambda b: (all(torch.equal(x, y) for x, y in zip(a[0], b[0])),
                      bool(torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])
                           and all(torch.equal(a[4][k], b[4][k]) for k in a[4])))
    out["losses_equal"], out["state_equal"] = same(runs[1])
    out["overlap_losses_equal"], out["overlap_state_equal"] = same(runs[2])
    out["fallback_losses_equal"], out["fallback_state_equal"] = same(runs[3])
    out["final_loss"] = float(a[0][-1])
    out["ok"] = bool(out["losses_equal"] and out["state_equal"] and out["one_rank_mean_leaves_bucket_unchanged"]
                     and out["overlap_losses_equal"] and out["overlap_state_equal"] and out["split_graphs"] == 1
                     and out["fallback_losses_equal"] and out["fallback_state_equal"] and out["fallback_split_graphs"] == 0
                     and bool(out["fallback"]) and all(f is None for f in out["normal_path_fallbacks"]))
    print(json.dumps(out), flush=True)
    import gc
    gc.collect()
    torch.cuda.synchronize()
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if out["ok"] else 1)


if __name__ == "__main__":
    main()
