#!/usr/bin/env python3
"""bench.py -- mel frames/sec of one teacher-forced Transformer-TTS training step on N MI355X.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = the arithmetic of the reference's `LightningModule.training_step` (lightning_module.py:45-86):
no-grad forward (train mode) -> scheduled-sampling mix -> forward -> loss -> backward, followed by what the
reference's Trainer does per optimizer step (train.py:38-51): gradient all-reduce (N > 1), global-norm clip 1.0,
Adam + Noam LR.  Dropout is ON (p = 0.5 / 0.1 as config.yaml), BatchNorm in train mode, fp32 throughout.
Workload at N = 1: BASELINE.json configs[2] -- batch 64, d_model 256, 3+3 layers, 4 heads, dense synthetic
LJSpeech-shaped batch (100 phonemes, 870 frames x 80 mels per utterance, seed 1234); weak scaling (64 per GPU).
Inputs are resident in HBM before the timed region.  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import gc
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: bf16 MFMA, dense


def measured_traffic(kernel: str):
    """HBM bytes per launch of the dominant kernel from the committed PMC collection (separate rocprofv3 --pmc passes
    of this same command, FETCH_SIZE doubled per the gfx950 note; profiles/r01_traffic.json), or None."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_traffic.json")) as f:
            t = json.load(f)
        ent = t.get(kernel)
        return ent["hbm_bytes_per_launch"] if ent else None
    except Exception:  # noqa: BLE001
        return None


def algorithmic_flops_forward(cfg, p: int, m: int) -> float:
    """useful forward FLOPs of one utterance with p phonemes, m frames (causal half counted)."""
    d, k = cfg["d_model"], cfg["encoder_prenet_kernel_size"]
    n_mel, Le, Ld = cfg["n_mels"], cfg["encoder_n_layers"], cfg["decoder_n_layers"]
    pre, post = cfg["encoder_prenet_n_layers"], cfg["postnet_n_layers"]
    dffe, dffd = cfg["encoder_d_ffn"], cfg["decoder_d_ffn"]
    E = pre * 2 * d * d * k + 2 * d * d + Le * (8 * d * d + 4 * d * dffe)
    D = 2 * (n_mel * d + d * d) + Ld * (12 * d * d + 4 * d * dffd) + 2 * d * (n_mel + 1) \
        + 2 * cfg["postnet_kernel_size"] * (2 * n_mel * d + (post - 2) * d * d)
    return p * E + m * D + Le * 4 * p * p * d + Ld * 4 * p * d * d + Ld * 4 * m * p * d + Ld * 2 * m * (m + 1) * d


class GemmProbe:
    """HIP-event timing of every forward-type GEMM launch (nn.Linear forward / data-gradient, Conv1d forward /
    data-gradient) on torch's current stream -- the stream the kernels are launched on.  Each launch is attributed to
    the kernel instantiation the library dispatches it to (ttts_gemm_tile_choice); `summary()` reports the
    instantiation with the largest total time = the dominant kernel of the step."""

    TILES = {1: "64,64,2,2", 2: "128,128,2,2", 3: "64,128,2,2", 4: "128,96,4,1"}
    # name -> (x6?, extractor of (M, N, K) from the C-ABI argument tuple)
    CALLS = {
        "ttts_linear_fwd": (0, lambda a: (a[5], a[6], a[7])),
        "ttts_linear_fwd_x6": (1, lambda a: (a[5], a[6], a[7])),
        "ttts_linear_bwd_data": (0, lambda a: (a[4], a[6], a[5])),        # (dy, w, res, dx, M, N, K): out N_gemm = K, red = N
        "ttts_linear_bwd_data_x6": (1, lambda a: (a[4], a[6], a[5])),
        "ttts_conv1d_fwd": (0, lambda a: (a[4] * a[5], a[7], a[6] * a[8])),
        "ttts_conv1d_fwd_x6": (1, lambda a: (a[4] * a[5], a[7], a[6] * a[8])),
        "ttts_conv1d_bwd_data": (0, lambda a: (a[3] * a[4], a[5], a[6] * a[7])),
        "ttts_conv1d_bwd_data_x6": (1, lambda a: (a[3] * a[4], a[5], a[6] * a[7])),
    }

    def __init__(self, lib):
        self.lib, self.records, self.orig = lib, [], {}

    def __enter__(self):
        for n, (x6, dims) in self.CALLS.items():
            fn = getattr(self.lib, n)
            self.orig[n] = fn

            def wrapped(*a, _fn=fn, _x6=x6, _dims=dims):
                M, N, K = _dims(a)
                tile = self.lib.ttts_gemm_tile_choice(M, N, _x6)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                rc = _fn(*a)
                e1.record()
                self.records.append((e0, e1, 2.0 * M * N * K, (_x6, tile)))
                return rc
            setattr(self.lib, n, wrapped)
        return self

    def __exit__(self, *exc):
        for n, fn in self.orig.items():
            setattr(self.lib, n, fn)

    def summary(self):
        torch.cuda.synchronize()
        groups = {}
        for e0, e1, fl, key in self.records:
            g = groups.setdefault(key, [0, 0.0, 0.0])
            g[0] += 1
            g[1] += e0.elapsed_time(e1)
            g[2] += fl
        if not groups:
            return None
        key = max(groups, key=lambda k: groups[k][1])
        n, ms, fl = groups[key]
        x6, tile = key
        name = (f"gemm_bf16x6_kernel<{self.TILES[tile]}>" if x6 else f"gemm_f32_kernel<{self.TILES[tile]},true,*>")
        return {"kernel": name, "x6": bool(x6), "launches": n, "avg_ms": ms / n, "avg_flops": fl / n,
                "tflops": fl / (ms * 1e-3) / 1e12, "total_ms": ms}


def usable_cores() -> int:
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota (a GPU box exposes 256
    logical CPUs but grants a 16-CPU share; oversubscribing the quota stalls the run)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:  # noqa: BLE001
        pass
    return max(1, n)


def cpu_baseline(cfg, B: int, Tp: int, Tm: int, steps: int) -> dict:
    """The oracle (CPU restatement, kind 'port') timed on this host's cores on a bounded sample of the same workload."""
    from oracle import fill_state, synth_batch, oracle_training_step
    torch.set_num_threads(usable_cores())
    sd = fill_state(cfg, 42)
    for k, v in sd.items():
        if v.is_floating_point() and "running" not in k and k != "pe.pe":
            v.requires_grad_(True)
    batch = synth_batch(B, Tp, Tm, cfg["n_mels"], cfg["n_phon"], ragged=False, seed=1234)
    times = []
    for i in range(steps + 1):
        for v in sd.values():
            v.grad = None
        t0 = time.perf_counter()
        loss, _, _ = oracle_training_step(sd, cfg, batch, epoch=0, dropout=True)
        loss["total"].backward()
        dt = time.perf_counter() - t0
        if i > 0:
            times.append(dt)
    med = statistics.median(times)
    return {"value": B * Tm / med, "unit": "mel frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle training step (2 fwd + bwd, fp32, dropout on), dense batch {B} x {Tm} frames x {Tp} phonemes, "
                      f"median of {len(times)} steps after 1 warm-up, {med:.2f} s/step"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=64, help="utterances per GPU")
    ap.add_argument("--tp", type=int, default=100)
    ap.add_argument("--tm", type=int, default=870)
    ap.add_argument("--config", default="base")
    ap.add_argument("--ragged", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=8)
    ap.add_argument("--cpu-steps", type=int, default=5)
    ap.add_argument("--no-probe", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the measured path)")
    # Rehearsal mode for a one-GPU box (TTTS_BENCH_REHEARSAL=1): every rank uses cuda:0 and the collectives run over
    # gloo on host copies, so the launcher protocol (env parsing, barriers, MAX-over-ranks timing, rank-0 JSON) can be
    # exercised without N GPUs.  The real path is one rank per GPU over RCCL ("nccl" on ROCm).
    rehearsal = os.environ.get("TTTS_BENCH_REHEARSAL", "0") == "1"
    dev_index = 0 if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from oracle.spec import model_config
    from oracle.synth import synth_batch
    from transformertts_amd import _lib, ops
    from transformertts_amd.lightning_module import LightningModule
    from transformertts_amd.parallel import FlatGradBucket, broadcast_module_state, overlap_tail_with_backward

    cfg = model_config(args.config)
    config = {"model": dict(cfg, device="cuda"), "loss": {"stop_weight": 8.0},
              "training": {"num_epochs": 300, "teacher_forcing_mode": "linear", "warmup_steps": 4000,
                           "sync_loss_every_step": False, "fused_clip_norm": 1.0}}
    torch.manual_seed(42)
    ops.seeds.manual_seed(42 + rank)
    lm = LightningModule(config).to(dev)
    lm.train()
    broadcast_module_state(lm)
    opt_cfg = lm.configure_optimizers()          # FlatAdam: flat parameters / gradients / moments, clip folded in
    optimizer, scheduler = opt_cfg["optimizer"], opt_cfg["lr_scheduler"]["scheduler"]
    bucket = optimizer.bucket
    # N > 1: the decoder / postnet / head gradients (55 % of the bucket) are exchanged while backward is still in the
    # encoder side; TTTS_DP_OVERLAP=0 keeps the single all-reduce after backward
    overlap = None
    if world > 1 and os.environ.get("TTTS_DP_OVERLAP", "1") == "1":
        overlap = overlap_tail_with_backward(bucket, lm.model, lm.model.decoder)

    # per-rank shard of the global synthetic batch (weak scaling: args.batch utterances per GPU)
    batch = synth_batch(args.batch, args.tp, args.tm, cfg["n_mels"], cfg["n_phon"], ragged=args.ragged, seed=1234 + rank)
    batch = {k: v.to(dev) for k, v in batch.items()}
    frames_rank = int(batch["melspec_lens"].sum().item())
    flops_rank = 4.0 * sum(algorithmic_flops_forward(cfg, int(p), int(m))
                           for p, m in zip(batch["phoneme_lens"].tolist(), batch["melspec_lens"].tolist()))

    def step(i):
        optimizer.zero_grad()
        loss = lm.training_step(batch, i)
        loss.backward()
        bucket.finish_allreduce()                # waits for the overlapped tail and reduces the rest (no-op at N = 1)
        optimizer.step()                         # global-norm clip (1.0) + Adam, two kernels over the flat bucket
        scheduler.step()
        return loss

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    t_start = time.perf_counter()

    def note(msg):
        if rank == 0:
            print(f"[bench +{time.perf_counter() - t_start:7.2f}s] {msg}", file=sys.stderr, flush=True)

    note(f"model and batch on device ({frames_rank} frames/rank)")
    for i in range(args.warmup):
        step(i)
        torch.cuda.synchronize()
        note(f"warm-up step {i} done")
    # A full (generation-2) pass of Python's cyclic collector costs ~80 ms with torch's object graph loaded and tends to
    # fire a few steps into a run: collect now and freeze the survivors so that it cannot land inside the timed steps.
    gc.collect()
    gc.freeze()
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(args.warmup + i)
    host_elapsed = time.perf_counter() - t0     # time to ENQUEUE the steps (host side); the GPU may still be running
    fence()
    elapsed = time.perf_counter() - t0
    note(f"{args.steps} timed steps: {elapsed / args.steps * 1e3:.2f} ms/step (host enqueue {host_elapsed / args.steps * 1e3:.2f} ms/step)")
    red_dev = torch.device("cpu") if rehearsal else dev
    t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    tot = torch.tensor([float(frames_rank), flops_rank], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())
    frames_all, flops_all = float(tot[0].item()), float(tot[1].item())
    final_loss = float(loss.item())

    probe = None
    if not args.no_probe:
        # one extra, untimed step on EVERY rank (it contains the gradient all-reduce); rank 0 instruments it
        if rank == 0:
            with GemmProbe(_lib.load()) as gp:
                step(args.warmup + args.steps)
            probe = gp.summary()
            note("instrumented step done")
        else:
            step(args.warmup + args.steps)
    fence()

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        out = {
            "metric": "mel frames/sec teacher-forced step, LJSpeech-shape batch, 1/2/4/8 MI355X",
            "value": frames_all * args.steps / elapsed, "unit": "mel frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[2]: training step (no-grad fwd + fwd + loss + bwd + clip + Adam), "
                                   f"batch {args.batch}/GPU x {args.tm} frames x {args.tp} phonemes "
                                   f"({'ragged' if args.ragged else 'dense'}), {args.config} config d_model {cfg['d_model']}, "
                                   f"{cfg['encoder_n_layers']}+{cfg['decoder_n_layers']} layers, dropout on, fp32",
                       "global_batch": args.batch * world, "frames_per_step": frames_all, "parallelism": f"dp{world}",
                       "grad_allreduce": ("tail overlapped with backward" if overlap is not None else "one collective after backward"),
                       "final_loss": final_loss, "per_step_loss_item_sync": False},
            "host_enqueue_ms_per_step": host_elapsed / args.steps * 1e3,
            "step_algorithmic_tflops": flops_all / 1e12,
            "step_achieved_tflops_per_gpu": flops_all / world / (elapsed / args.steps) / 1e12,
        }
        if probe is not None:
            # fp32 in / fp32 out / fp32-accurate products: priced against the fp32 MFMA peak of the dtype.  The
            # split-precision kernel executes 6 bf16 MFMA flops per algorithmic flop; that rate and its fraction of the
            # 2.5 PFLOP/s bf16 peak are reported next to it.
            out["roofline"] = {"bound": "mfma", "kernel": probe["kernel"],
                               "achieved": probe["tflops"], "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                               "frac": probe["tflops"] / PEAK_F32_MFMA_TFLOPS, "traffic": measured_traffic(probe["kernel"]),
                               "launches_per_step": probe["launches"], "avg_launch_ms": probe["avg_ms"],
                               "avg_launch_gflop": probe["avg_flops"] / 1e9, "step_share_ms": probe["total_ms"]}
            if probe["x6"]:
                out["roofline"]["executed_bf16_tflops"] = 6.0 * probe["tflops"]
                out["roofline"]["frac_of_bf16_peak"] = 6.0 * probe["tflops"] / PEAK_BF16_MFMA_TFLOPS
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, args.cpu_batch, args.tp, args.tm, args.cpu_steps)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
