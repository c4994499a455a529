#!/usr/bin/env python3
"""bench.py -- mel frames/sec of one teacher-forced Transformer-TTS training step on N MI355X.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python bench.py --gpus N --steps K --warmup W          (N > 1 with no launcher: starts the line below as a child)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = the arithmetic of the reference's `LightningModule.training_step` (lightning_module.py:45-86):
no-grad forward (train mode) -> scheduled-sampling mix -> forward -> loss -> backward, followed by what the
reference's Trainer does per optimizer step (train.py:38-51): gradient all-reduce (N > 1), global-norm clip 1.0,
Adam + Noam LR.  Dropout is ON (p = 0.5 / 0.1 as config.yaml), BatchNorm in train mode, fp32 throughout.
Workload at N = 1: BASELINE.json configs[2] -- batch 64, d_model 256, 3+3 layers, 4 heads, dense synthetic
LJSpeech-shaped batch (100 phonemes, 870 frames x 80 mels per utterance, seed 1234); weak scaling (64 per GPU).
Inputs are resident in HBM before the timed region.  Prints ONE JSON line on rank 0.

The step is driven by `transformertts_amd.step.TrainStep`: after two eager steps it is captured into ONE HIP graph
(zero-grad + both forwards + loss + backward + clip + Adam; at N > 1 the graph ends after backward and the RCCL
all-reduce + optimizer kernels follow it on the stream) and replayed.  TTTS_GRAPH=0 keeps the eager launch path
(with TTTS_DP_OVERLAP=1 the tail of the gradient bucket is then exchanged while backward is still in the encoder).
Other configs: --config scaled --batch 32 (BASELINE configs[4] per-GPU shard), --batch 16 (configs[1] shape, fp32).
--ragged --cycle N: a stream of N differently shaped LJSpeech-like batches, round-robin (one graph per shape out of the
shape-keyed cache; --lattice P,M rounds the padded lengths to multiples of P phonemes / M frames first).
After the timed steps the same step is replayed for --sustain seconds more and the median step time of that stretch is
reported as `sustained` (the timed region of a default run is a third of a second: too short to show the clocks the chip
holds under sustained matrix load).
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



def self_launch(argv) -> int | None:
    """`python bench.py --gpus N` with N > 1 and no launcher environment: start
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same arguments>` as a CHILD process
    (never exec: see the environment notes on replacing a process; nothing here has touched the GPU -- torch is not even
    imported yet), relay its stdout / stderr and return its exit code.  Returns None when this process is itself a rank
    (RANK / WORLD_SIZE set by a launcher) or N == 1.  The reference has no counterpart (train.py:38-51 is devices=1)."""
    import socket
    import subprocess
    n = 1
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            n = int(argv[i + 1])
        elif a.startswith("--gpus="):
            n = int(a.split("=", 1)[1])
    if n <= 1 or "RANK" in os.environ or "WORLD_SIZE" in os.environ:
        return None
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = str(sock.getsockname()[1])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "4"))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    print(f"[bench] --gpus {n} without a launcher: starting {' '.join(cmd[1:9])} ...", file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


if __name__ == "__main__":
    _rc = self_launch(sys.argv[1:])
    if _rc is not None:
        raise SystemExit(_rc)

import torch
import torch.distributed as dist

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: bf16 MFMA, dense


TRAFFIC_FILES = ("r06_traffic.json", "r05_traffic.json", "r04_traffic.json")     # see tools/collect_traffic.py; round 3's file mis-normalised WRITE_SIZE (x1.9) and is not read


def measured_traffic(kernel: str, workload: str = "base_b64"):
    """(HBM bytes per launch of `kernel`, source file) from the committed PMC collection -- separate rocprofv3
    `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of this same command, FETCH_SIZE doubled per the gfx950 note of
    MI355X_MICROARCH.md -- or (None, None).  PMC passes cannot run inside the timed bench, so the figure is a stamped
    reading of the build named in the file, not of this run.  A workload other than the default one has its own file
    (`r03_traffic_<workload>.json`) or no figure: the default workload's bytes per launch are not its bytes."""
    files = TRAFFIC_FILES if workload == "base_b64" else (f"r04_traffic_{workload}.json",)
    for name in files:
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                t = json.load(f)
        except Exception:  # noqa: BLE001
            continue
        ent = t.get(kernel)
        if ent:
            return ent["hbm_bytes_per_launch"], f"profiles/{name}" + (f" ({t['_meta']})" if "_meta" in t else "")
    return None, None


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

    TILES = {1: "64,64,2,2", 2: "128,128,2,2", 3: "64,128,2,2", 4: "128,96,4,1", 6: "256,256,2,4", 7: "256,128,4,2",
             8: "256,128,2,2", 9: "wide 256,256,2,2"}
    # name -> (x6?, extractor of (M, N, K) from the C-ABI argument tuple)
    CALLS = {
        "ttts_linear_fwd": (0, lambda a: (a[5], a[6], a[7])),
        "ttts_linear_fwd_x6": (1, lambda a: (a[5], a[6], a[7])),
        "ttts_linear_fwd_h3": (2, lambda a: (a[5], a[6], a[7])),
        "ttts_conv1d_fwd_h3": (2, lambda a: (a[4] * a[5], a[7], a[6] * a[8])),
        "ttts_linear_bwd_data_h3": (2, lambda a: (a[4], a[6], a[5])),
        "ttts_conv1d_bwd_data_h3": (2, lambda a: (a[3] * a[4], a[5], a[6] * a[7])),
        "ttts_linear_bwd_data": (0, lambda a: (a[4], a[6], a[5])),        # (dy, w, res, dx, M, N, K): out N_gemm = K, red = N
        "ttts_linear_bwd_data_x6": (1, lambda a: (a[4], a[6], a[5])),
        "ttts_conv1d_fwd": (0, lambda a: (a[4] * a[5], a[7], a[6] * a[8])),
        "ttts_conv1d_fwd_x6": (1, lambda a: (a[4] * a[5], a[7], a[6] * a[8])),
        "ttts_conv1d_bwd_data": (0, lambda a: (a[3] * a[4], a[5], a[6] * a[7])),
        "ttts_conv1d_bwd_data_x6": (1, lambda a: (a[3] * a[4], a[5], a[6] * a[7])),
        # LDS-DMA kernel (gemm_h3i.hip): image operand / raw fp32 operand
        "ttts_linear_fwd_h3i": (2, lambda a: (a[6], a[7], a[8])),
        "ttts_linear_bwd_data_h3i": (2, lambda a: (a[5], a[7], a[6])),
        "ttts_linear_fwd_h3d": (2, lambda a: (a[5], a[6], a[7])),
        "ttts_linear_bwd_data_h3d": (2, lambda a: (a[4], a[6], a[5])),
        # ... with a head-image output (attention in-projections): (x, planes, bias, image, row_inv, M, N, K, ...)
        "ttts_linear_fwd_h3d_img": (2, lambda a: (a[5], a[6], a[7])),
    }
    # gemm_h3i_kernel<A_RAW, HAS_RES, HAS_GATE, DROP> as dispatch_h3i instantiates it: name -> argument positions of
    # (residual, relu gate or None, dropout probability or None)
    DMA = {"ttts_linear_fwd_h3i": (False, 4, None, 10), "ttts_linear_bwd_data_h3i": (False, 3, 8, None),
           "ttts_linear_fwd_h3d": (True, 3, None, 9), "ttts_linear_bwd_data_h3d": (True, 2, 7, None)}

    def __init__(self, lib):
        self.lib, self.records, self.orig, self.shapes = lib, [], {}, []

    def __enter__(self):
        for n, (x6, dims) in self.CALLS.items():
            fn = getattr(self.lib, n)
            self.orig[n] = fn

            def wrapped(*a, _fn=fn, _x6=x6, _dims=dims, _name=n):
                # unshifted operands only: T > 0 (argument 13 of the linear forward: the go-frame shift) and convolutions go
                # through the clipping loader of the 8-wave kernel
                _plain = _name == "ttts_linear_bwd_data_h3" or (_name == "ttts_linear_fwd_h3" and a[13] <= 0)
                M, N, K = _dims(a)
                tile = self.lib.ttts_gemm_tile_choice(M, N, K, _x6)
                rows = 256
                if _x6 == 2 and tile == 6 and K >= 96:
                    # csrc/gemm_h3.hip h3_wide_rows: 224-row tiles where rounds x height is smaller; shifted / clipped operands
                    # (convolutions, the go-frame shift) take the one-wave-per-SIMD kernel on its 224-row variant only
                    nx = -(-N // 256)
                    r256, r224 = -(-(nx * -(-M // 256)) // 256) * 256, -(-(nx * -(-M // 224)) // 256) * 224
                    rows = 224 if r224 < r256 else 256
                    if _plain or rows == 224:
                        tile = 9    # gemm_h3_wide_kernel (one wave per SIMD)
                if _name == "ttts_linear_fwd_h3d_img":
                    tile = "gemm_h3i_kernel<128,true,false,false,false,true>"
                if _name in self.DMA:
                    raw, i_res, i_gate, i_p = self.DMA[_name]
                    flag = lambda v: "true" if v else "false"      # noqa: E731
                    gated = i_gate is not None and bool(a[i_gate])
                    # rows per tile as csrc/gemm_h3i.hip h3i_big_tile picks them
                    big = N >= 1024 and K <= 256 and not gated and (-(-M // 256)) * (-(-N // 256)) >= 256
                    tile = "gemm_h3i_kernel<%d,%s,%s,%s,%s,false>" % (256 if big else 128, flag(raw), flag(a[i_res]), flag(gated),
                                                                      flag(i_p is not None and a[i_p] > 0))
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                rc = _fn(*a)
                e1.record()
                # compulsory HBM bytes of the launch: the activation read once, the output written once, the weights once
                # (residual / gate operands of some epilogues not counted)
                # bytes that cross the CUs' global-memory paths on a 256 x 256 tiling (any form of the 256-wide kernels): every
                # tile loads its 256 x K activation block and its 256 x K weight block (re-reads come from L2, but through the same
                # per-CU path) and stores 256 x 256 outputs
                cu_bytes = (-(-M // rows)) * (-(-N // 256)) * ((rows + 256.0) * K * 4) + 4.0 * M * N
                if isinstance(tile, str) and not tile.startswith("gemm_h3i_kernel<256"):       # 128 x 256 tiles
                    cu_bytes = (-(-M // 128)) * (-(-N // 256)) * ((128 + 256) * K * 4.0) + 4.0 * M * N
                self.records.append((e0, e1, 2.0 * M * N * K, (_x6, tile), 4.0 * (M * K + M * N + N * K), cu_bytes))
                self.shapes.append((_name, int(M), int(N), int(K)))
                return rc
            setattr(self.lib, n, wrapped)
        return self

    def __exit__(self, *exc):
        for n, fn in self.orig.items():
            setattr(self.lib, n, fn)

    def shape_table(self):
        """per (entry point, M, N, K): launches, mean us, TF -- development aid (`--gemm-shapes`)"""
        torch.cuda.synchronize()
        t = {}
        for (e0, e1, fl, key, _, _), sh in zip(self.records, self.shapes):
            g = t.setdefault((sh, key[1] if isinstance(key[1], str) else self.TILES[key[1]]), [0, 0.0, fl])
            g[0] += 1
            g[1] += e0.elapsed_time(e1)
        rows = sorted(t.items(), key=lambda kv: -kv[1][1])
        return [f"{v[1] * 1e3:8.1f} us total  {v[0]:3d} x {v[1] / v[0] * 1e3:7.1f} us  {v[2] / (v[1] / v[0] * 1e-3) / 1e12:6.1f} TF  "
                f"{k[0][0]:28s} M={k[0][1]:6d} N={k[0][2]:5d} K={k[0][3]:5d}  [{k[1]}]" for k, v in rows]

    def summary(self):
        torch.cuda.synchronize()
        groups = {}
        for e0, e1, fl, key, nbytes, cu_bytes in self.records:
            g = groups.setdefault(key, [0, 0.0, 0.0, 0.0, 0.0])
            g[0] += 1
            g[1] += e0.elapsed_time(e1)
            g[2] += fl
            g[3] += nbytes
            g[4] += cu_bytes
        if not groups:
            return None
        key = max(groups, key=lambda k: groups[k][1])
        n, ms, fl, nbytes, cu_bytes = groups[key]
        x6, tile = key

        def label(k):
            return k[1] if isinstance(k[1], str) else f"{('f32', 'bf16x6', 'h3')[k[0]]}<{self.TILES[k[1]]}>"

        name = tile if isinstance(tile, str) else (
            f"gemm_f32_kernel<{self.TILES[tile]},true,*>", f"gemm_bf16x6_kernel<{self.TILES[tile]}>",
            "gemm_h3_wide_kernel" if tile == 9 else f"gemm_h3_kernel<{self.TILES[tile]}>")[x6]
        others = {label(k): {"launches": v[0], "total_ms": v[1], "tflops": v[2] / (v[1] * 1e-3) / 1e12}
                  for k, v in groups.items()}
        return {"kernel": name, "form": x6, "launches": n, "avg_ms": ms / n, "avg_flops": fl / n,
                "tflops": fl / (ms * 1e-3) / 1e12, "total_ms": ms, "all": others, "avg_bytes": nbytes / n,
                "gbps": nbytes / (ms * 1e-3) / 1e9, "cu_gbps": cu_bytes / (ms * 1e-3) / 1e9, "avg_cu_bytes": cu_bytes / n,
                "tile256": tile in (6, 9) or isinstance(tile, str)}


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


def _cpu_point(cfg, B: int, Tp: int, Tm: int, steps: int):
    from oracle import fill_state, synth_batch, oracle_training_step          # checker code, used here as the timed CPU port
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
    return {"batch": B, "frames": Tm, "phonemes": Tp, "s_per_step": med, "frames_per_s": B * Tm / med, "timed_steps": len(times)}


def cpu_baseline(cfg, Tp: int, cfg_name: str) -> dict:
    """The oracle (CPU restatement, kind 'port') timed on this host's cores on the two points SURVEY.md section 8d makes
    mandatory: batch 1 x 600 frames (BASELINE configs[0]) and batch 16 x 870 frames (configs[1] shape); `value` is the
    batch-16 point (the larger sample of the same workload as the GPU line).  About 20-30 s of CPU work in total."""
    torch.set_num_threads(usable_cores())
    p1 = _cpu_point(cfg, 1, Tp, 600, 5)
    p16 = _cpu_point(cfg, 16, Tp, 870, 2 if cfg_name == "base" else 1)
    return {"value": p16["frames_per_s"], "unit": "mel frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle training step (2 fwd + bwd, fp32, dropout on), {cfg_name} config, dense: batch 16 x 870 frames x {Tp} "
                      f"phonemes, median of {p16['timed_steps']} after 1 warm-up, {p16['s_per_step']:.2f} s/step [value]; batch 1 x 600 "
                      f"frames: {p1['frames_per_s']:.0f} frames/s, {p1['s_per_step']:.3f} s/step",
            "points": [p1, p16]}


def rehearsal_gradient_check(ts, rank: int, world: int) -> dict:
    """Two gloo ranks on one GPU (TTTS_BENCH_REHEARSAL=1): the bucket after the exchange -- overlapped tail or single
    collective, whichever this run uses -- must equal the MEAN of the ranks' single-process gradients (DDP semantics,
    SURVEY.md section 8e).  Pass 1 computes this rank's local gradient with the exchange disabled, pass 2 repeats the
    same step (same dropout seeds: site seeds restart per step, same step-state word) with the exchange on."""
    from transformertts_amd import ops
    state, bucket = ts.state, ts.bucket
    fired0 = ts.trigger.fired if ts.trigger is not None else 0
    state.push(seed=0x5EED + rank, lr=0.0, p_tf=1.0, step=1)
    with state:
        if ts.trigger is not None:
            ts.trigger.enabled = False
        ts._forward_backward()
        local = bucket.flat.detach().cpu().double()
        gathered = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(gathered, local)
        mean = sum(gathered) / world
        if ts.trigger is not None:
            ts.trigger.enabled = True
        ts._forward_backward()
        bucket.finish_allreduce(ts.group)
    torch.cuda.synchronize()
    got = bucket.flat.detach().cpu().double()
    rel = float((got - mean).norm() / mean.norm())
    differ = float((gathered[0] - gathered[-1]).norm() / mean.norm())          # the ranks really had different gradients
    return {"rel_l2_vs_mean_of_rank_gradients": rel, "rank_gradients_differ_rel": differ,
            "overlap_requested": ts.trigger is not None,
            "tail_trigger_fired": bool(ts.trigger is not None and ts.trigger.fired == fired0 + 1)}


def launch_check(args, world: int, rank: int):
    """--launch-check: what the driver's N-rank command exercises around the measured step, with the step left out --
    rendezvous on 127.0.0.1, one barrier on each side of a (trivial) timed region, MAX of the ranks' clocks, SUM of the
    ranks' unit counts, ONE JSON line from rank 0.  Runs on a box without a GPU (gloo); tests/test_host_logic.py starts it
    through `self_launch`.  The line says `launch_check: true` and carries no metric value."""
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))
    if world > 1:
        dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    units = torch.tensor([float(args.batch * args.tm)], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(units, op=dist.ReduceOp.SUM)
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "frames_per_step": float(units.item()), "max_rank_seconds": float(t.item()), "value": None}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def workload_name(args) -> str:
    if args.config == "base" and args.batch == 64:
        return "BASELINE configs[2] (batch 64, 1 GPU; per-GPU shard of configs[3])"
    if args.config == "base" and args.batch == 16:
        return "BASELINE configs[1] shape (batch 16) in fp32"
    if args.config == "scaled":
        return f"BASELINE configs[4] per-GPU shard (scaled model, batch {args.batch})"
    return f"{args.config} config, batch {args.batch}"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="utterances per GPU")
    ap.add_argument("--tp", type=int, default=100)
    ap.add_argument("--tm", type=int, default=870)
    ap.add_argument("--config", default="base")
    ap.add_argument("--ragged", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-probe", action="store_true")
    ap.add_argument("--gemm-shapes", action="store_true", help="print the instrumented step's per-shape GEMM table to stderr")
    ap.add_argument("--cycle", type=int, default=1, help="number of distinct ragged batch shapes fed round-robin")
    ap.add_argument("--lattice", default="", help="P,M: pad batches to multiples of P phonemes / M frames (graph-cache key)")
    ap.add_argument("--accumulate", type=int, default=1, help="micro-batches per optimizer step (train.py:42 uses 4)")
    ap.add_argument("--alignments", dest="alignments", action="store_true", default=True,
                    help="(default) the grad forward of every timed step writes the per-head cross-attention maps (B,H,Tm,Tp) as the "
                         "reference's forward does (model/layers.py:68-73); the step's results do not depend on them")
    ap.add_argument("--no-alignments", dest="alignments", action="store_false",
                    help="the timed step skips the maps training_step() discards (lightning_module.py:78-79): the secondary figure as the headline")
    ap.add_argument("--no-alignments-figure", action="store_true", help="skip the secondary figure taken with the other --alignments setting")
    ap.add_argument("--no-image-operands", action="store_true",
                    help="A/B aid: keep every GEMM on the kernels that split the fp32 activation in their loader (gemm_h3): no launch "
                         "of the LDS-DMA kernel (gemm_h3i)")
    ap.add_argument("--no-dma-big-fwd", action="store_true",
                    help="development A/B: 256 -> 1024 forward GEMMs on the gemm_h3 tile instead of the 256-row LDS-DMA tile")
    ap.add_argument("--no-head-images", action="store_true",
                    help="A/B aid: attention in-projections write fp32 and attention runs on the kernels that split q / k / v while they "
                         "stage them (csrc/attention.hip) instead of head images + LDS-DMA (csrc/attention_img.hip)")
    ap.add_argument("--no-fused-kv", action="store_true",
                    help="A/B aid: every decoder layer projects the encoder memory to its cross-attention K/V itself (the reference's "
                         "structure, model/layers.py:54-74) instead of ONE stacked projection for all layers (ops.cross_kv_projection)")
    ap.add_argument("--no-wgrad-groups", action="store_true",
                    help="A/B aid: every small weight gradient is a launch of its own instead of a member of a grouped launch "
                         "(ops.ReduceQueue.defer_wgrad, ttts_linear_bwd_weight_h3_group)")
    ap.add_argument("--wgrad-side-rows", type=int, default=0,
                    help="A/B aid: grouped weight gradients over at most this many rows run on a side stream (0: none unless --wgrad-side-stream)")
    ap.add_argument("--wgrad-side-stream", action="store_true",
                    help="development A/B: the grouped weight-gradient launches run on a side stream beside the data-gradient chain")
    ap.add_argument("--no-twin-encoder", action="store_true",
                    help="A/B aid: each forward of the step encodes its phonemes itself (the reference's structure) instead of ONE "
                         "encoder pass over a batch of 2 B for both (model.encode_twin)")
    ap.add_argument("--no-twin-postnet", action="store_true",
                    help="A/B switch: the no-grad forward runs its own post-net pass (as the reference does) instead of sharing the "
                         "grad forward's as a twin batch (ops.PostnetTwin)")
    ap.add_argument("--layernorm-images", action="store_true",
                    help="A/B aid: LayerNorm forward / backward also write the image operand of their output and the GEMMs behind "
                         "them take it (measured slower over the step: transformertts_amd/ops.py, LAYERNORM_IMAGES)")
    ap.add_argument("--launch-check", action="store_true",
                    help="exercise the launcher protocol only (process group, barrier, MAX-over-ranks clock, rank-0 JSON line) "
                         "over gloo on the host: no model, no GPU, not a measurement")
    ap.add_argument("--sustain", type=float, default=float(os.environ.get("TTTS_BENCH_SUSTAIN", "10")), help="seconds of further replays after the timed steps (0: skip)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if args.launch_check:
        return launch_check(args, world, rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the measured path)")
    # Rehearsal mode for a one-GPU box (TTTS_BENCH_REHEARSAL=1): every rank uses cuda:0 and the collectives run over
    # gloo on host copies, so the launcher protocol (env parsing, barriers, MAX-over-ranks timing, rank-0 JSON) can be
    # exercised without N GPUs.  The real path is one rank per GPU over RCCL ("nccl" on ROCm).
    rehearsal = os.environ.get("TTTS_BENCH_REHEARSAL", "0") == "1"
    dev_index = 0 if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1 or os.environ.get("TTTS_FORCE_DIST", "0") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from transformertts_amd import _lib, ops
    from transformertts_amd.lightning_module import LightningModule
    from transformertts_amd.parallel import broadcast_module_state
    from transformertts_amd.step import TrainStep
    from transformertts_amd.workload import model_config, synth_batch

    if args.no_image_operands:
        ops.DMA_GEMMS = False
    if args.layernorm_images:
        ops.LAYERNORM_IMAGES = True
    if args.no_head_images:
        ops.HEAD_IMAGES = False
    if args.no_dma_big_fwd:
        ops.DMA_BIG_FWD = False
    if args.no_fused_kv:
        ops.FUSED_CROSS_KV = False
    if args.no_wgrad_groups:
        ops.WGRAD_GROUPS = False
    if args.wgrad_side_stream or args.wgrad_side_rows > 0:
        ops.WGRAD_SIDE_STREAM = True
        ops.WGRAD_SIDE_MAX_ROWS = args.wgrad_side_rows
    if args.no_twin_encoder:
        ops.TWIN_ENCODER = False
    if args.no_twin_postnet:
        ops.TWIN_POSTNET = False
    cfg = model_config(args.config)
    config = {"model": dict(cfg, device="cuda"), "loss": {"stop_weight": 8.0},
              "training": {"num_epochs": 300, "teacher_forcing_mode": "linear", "warmup_steps": 4000,
                           "sync_loss_every_step": False, "fused_clip_norm": 1.0,
                           "train_step_alignments": bool(args.alignments)}}
    torch.manual_seed(42)
    lm = LightningModule(config).to(dev)
    lm.train()
    broadcast_module_state(lm)
    opt_cfg = lm.configure_optimizers()          # FlatAdam: flat parameters / gradients / moments, clip folded in
    optimizer, scheduler = opt_cfg["optimizer"], opt_cfg["lr_scheduler"]["scheduler"]

    # per-rank shard of the global synthetic batch (weak scaling: args.batch utterances per GPU).  --cycle N: N batches whose
    # maxima differ (as the reference's collate_fn pads each batch to its own longest utterance, dataset.py:71-103)
    batches = []
    for i in range(max(1, args.cycle)):
        tp_i, tm_i = max(8, args.tp - (3 * i) % 29), max(95, args.tm - (37 * i) % 311)
        b = synth_batch(args.batch, tp_i, tm_i, cfg["n_mels"], cfg["n_phon"], ragged=args.ragged, seed=1234 + rank + 101 * i)
        batches.append({k: v.to(dev) for k, v in b.items()})
    batch = batches[0]
    frames_each = [int(b["melspec_lens"].sum().item()) for b in batches]
    flops_each = [4.0 * sum(algorithmic_flops_forward(cfg, int(p), int(m))
                            for p, m in zip(b["phoneme_lens"].tolist(), b["melspec_lens"].tolist())) for b in batches]
    # work of the timed region: step i takes batch (warmup + i) mod cycle
    frames_rank = sum(frames_each[(args.warmup + i) % len(batches)] for i in range(args.steps)) / args.steps
    flops_rank = sum(flops_each[(args.warmup + i) % len(batches)] for i in range(args.steps)) / args.steps

    use_graph = os.environ.get("TTTS_GRAPH", "1") == "1"
    want_overlap = os.environ.get("TTTS_DP_OVERLAP", "1") == "1"
    # one step = zero-grad, training_step (2 forwards + loss), backward, [all-reduce], clip + Adam, scheduler.step();
    # different ranks draw different dropout masks (seed 42 + rank)
    lattice = tuple(int(v) for v in args.lattice.split(",")) if args.lattice else None
    force_dist = world == 1 and dist.is_initialized()      # TTTS_FORCE_DIST=1: RCCL in a one-rank group
    ts = TrainStep(lm, optimizer, scheduler, batch, graph=use_graph, seed=42 + rank, overlap=want_overlap,
                   eager_warmup=2, accumulate=args.accumulate, lattice=lattice, force_collective=force_dist)
    cycling = len(batches) > 1

    def step(i):
        return ts(batches[i % len(batches)]) if cycling else ts()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    t_start = time.perf_counter()

    def note(msg):
        if rank == 0:
            print(f"[bench +{time.perf_counter() - t_start:7.2f}s] {msg}", file=sys.stderr, flush=True)

    note(f"model and batch on device ({frames_rank} frames/rank)")
    # The warm-up ends the way the timed steps run: eager steps (two at least: the capture needs the weight-plane caches and
    # their descriptor table), the capture, and the LAST warm-up steps (two from --warmup 4 on, one at 3) as the graph's first
    # replays: the one-time upload of the first (~3 ms) is initialisation like the rest of the warm-up, and the card is busy
    # right up to the timed region (the collector's ~80 ms pause runs before them, not between them and the clock).  Round 6's
    # headline used to sit 0.16 ms per step above its own sustained median for exactly these two reasons
    # (`sustained.ratio_to_headline` 0.988; same box: 13.40 -> 13.30 ms).
    single = use_graph and not (cycling or args.accumulate > 1)
    n_replay = 2 if args.warmup >= 4 else 1
    n_eager = max(2, args.warmup - n_replay) if (single and args.warmup >= 3) else args.warmup
    for i in range(n_eager):
        step(i)
        torch.cuda.synchronize()
        note(f"warm-up step {i} done")
    if use_graph:
        while ts.index < 2:
            step(ts.index)                       # --warmup < 2: the capture still needs two eager steps before it
        if not single:                           # every shape / accumulation role is captured at its first use: do a full
            for i in range(len(batches) * args.accumulate):          # round of them before the timed region
                step(args.warmup + i)
            torch.cuda.synchronize()
        else:
            ts.ensure_captured()                 # capture outside the timed region (nothing executes during capture)
        note(f"step captured into HIP graphs ({ts.n_graphs})")
    # A full (generation-2) pass of Python's cyclic collector costs ~80 ms with torch's object graph loaded and tends to
    # fire a few steps into a run: collect now and freeze the survivors so that it cannot land inside the timed steps.
    gc.collect()
    gc.freeze()
    for i in range(n_eager, args.warmup):
        step(i)
        torch.cuda.synchronize()
        note(f"warm-up step {i} done ({'graph replay' if use_graph else 'eager'})")
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
        # (taken right behind the timed steps, BEFORE the sustained stretch: the kernel's clock then is the timed region's, not that
        # of a chip ten seconds into a soak -- round 5's probe read 108 us per launch where the step's own launches took 98)
        graph_was = ts.use_graph
        ts.use_graph = False                     # the probe times individual launches: this step runs eagerly
        if rank == 0:
            with GemmProbe(_lib.load()) as gp:
                step(args.warmup + args.steps)
            probe = gp.summary()
            note("instrumented step done")
            if args.gemm_shapes:
                print("\n".join(gp.shape_table()), file=sys.stderr, flush=True)
        else:
            step(args.warmup + args.steps)
    fence()
    if not args.no_probe:
        ts.use_graph = graph_was                 # the sustained stretch replays the graphs again
    # sustained figure: keep stepping for args.sustain seconds, in stretches of `args.steps` steps fenced like the timed ones
    sustained = None
    if args.sustain > 0 and not rehearsal:
        stretch, t_s0, k = [], time.perf_counter(), args.warmup + args.steps
        while True:
            fence()
            ta = time.perf_counter()
            for i in range(args.steps):
                step(k + i)
            fence()
            tb = time.perf_counter()
            k += args.steps
            stretch.append((tb - ta) / args.steps * 1e3)
            go = torch.tensor([1.0 if tb - t_s0 < args.sustain else 0.0], dtype=torch.float64, device=red_dev)
            if world > 1:
                dist.all_reduce(go, op=dist.ReduceOp.MIN)       # every rank leaves the loop in the same iteration
            if go.item() == 0.0:
                break
        med = statistics.median(stretch)
        sustained = {"seconds": time.perf_counter() - t_s0, "steps": len(stretch) * args.steps,
                     "median_ms_per_step": med, "min_ms_per_step": min(stretch), "max_ms_per_step": max(stretch),
                     "ratio_to_headline": med / (elapsed / args.steps * 1e3),
                     "note": "rank-0 clock around fenced stretches of --steps steps each, after the timed region"}
        note(f"sustained: median {med:.2f} ms/step over {sustained['steps']} further steps")

    # Secondary figure: the same step with the other setting of the attention maps.  The reference's forward always
    # materialises them (model/layers.py:68-73) and its training_step then drops them (lightning_module.py:78-79): the headline
    # step WRITES them, as the reference does (`config.alignments_written`); the secondary figure is the step that skips the
    # maps nobody reads.
    with_alignments = None
    if (not args.no_alignments_figure and world == 1 and not cycling and args.accumulate == 1
            and use_graph and not rehearsal):
        lm.config["training"]["train_step_alignments"] = not args.alignments
        ts2 = TrainStep(lm, optimizer, scheduler, batch, graph=True, seed=42 + rank, eager_warmup=2)
        for _ in range(2):
            ts2()
        ts2.ensure_captured()
        ts2()
        fence()
        ta = time.perf_counter()
        for _ in range(args.steps):
            ts2()
        fence()
        ms2 = (time.perf_counter() - ta) / args.steps * 1e3
        lm.config["training"]["train_step_alignments"] = bool(args.alignments)
        maps_mb = cfg["decoder_n_layers"] * args.batch * cfg["decoder_n_head"] * args.tm * args.tp * 4 / 1e6
        with_alignments = {"ms_per_step": ms2, "value": frames_all / (ms2 * 1e-3), "steps": args.steps,
                           "alignments_written": not args.alignments,
                           "alignment_bytes_per_step": maps_mb * 1e6,
                           "note": ("same step, the grad forward SKIPS the per-head cross-attention maps training_step() discards "
                                    if args.alignments else "same step, the grad forward writes the per-head cross-attention maps ")
                                   + f"({maps_mb:.0f} MB; the no-grad forward needs none either way)"}
        note(f"attention maps {'skipped' if args.alignments else 'written'}: {ms2:.2f} ms/step")
        del ts2

    rehearsal_check = None
    if rehearsal and world > 1:
        rehearsal_check = rehearsal_gradient_check(ts, rank, world)
        note(f"rehearsal gradient check: {rehearsal_check}")
        if not rehearsal_check["rel_l2_vs_mean_of_rank_gradients"] <= 1e-6:
            raise SystemExit(f"rehearsal: reduced gradients differ from the mean of the rank gradients: {rehearsal_check}")


    if rank == 0:
        ms = elapsed / args.steps * 1e3
        out = {
            "metric": "mel frames/sec teacher-forced step, LJSpeech-shape batch, 1/2/4/8 MI355X",
            "value": frames_all * args.steps / elapsed, "unit": "mel frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{workload_name(args)}: training step (no-grad fwd + fwd + loss + bwd + clip + Adam), "
                                   f"batch {args.batch}/GPU x {args.tm} frames x {args.tp} phonemes "
                                   f"({'ragged' if args.ragged else 'dense'}), {args.config} config d_model {cfg['d_model']}, "
                                   f"{cfg['encoder_n_layers']}+{cfg['decoder_n_layers']} layers, dropout on, fp32",
                       "global_batch": args.batch * world, "frames_per_step": frames_all, "parallelism": f"dp{world}",
                       "launch_path": ((f"one HIP graph per step ({ts.n_graphs} captured: one per batch shape and accumulation role)"
                                        if ts.graphed else "eager: one ctypes launch per kernel")
                                       + (f" [{ts.capture_fallback}]" if ts.capture_fallback else "")),
                       "batch_stream": (f"{len(batches)} distinct ragged shapes round-robin" + (f", lattice {lattice}" if lattice else "")
                                        if cycling else "one resident batch"),
                       "accumulate": args.accumulate,
                       "process_group": (dist.get_backend() + f" x{world}") if dist.is_initialized() else "none",
                       "grad_allreduce": ("none (1 GPU)" if not ts.dp else
                                          "tail overlapped with backward" if ts.trigger is not None and (
                                              not ts.graphed or any(sl.tails for sl in ts._slots.values()))
                                          else "one collective after backward"),
                       "alignments_written": bool(args.alignments),
                       "arithmetic": "3 x f16 MFMA terms per fp32 product (hi/lo f16 splits of both operands), fp32 accumulate",
                       "dma_gemms": not args.no_image_operands, "layernorm_images": bool(args.layernorm_images),
                       "head_images": not args.no_head_images, "fused_cross_kv": not args.no_fused_kv, "wgrad_groups": not args.no_wgrad_groups, "twin_encoder": not args.no_twin_encoder, "twin_postnet": not args.no_twin_postnet,
                       "final_loss": final_loss, "per_step_loss_item_sync": False},
            "host_enqueue_ms_per_step": host_elapsed / args.steps * 1e3,
            "sustained": sustained,
            ("without_alignments" if args.alignments else "with_alignments"): with_alignments,
            "step_algorithmic_tflops": flops_all / 1e12,
            "step_achieved_tflops_per_gpu": flops_all / world / (elapsed / args.steps) / 1e12,
        }
        if rehearsal_check is not None:
            out["config"]["rehearsal_gradient_check"] = rehearsal_check
        if probe is not None:
            # The dominant kernel forms every fp32 product from THREE f16 x f16 MFMA terms (hi / lo splits of both operands,
            # csrc/gemm_h3.hip), so the unit that bounds it is the f16 matrix pipe: its ceiling in algorithmic (fp32-equivalent)
            # FLOP/s is the dense f16 peak / 3.  `frac` is priced against THAT ceiling (executed f16 flops over the dense f16
            # peak: the same number); the fp32-MFMA peak (157.3 TF, what a plain v_mfma_f32 kernel is bound by, and the peak of
            # the dtype) is reported beside it as a floor the kernel beats.
            traffic, traffic_src = measured_traffic(probe["kernel"], f"{args.config}_b{args.batch}")
            terms = (0, 6, 3)[probe["form"]]
            if terms:
                peak = PEAK_BF16_MFMA_TFLOPS / terms
                peak_note = (f"dense {'bf16' if terms == 6 else 'f16'} MFMA peak (2.5 PFLOP/s) / {terms} MFMA terms per fp32 product")
            else:
                peak, peak_note = PEAK_F32_MFMA_TFLOPS, "fp32-input MFMA peak"
            out["roofline"] = {"bound": "mfma", "kernel": probe["kernel"],
                               "achieved": probe["tflops"], "peak": peak, "unit": "TFLOP/s",
                               "frac": probe["tflops"] / peak, "peak_is": peak_note,
                               "traffic": traffic, "traffic_source": traffic_src,
                               "vs_fp32_mfma_peak": probe["tflops"] / PEAK_F32_MFMA_TFLOPS,
                               "launches_per_step": probe["launches"], "avg_launch_ms": probe["avg_ms"],
                               "avg_launch_gflop": probe["avg_flops"] / 1e9, "step_share_ms": probe["total_ms"]}
            if terms:
                out["roofline"]["executed_16bit_mfma_tflops"] = terms * probe["tflops"]
            # the same launches against the other ceiling: at K = 256 (most of them) the kernel moves 285 MB for 29 GFLOP, and
            # its time follows the bytes its tiles request from L2 / HBM rather than the MFMA count
            out["roofline"]["hbm_view"] = {"achieved": probe["gbps"], "peak": 8000.0, "unit": "GB/s",
                                           "frac": probe["gbps"] / 8000.0,
                                           "algorithmic_bytes_per_launch": probe["avg_bytes"],
                                           "bytes_are": "4 * (M*K + M*N + N*K) per launch: activation read, output write, weights"}
            if probe["tile256"]:
                # ... and against the ceiling that actually binds a 256 x 256 tiling at K = 256: a CU's global-memory path streams
                # ~10 B per cycle (MI355X_MICROARCH.md, cycle constants: `global_load_dwordx4` HBM-bound; 11-13 measured for
                # mixed / LDS-DMA prologues), and a tile moves 2 * 256 * K * 4 B of operands + 256 KB of output through it
                # for 2 * 256 * 256 * K * 3 MFMA flops -- 128 flop/B at K = 256, where the matrix pipe would need 407
                cu_peak = 256 * 10.0 * 2.4
                out["roofline"]["cu_path_view"] = {"achieved": probe["cu_gbps"], "peak": cu_peak, "unit": "GB/s",
                                                   "frac": probe["cu_gbps"] / cu_peak,
                                                   "bytes_per_launch": probe["avg_cu_bytes"],
                                                   "bytes_are": "per output tile (256 x 256; 128 x 256 for gemm_h3i): activation block + weight block loaded, output stored",
                                                   "peak_is": "256 CUs x ~10 B/cycle/CU x 2.4 GHz (global_load_dwordx4, HBM-bound); L2 hits can exceed it"}
            out["roofline"]["gemm_kernels"] = probe["all"]
        if not args.no_cpu_baseline and world == 1:          # (rank 0 at N = 1 only: the other ranks would wait for it)
            out["cpu_baseline"] = cpu_baseline(cfg, args.tp, args.config)
        print(json.dumps(out), flush=True)
    if world > 1 or dist.is_initialized():
        # captured graphs and everything else that holds device work go BEFORE the process group does
        for sl in ts._slots.values():
            sl.drop_graphs()
        del ts
        gc.collect()
        torch.cuda.synchronize()
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
