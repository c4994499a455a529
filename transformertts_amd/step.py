"""TrainStep: one optimizer step of the reference's training loop as the MI355X runs it -- optionally as ONE captured
HIP graph.

What a step is (reference lightning_module.py:45-86 + what its Trainer does around it, train.py:38-51):

    zero gradients -> training_step (no-grad forward, scheduled-sampling mix, forward, loss) -> backward
    -> [N > 1: gradient all-reduce (mean) over RCCL] -> global-norm clip -> Adam (lr = Noam factor) -> scheduler.step()

Eagerly that is ~700 kernel launches issued one by one from Python (16 ms of host time per step, measured in round 1:
more than the kernels need once they get faster, and already more than a batch-16 step needs).  The launch sequence of a
step is identical from step to step for a fixed batch shape, so it is captured once into a HIP graph and replayed; the
host then spends one small host-to-device copy plus one graph launch per step.

Kernel arguments are frozen at capture, so everything that varies per step is read from device memory by the kernels
(`ops.StepState`, `ttts_step_state` in include/ttts_hip.h): the dropout / sampling seed word, the learning rate, the
Adam step count and the teacher-forcing ratio.  Dropout SITE seeds are numbered from zero at the start of every step in
both modes, so an eager step and a replayed step with the same state block are bit-identical
(tests/test_hip_graph.py).

Data parallel (N > 1): the graph ends after backward; the all-reduce of the flat bucket and the optimizer kernels are
issued eagerly behind it on the same stream (one collective + three launches).  Overlapping the tail of the bucket with
the rest of backward (parallel.overlap_tail_with_backward) needs a host hook inside backward and is therefore an
eager-mode option.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import ops
from .optim import FlatAdam
from .parallel import overlap_tail_with_backward

_MIX = 0x9E3779B97F4A7C15


def _step_seed(base: int, step: int) -> int:
    """splitmix64 of (base, step): the seed word of step `step`."""
    z = (base + step * _MIX) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return z ^ (z >> 31)


class TrainStep:
    def __init__(self, lm, optimizer: FlatAdam, scheduler, batch: Dict[str, torch.Tensor], *, graph: bool = True,
                 seed: int = 0, group=None, overlap: bool = False, eager_warmup: int = 2):
        """`lm`: the LightningModule counterpart (training_step surface); `batch`: device tensors with the keys of the
        reference's collate_fn -- kept as STATIC buffers: later batches of the same shape are copied into them
        (`load`).  `graph=True` captures after `eager_warmup` eager steps.  `overlap=True` (eager mode, N > 1 only)
        starts the all-reduce of the decoder / postnet / head gradients while backward is still in the encoder."""
        if not isinstance(optimizer, FlatAdam):
            raise TypeError("TrainStep drives FlatAdam (flat parameter / gradient / moment buffers)")
        self.lm, self.opt, self.sched, self.group = lm, optimizer, scheduler, group
        self.bucket = optimizer.bucket
        self.batch = {k: v for k, v in batch.items() if isinstance(v, torch.Tensor)}
        dev = self.bucket.flat.device
        if any(v.device != dev for v in self.batch.values()):
            raise ValueError("TrainStep: the batch must live on the model's HIP device")
        self.state = ops.StepState(dev)
        self._one = torch.ones((), dtype=torch.float32, device=dev)
        self.base_seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.index = 0                       # steps taken
        self.use_graph = bool(graph)
        self.eager_warmup = max(2, int(eager_warmup))     # step 1 builds the weight-plane caches, step 2 their table
        self._graph: Optional[torch.cuda.CUDAGraph] = None
        self._loss: Optional[torch.Tensor] = None
        self._graph_has_optimizer = False
        import torch.distributed as dist
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        self.trigger = None
        if overlap and self.world > 1 and not self.use_graph:
            self.trigger = overlap_tail_with_backward(self.bucket, lm.model, lm.model.decoder, group)

    # ------------------------------------------------------------------------------------------------ pieces
    def load(self, batch: Dict[str, torch.Tensor]) -> None:
        """Copy the next batch into the static buffers (same shapes; a different shape needs its own TrainStep)."""
        for k, dst in self.batch.items():
            src = batch[k]
            if src.shape != dst.shape:
                raise ValueError(f"TrainStep.load: {k} has shape {tuple(src.shape)}, the step was built for {tuple(dst.shape)}")
            dst.copy_(src, non_blocking=True)

    def _push_state(self) -> None:
        self.state.push(seed=_step_seed(self.base_seed, self.index), lr=float(self.opt.param_groups[0]["lr"]),
                        p_tf=float(self.lm.teacher_forcing_ratio()), step=self.opt._step + 1)

    def _forward_backward(self) -> torch.Tensor:
        ops.seeds.counter = 0                # site seeds are numbered per step; the step's seed word makes them fresh
        ops.abort_deferred()                 # leftovers of a backward pass that raised
        self.opt.zero_grad()
        loss = self.lm.training_step(self.batch, self.index)
        dev = self.bucket.flat.device
        ops.amax_arena_reset(dev)            # one memset for all partial-maxima arrays of this backward pass
        loss.backward(gradient=self._one)    # a resident 1.0 instead of autograd's ones_like fill kernel
        ops.amax_arena_release(dev)
        return loss

    def _reduce_and_update(self) -> None:
        self.bucket.finish_allreduce(self.group)     # waits for an overlapped tail and reduces the rest; no-op at N = 1
        self.opt.step()

    # ------------------------------------------------------------------------------------------------ one step
    def __call__(self, batch: Optional[Dict[str, torch.Tensor]] = None) -> torch.Tensor:
        if batch is not None:
            self.load(batch)
        self._push_state()
        if not self.use_graph or self.index < self.eager_warmup:
            with self.state:
                loss = self._forward_backward()
                self._reduce_and_update()
        else:
            if self._graph is None:
                self._capture()
            self._graph.replay()
            if self._graph_has_optimizer:
                self.opt.note_external_step()
            else:
                with self.state:
                    self._reduce_and_update()
            loss = self._loss
        self.sched.step()
        self.index += 1
        return loss

    def _capture(self) -> None:
        """Record zero-grad + training_step + backward (+ clip + Adam at N = 1) of ONE step into a HIP graph.  Nothing
        executes during capture; the caller replays it right away."""
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        step0 = self.opt._step
        with torch.cuda.graph(g):
            with self.state:
                self._loss = self._forward_backward()
                if self.world == 1:
                    self.opt.step()
                    self._graph_has_optimizer = True
        self.opt._step = step0               # the capture pass ran the host bookkeeping of a step that did not execute
        self._graph = g

    def ensure_captured(self) -> None:
        """Capture now (outside any timed region) instead of lazily inside the first step past the eager warm-up.  Needs
        two eager steps before it: the first creates the weight-plane caches, the second their descriptor table (pinned
        host allocation + upload, which must not land inside a capture)."""
        if self.use_graph and self._graph is None:
            if self.index < 2:
                raise RuntimeError("TrainStep.ensure_captured: run at least two (eager) steps first")
            self._push_state()               # harmless: the next step pushes its own state again
            self._capture()
            self.eager_warmup = min(self.eager_warmup, self.index)

    @property
    def graphed(self) -> bool:
        return self._graph is not None
