"""Data-parallel gradient exchange: one flat fp32 bucket, one all-reduce (mean) per optimizer step.

The reference has no distributed code at all (`devices=1`, train.py:47).  The path shards over utterances
(SURVEY.md section 8e): every rank runs the full 7.9 M-parameter replica on its own shard of the batch, BatchNorm
statistics stay per-rank (no SyncBN), and the only exchange is the gradient mean -- DDP semantics.  All parameter
gradients live as views into ONE contiguous buffer (31.6 MB for the base config), so the exchange is a single
RCCL all-reduce over xGMI (backend "nccl" on ROCm) with no per-tensor launches and no flatten/unflatten copies;
on CPU test rigs the same code runs over gloo.
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


class FlatGradBucket:
    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("FlatGradBucket: no trainable parameters")
        dev, dt = self.params[0].device, self.params[0].dtype
        sizes = [p.numel() for p in self.params]
        # 64-float alignment of every view keeps each gradient 256-B aligned for the vectorised kernels
        self.offsets, off = [], 0
        for n in sizes:
            self.offsets.append(off)
            off += (n + 63) // 64 * 64
        self.flat = torch.zeros(off, dtype=dt, device=dev)
        self.views = [self.flat[o:o + n].view_as(p) for o, n, p in zip(self.offsets, sizes, self.params)]
        self.attach()

    def attach(self) -> None:
        """Point every .grad at its slice of the flat buffer and register the slice as the parameter's gradient
        sink: the HIP backward kernels then add their results straight into the bucket (ops._sink); anything that
        still flows through autograd accumulates in place into the same memory."""
        for p, v in zip(self.params, self.views):
            p.grad = v
            p._ttts_grad_sink = v

    def zero(self) -> None:
        self.flat.zero_()
        self.attach()

    def allreduce_mean(self, group: Optional[dist.ProcessGroup] = None, async_op: bool = False):
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
            return None
        if dist.get_backend(group) == "gloo":   # gloo has no AVG; device buffers are staged through the host
            if self.flat.is_cuda:
                host = self.flat.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
                self.flat.copy_(host.div_(dist.get_world_size(group)))
                return None
            work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
            if async_op:
                return work
            self.flat.div_(dist.get_world_size(group))
            return None
        return dist.all_reduce(self.flat, op=dist.ReduceOp.AVG, group=group, async_op=async_op)

    def clip_grad_norm_(self, max_norm: float) -> torch.Tensor:
        """Global-norm clipping over the flat buffer (train.py:41 `gradient_clip_val`), after the all-reduce."""
        norm = torch.linalg.vector_norm(self.flat)
        scale = torch.clamp(max_norm / (norm + 1e-6), max=1.0)
        self.flat.mul_(scale)
        return norm


def broadcast_module_state(module: torch.nn.Module, src: int = 0, group=None) -> None:
    """Make every replica start from rank `src`'s parameters and buffers."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    stage = dist.get_backend(group) == "gloo"
    for t in list(module.parameters()) + list(module.buffers()):
        if stage and t.is_cuda:                 # gloo rehearsal on a GPU box: broadcast a host copy
            host = t.data.cpu()
            dist.broadcast(host, src=src, group=group)
            t.data.copy_(host)
        else:
            dist.broadcast(t.data, src=src, group=group)


def shard_batch(batch: dict, rank: int, world: int) -> dict:
    """Contiguous per-rank shard of a collated batch; each rank re-trims to its own maxima (dataset.py:71-103 order
    is preserved: rows stay sorted by phoneme length inside the shard)."""
    B = batch["phoneme"].size(0)
    if B % world != 0:
        raise ValueError(f"global batch {B} is not divisible by world size {world}")
    per = B // world
    sl = slice(rank * per, (rank + 1) * per)
    pl, ml = batch["phoneme_lens"][sl], batch["melspec_lens"][sl]
    tp, tm = int(pl.max()), int(ml.max())
    return {"phoneme": batch["phoneme"][sl, :tp].contiguous(), "melspec": batch["melspec"][sl, :tm].contiguous(),
            "phoneme_lens": pl.contiguous(), "melspec_lens": ml.contiguous()}
