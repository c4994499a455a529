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
        # deferred second-stage reductions of the parameter-gradient kernels that write into this bucket: a queue of
        # the C ABI owned by the bucket (device buckets only), run by flush_reductions()
        self.queue = None
        if self.flat.is_cuda:
            from . import ops
            self.queue = ops.ReduceQueue()
        self._tail = None
        # True: run the collectives even in a one-rank group (rehearsing the RCCL path on a single GPU: init, AVG over
        # the bucket behind a graph replay, the optimizer behind the collective)
        self.force_collective = False
        self.attach()

    def attach(self) -> None:
        """Point every .grad at its slice of the flat buffer and register the slice as the parameter's gradient
        sink: the HIP backward kernels then add their results straight into the bucket (ops._sink); anything that
        still flows through autograd accumulates in place into the same memory."""
        for p, v in zip(self.params, self.views):
            p.grad = v
            p._ttts_grad_sink = v
            p._ttts_reduce_queue = self.queue

    def flush_reductions(self) -> None:
        """Run the queued parameter-gradient reductions on the current stream: after this (in stream order) the bucket
        holds every gradient of the backward passes so far.  Called after backward, before any collective over the bucket
        and before the optimizer reads it; cheap when nothing is queued."""
        if self.queue is not None:
            self.queue.flush()

    def zero(self) -> None:
        if self.queue is not None:
            self.queue.clear()       # leftovers of a backward pass that raised: their destinations are being zeroed anyway
        self._tail = None
        if self.flat.is_cuda:        # a memset on the stream, not a fill kernel (one node of the captured step graph)
            from . import _lib, ops
            _lib.check(_lib.load().ttts_zero(ops._p(self.flat), self.flat.numel() * self.flat.element_size(), ops._stream()),
                       "ttts_zero")
        else:                        # host-side buckets exist only in the gloo CPU tests
            self.flat.zero_()
        self.attach()

    def _solo(self, group) -> bool:
        """No exchange to do: no process group, or a group of one (unless `force_collective`)."""
        if not (dist.is_available() and dist.is_initialized()):
            return True
        return dist.get_world_size(group) == 1 and not self.force_collective

    def allreduce_mean(self, group: Optional[dist.ProcessGroup] = None, async_op: bool = False):
        self.flush_reductions()
        if self._solo(group):
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

    # ---- overlap with backward: the tail of the bucket (parameters registered from `first_param` on) can be reduced as
    # soon as backward has passed the module that owns `first_param`, while the gradients of earlier modules are still
    # being computed.  For this model: decoder + postnet + heads (55 % of the bytes) finish before the encoder side starts.
    def offset_of(self, param: torch.nn.Parameter) -> int:
        for p, o in zip(self.params, self.offsets):
            if p is param:
                return o
        raise ValueError("offset_of: parameter is not in the bucket")

    def _reduce_range(self, lo: int, hi: int, group, async_op: bool):
        view = self.flat[lo:hi]
        world = dist.get_world_size(group)
        if dist.get_backend(group) == "gloo":
            if view.is_cuda:                     # rehearsal on a GPU box: staged through the host, synchronous
                host = view.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
                view.copy_(host.div_(world))
                return None
            work = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
            if async_op:
                return (work, view, world)
            view.div_(world)
            return None
        work = dist.all_reduce(view, op=dist.ReduceOp.AVG, group=group, async_op=async_op)
        return (work, None, world) if async_op else None

    def start_tail_allreduce(self, lo: int, group: Optional[dist.ProcessGroup] = None) -> None:
        """Begin the (asynchronous) mean all-reduce of flat[lo:].  Call it from a backward hook once every gradient in
        that range is final; every rank must call it at the same point of its step."""
        self.flush_reductions()
        if self._solo(group):
            return
        if getattr(self, "_tail", None) is not None:
            raise RuntimeError("start_tail_allreduce: a tail reduction is already in flight")
        self._tail = (lo, self._reduce_range(lo, self.flat.numel(), group, True))

    def finish_allreduce(self, group: Optional[dist.ProcessGroup] = None) -> None:
        """Reduce whatever `start_tail_allreduce` has not covered and wait for the tail: after this the whole bucket
        holds the mean gradient.  Without a started tail it is `allreduce_mean()`."""
        self.flush_reductions()
        if self._solo(group):
            return
        tail = getattr(self, "_tail", None)
        self._tail = None
        if tail is None:
            self.allreduce_mean(group)
            return
        lo, pending = tail
        if lo > 0:
            self._reduce_range(0, lo, group, False)
        if pending is not None:
            work, view, world = pending
            work.wait()
            if view is not None:
                view.div_(world)

    def clip_grad_norm_(self, max_norm: float) -> torch.Tensor:
        """Global-norm clipping over the flat buffer (train.py:41 `gradient_clip_val`), after the all-reduce."""
        norm = torch.linalg.vector_norm(self.flat)
        scale = torch.clamp(max_norm / (norm + 1e-6), max=1.0)
        self.flat.mul_(scale)
        return norm


class _TailTrigger:
    """Fires `on_ready(lo)` once per backward pass, when the gradient of EVERY grad-requiring tensor that entered
    `boundary` in the last grad-enabled forward has been computed -- i.e. when backward has left the module and all of
    its parameter gradients (and those of everything after it) have been enqueued.

    The trigger hangs on the input TENSORS (Tensor.register_hook), found by a forward pre-hook that sees positional and
    keyword arguments alike.  A module-level `register_full_backward_hook` is NOT usable here: it only tracks
    positional inputs, and for a module called with keyword arguments PyTorch fires it as soon as the gradient w.r.t.
    the module's OUTPUT exists -- before any of its parameter gradients do."""

    def __init__(self, boundary: torch.nn.Module, lo: int, on_ready, before=None):
        self.lo, self.on_ready, self.before = lo, on_ready, before
        self.pending = 0
        self.generation = 0
        self.fired = 0
        self.enabled = True               # False: forwards do not arm the trigger (nothing fires)
        self._pre = boundary.register_forward_pre_hook(self._arm, with_kwargs=True)

    def _arm(self, module, args, kwargs):
        if not torch.is_grad_enabled() or not self.enabled:
            return None
        tensors = [t for t in list(args) + list(kwargs.values())
                   if isinstance(t, torch.Tensor) and t.requires_grad and t.is_floating_point()]
        self.generation += 1
        gen = self.generation
        self.pending = len(tensors)           # no visible grad-requiring input: never fires, finish_allreduce() does it all

        def arrived(grad, _gen=gen):
            if _gen == self.generation:
                self.pending -= 1
                if self.pending == 0:
                    self.fired += 1
                    if self.before is not None:
                        self.before()          # parameter-gradient reductions queued so far land in the bucket first
                    self.on_ready(self.lo)
            return None
        seen = set()
        for t in tensors:
            if id(t) in seen:                  # the same tensor passed twice gets one hook
                self.pending -= 1
                continue
            seen.add(id(t))
            t.register_hook(arrived)
        return None

    def remove(self) -> None:
        self._pre.remove()
        self.generation += 1


def overlap_tail_with_backward(bucket: FlatGradBucket, model: torch.nn.Module, boundary: torch.nn.Module, group=None,
                               on_ready=None):
    """Arrange for the bucket's tail -- the gradients of `boundary` and of every module registered after it -- to be
    all-reduced as soon as backward has left `boundary`, overlapping the exchange with the rest of backward.  Valid
    only if those later modules sit AFTER `boundary` in the forward pass too, so that their gradients are final by
    then: that is checked structurally here (their parameters must form exactly the bucket's tail).
    Returns the trigger (call `.remove()` to uninstall), or None when the layout does not allow it (the caller then
    just uses finish_allreduce(), which reduces everything at once).  The optimizer side calls
    bucket.finish_allreduce() either way.  `on_ready(lo)` replaces the default action (tests)."""
    first = next(iter(boundary.parameters()), None)
    if first is None:
        return None
    try:
        lo = bucket.offset_of(first)
    except ValueError:
        return None
    after, seen = set(), False
    for _, child in model.named_children():
        seen = seen or child is boundary
        if seen:
            after.update(id(p) for p in child.parameters())
    in_bucket = {id(p) for p in bucket.params}
    tail = {id(p) for p, o in zip(bucket.params, bucket.offsets) if o >= lo}
    if not seen or lo == 0 or tail != {i for i in after if i in in_bucket}:
        return None
    if on_ready is None:
        def on_ready(lo_):
            bucket.start_tail_allreduce(lo_, group)
    return _TailTrigger(boundary, lo, on_ready, before=bucket.flush_reductions)


def broadcast_module_state(module: torch.nn.Module, src: int = 0, group=None, force: bool = False) -> None:
    """Make every replica start from rank `src`'s parameters and buffers (`force`: also in a one-rank group)."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
        return
    stage = dist.get_backend(group) == "gloo"
    for t in list(module.parameters()) + list(module.buffers()):
        if stage and t.is_cuda:                 # gloo rehearsal on a GPU box: broadcast a host copy
            host = t.data.cpu()
            dist.broadcast(host, src=src, group=group)
            t.data.copy_(host)
        else:
            dist.broadcast(t.data, src=src, group=group)
    # raw .data writes do not move autograd's version counters: drop every cached bf16 weight split explicitly
    from . import ops
    ops.bump_param_epoch()


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
