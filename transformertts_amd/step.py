"""TrainStep: one optimizer step of the reference's training loop as the MI355X runs it -- as captured HIP graphs, one per
batch shape.

What a step is (reference lightning_module.py:45-86 + what its Trainer does around it, train.py:38-51):

    zero gradients -> [training_step (no-grad forward, scheduled-sampling mix, forward, loss) -> backward] x accumulate
    -> [N > 1: gradient all-reduce (mean) over RCCL] -> global-norm clip -> Adam (lr = Noam factor) -> scheduler.step()

Eagerly that is ~700 kernel launches issued one by one from Python (16 ms of host time per step, measured in round 1:
more than the kernels need once they get faster, and already more than a batch-16 step needs).  The launch sequence is
identical from step to step for a fixed batch shape, so it is captured into a HIP graph and replayed; the host then
spends one small host-to-device copy plus one graph launch per micro-batch.

Real batches vary in shape (dataset.py:71-103 pads to the batch maxima), so the graphs live in a cache keyed on the
batch shape `(B, Tp, Tm)` and the micro-batch's role in the accumulation window; all of them share ONE memory pool (they
never run concurrently), so the cache costs the memory of its largest member, not the sum.  `lattice=(p, m)` rounds the
padded lengths up to multiples of p phonemes / m frames before the lookup, which bounds the number of distinct shapes
(LJSpeech under a length-bucketed sampler: a few dozen): the extra rows are padding exactly like the reference's own
(zero input, masked in attention and in the loss by the true lengths); like the reference's padding rows they are
visible to BatchNorm's batch statistics and to the convolutions' receptive fields at an utterance's end, so the option
is off by default and the parity tests run on exact shapes.

Kernel arguments are frozen at capture, so everything that varies per step is read from device memory by the kernels
(`ops.StepState`, `ttts_step_state` in include/ttts_hip.h): the dropout / sampling seed word, the learning rate, the
Adam step count and the teacher-forcing ratio.  Dropout SITE seeds are numbered from zero at the start of every
micro-batch in both modes, so an eager step and a replayed step with the same state block are bit-identical
(tests/test_hip_graph.py).  The weight planes of the split-precision GEMMs are refreshed by an explicit batched launch at
the start of the first micro-batch after an optimizer step (`ops.PlaneTable`, owned by this object and recorded in the
graph), never through the lazy process-wide cache.

Data parallel (N > 1): the graph ends after backward; the all-reduce of the flat bucket and the optimizer kernels are
issued eagerly behind it on the same stream.  With `overlap=True` the exchange is split where backward leaves the decoder
(parallel.overlap_tail_with_backward: decoder, post-net and head gradients, 55 % of the bucket, are final there): eagerly a
host hook inside backward starts the tail's all-reduce; in graph mode the step is captured as TWO graphs cut at that hook
(the capture runs autograd on the calling thread so that the hook can end one capture and begin the next), and a replay is
graph A -> start the tail's all-reduce (RCCL's own stream, behind A) -> graph B (the encoder's backward, beside the exchange)
-> the head's all-reduce -> clip + Adam.  The collectives themselves are never captured.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch

from . import ops
from .optim import FlatAdam
from .parallel import overlap_tail_with_backward

# Stream-capture mode of every graph this module records.  The default ("global") makes a HIP call that is illegal during capture
# an error in ANY thread of the process -- including ProcessGroupNCCL's watchdog thread, which polls the events of earlier
# collectives with hipEventQuery: one capture in ten of a data-parallel run died with "operation not permitted when stream is
# capturing" raised in the watchdog (SIGABRT; found by looping tools/rccl_one_rank.py).  "thread_local" confines the check to
# the capturing thread; launches that other threads (autograd's) put on the capturing stream are recorded either way.
_CAPTURE_MODE = "thread_local"
_MIX = 0x9E3779B97F4A7C15
_KEYS = ("phoneme", "melspec", "phoneme_lens", "melspec_lens")


def _step_seed(base: int, step: int) -> int:
    """splitmix64 of (base, step): the seed word of step `step`."""
    z = (base + step * _MIX) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return z ^ (z >> 31)


def _round_up(n: int, m: int) -> int:
    return (n + m - 1) // m * m


class _Slot:
    """Static buffers of one batch shape and the graphs captured over them (one per accumulation role)."""
    __slots__ = ("batch", "graphs", "tails", "losses", "eager_runs", "used")

    def __init__(self, batch):
        self.batch = batch
        self.graphs: Dict[str, torch.cuda.CUDAGraph] = {}
        self.tails: Dict[str, torch.cuda.CUDAGraph] = {}      # second half of a step cut at the data-parallel tail trigger
        self.losses: Dict[str, torch.Tensor] = {}
        self.eager_runs = 0
        self.used = 0                        # TrainStep.index at the last use (LRU eviction)

    def drop_graphs(self) -> None:
        self.graphs.clear()
        self.tails.clear()
        self.losses.clear()


class TrainStep:
    def __init__(self, lm, optimizer: FlatAdam, scheduler, batch: Optional[Dict[str, torch.Tensor]] = None, *,
                 graph: bool = True, seed: int = 0, group=None, overlap: bool = False, eager_warmup: int = 2,
                 accumulate: int = 1, lattice: Optional[Tuple[int, int]] = None, max_shapes: int = 64,
                 force_collective: bool = False):
        """`lm`: the LightningModule counterpart (training_step surface).  `batch` (optional): device tensors with the keys
        of the reference's collate_fn; its shape gets the first set of static buffers.  Later batches of any shape are
        copied into the static buffers of THEIR shape (`__call__(batch)` / `load`), created on first sight.
        `graph=True` captures a shape at its first use after `eager_warmup` eager micro-batches overall.
        `accumulate=k`: gradients of k micro-batches are summed (each scaled by 1/k, as Lightning's
        accumulate_grad_batches does, train.py:42) before the optimizer steps.  `lattice=(p, m)`: see the module text.
        `overlap=True` (N > 1 only) starts the all-reduce of the decoder / postnet / head gradients while backward is still
        in the encoder -- from a host hook inside backward in eager mode, between the two graphs of a step cut at that hook in
        graph mode.  `force_collective=True` takes the data-parallel path (graph ends after backward,
        collective + optimizer behind it) in a one-rank process group too: RCCL on a single GPU."""
        if not isinstance(optimizer, FlatAdam):
            raise TypeError("TrainStep drives FlatAdam (flat parameter / gradient / moment buffers)")
        self.lm, self.opt, self.sched, self.group = lm, optimizer, scheduler, group
        self.bucket = optimizer.bucket
        dev = self.bucket.flat.device
        self.device = dev
        self.state = ops.StepState(dev)
        self.accumulate = max(1, int(accumulate))
        self._grad_seed = torch.full((), 1.0 / self.accumulate, dtype=torch.float32, device=dev)   # resident d(loss)
        self.base_seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.index = 0                       # micro-batches taken
        self.micro = 0                       # position inside the accumulation window
        self.use_graph = bool(graph)
        self.eager_warmup = max(2, int(eager_warmup))     # micro-batch 1 builds the weight-plane caches, 2 their table
        self.lattice = tuple(int(v) for v in lattice) if lattice else None
        self.max_shapes = int(max_shapes)
        self._slots: Dict[Tuple[int, int, int], _Slot] = {}
        self.evictions = 0
        self.recaptures = 0                  # times a moved / edited parameter invalidated the captured graphs
        self.split_capture = True            # cut the captured step at the tail trigger (two graphs) when there is a trigger
        self.capture_fallback = None         # what _capture_guarded had to give up, if anything
        self._cur: Optional[_Slot] = None
        self._pool = None                    # memory pool shared by every captured graph
        self._planes: Optional[ops.PlaneTable] = None
        import torch.distributed as dist
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        if force_collective and not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("TrainStep(force_collective=True) needs an initialised process group")
        self.bucket.force_collective = bool(force_collective)
        self.dp = self.world > 1 or bool(force_collective)       # the optimizer runs behind a collective, outside the graph
        self.trigger = None
        self._cut = None                     # during a split capture: callable that ends graph A and begins graph B
        if overlap and self.dp:
            self.trigger = overlap_tail_with_backward(self.bucket, lm.model, lm.model.decoder, group, on_ready=self._tail_ready)
        if batch is not None:
            self.load(batch)

    # ------------------------------------------------------------------------------------------------ batches
    @property
    def batch(self) -> Dict[str, torch.Tensor]:
        """Static buffers of the current shape."""
        if self._cur is None:
            raise RuntimeError("TrainStep: no batch loaded yet")
        return self._cur.batch

    def _key(self, batch) -> Tuple[int, int, int]:
        B, Tp = batch["phoneme"].shape
        Tm = batch["melspec"].shape[1]
        if self.lattice:
            Tp, Tm = _round_up(Tp, self.lattice[0]), _round_up(Tm, self.lattice[1])
        return int(B), int(Tp), int(Tm)

    def load(self, batch: Dict[str, torch.Tensor]) -> None:
        """Copy the next batch into the static buffers of its (lattice-rounded) shape, creating them on first sight."""
        key = self._key(batch)
        slot = self._slots.get(key)
        if slot is None:
            if len(self._slots) >= self.max_shapes:
                # least-recently-used shape leaves: its graphs are destroyed, its static buffers freed.  (Graphs hold kernel
                # nodes only -- `ttts_zero` is a fill kernel -- so a pool-mate's survival does not depend on this one's memory,
                # DESIGN 9.2.)  Never evict inside an accumulation window: the window's graphs belong to the current shape.
                victim = min((k for k, v in self._slots.items() if v is not self._cur), key=lambda k: self._slots[k].used,
                             default=None)
                if victim is None:
                    raise RuntimeError("TrainStep: max_shapes must be at least 2")
                torch.cuda.synchronize()     # the victim's last replay may still be running
                self._slots.pop(victim).drop_graphs()
                self.evictions += 1
                self._pool_check()
            B, Tp, Tm = key
            src = {k: batch[k] for k in _KEYS}
            if any(v.device != self.device for v in src.values()):
                raise ValueError("TrainStep: the batch must live on the model's HIP device")
            n_mels = src["melspec"].shape[2]
            static = {"phoneme": torch.zeros(B, Tp, dtype=src["phoneme"].dtype, device=self.device),
                      "melspec": torch.zeros(B, Tm, n_mels, dtype=src["melspec"].dtype, device=self.device),
                      "phoneme_lens": torch.zeros(B, dtype=src["phoneme_lens"].dtype, device=self.device),
                      "melspec_lens": torch.zeros(B, dtype=src["melspec_lens"].dtype, device=self.device)}
            slot = self._slots[key] = _Slot(static)
        dst = slot.batch
        for k in _KEYS:
            s, d = batch[k], dst[k]
            if s.shape == d.shape:
                d.copy_(s, non_blocking=True)
            else:                                  # lattice padding: id 0 / 0.0 beyond the batch's own maxima
                d.zero_()
                d[tuple(slice(0, n) for n in s.shape)].copy_(s, non_blocking=True)
        slot.used = self.index
        self._cur = slot

    # ------------------------------------------------------------------------------------------------ pieces
    def _push_state(self) -> None:
        self.state.push(seed=_step_seed(self.base_seed, self.index), lr=float(self.opt.param_groups[0]["lr"]),
                        p_tf=float(self.lm.teacher_forcing_ratio()), step=self.opt._step + 1)

    def _role(self) -> str:
        first, last = self.micro == 0, self.micro == self.accumulate - 1
        return "full" if (first and last) else "first" if first else "last" if last else "mid"

    def _forward_backward(self, role: Optional[str] = None, capturing: bool = False) -> torch.Tensor:
        """One micro-batch: [zero-grad + weight-plane refresh] + training_step + backward + flush of the deferred
        parameter-gradient reductions."""
        dev = self.device
        role = role or self._role()
        trig = self.trigger
        hold = trig is not None and trig.enabled and role not in ("full", "last")
        if hold:
            trig.enabled = False             # the exchange belongs to the window's last micro-batch only
        ops.seeds.counter = 0                # site seeds are numbered per micro-batch; the seed word makes them fresh
        ops.amax_arena_reset(dev)            # one memset for all partial-maxima arrays of this micro-batch
        try:
            if role in ("full", "first"):
                self.opt.zero_grad()         # also drops reductions a failed backward pass left queued
                if self._planes is not None and (capturing or self._planes.stale()):
                    self._planes.refresh()   # the optimizer stepped: every weight is re-split by one batched launch
            loss = self.lm.training_step(self._cur.batch, self.index)
            loss.backward(gradient=self._grad_seed)    # a resident 1/accumulate instead of autograd's ones_like fill kernel
            self.bucket.flush_reductions()
        finally:
            ops.amax_arena_release(dev)
            if hold:
                trig.enabled = True
        return loss

    def _tail_ready(self, lo: int) -> None:
        """The tail trigger fired: backward has left the decoder, bucket.flat[lo:] is final (its queued reductions have been
        flushed by the trigger).  Eager step: start the tail's all-reduce beside the rest of backward.  Capture: cut the graph
        here -- the collective is issued between the two replays, never recorded."""
        if self._cut is not None:
            self._cut()
        elif torch.cuda.is_current_stream_capturing():
            return          # one-graph capture (the fallback of _capture_guarded): the whole bucket is exchanged after the replay
        else:
            self.bucket.start_tail_allreduce(lo, self.group)

    def _reduce_and_update(self) -> None:
        self.bucket.finish_allreduce(self.group)     # waits for an overlapped tail and reduces the rest; no-op at N = 1
        self.opt.step()

    def _pool_check(self) -> None:
        """The shared capture pool lives as long as one graph captured into it does: with the last graph gone the handle
        is stale (the allocator asserts on it), so the next capture starts a new pool."""
        if not any(s.graphs for s in self._slots.values()):
            self._pool = None

    def _ensure_planes(self) -> None:
        if self._planes is None or not self._planes.valid():
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("TrainStep: weight planes moved during a capture")
            if self._planes is not None:
                # A parameter moved or was edited behind the table (load_state_dict, broadcast_module_state, .to()): every
                # graph captured so far recorded `ttts_weight_split_batched` over the OLD table's device descriptors, which
                # die with the old table.  Drop those graphs (they are re-captured at their next use) BEFORE the table goes.
                torch.cuda.synchronize()
                for slot in self._slots.values():
                    slot.drop_graphs()
                self.recaptures += 1
                self._pool_check()
                self._planes = ops.PlaneTable(self.lm.model)
                self._planes.refresh()       # whatever edited the parameters: the planes follow now, wherever the window stands
                return
            self._planes = ops.PlaneTable(self.lm.model)

    # ------------------------------------------------------------------------------------------------ one step
    def __call__(self, batch: Optional[Dict[str, torch.Tensor]] = None) -> torch.Tensor:
        """One micro-batch; the optimizer (and the scheduler) step on every `accumulate`-th call."""
        if batch is not None:
            self.load(batch)
        slot = self._cur
        if slot is None:
            raise RuntimeError("TrainStep: no batch loaded yet")
        slot.used = self.index
        ops.seeds.ensure_seeded()
        role = self._role()
        last = role in ("full", "last")
        self._push_state()
        if self.index >= 1:
            self._ensure_planes()
        eager = (not self.use_graph) or self.index < self.eager_warmup
        if eager:
            with self.state:
                loss = self._forward_backward(role)
                if last:
                    self._reduce_and_update()
            slot.eager_runs += 1
        else:
            if role not in slot.graphs:
                self._capture_guarded(slot, role)
            if not self.use_graph:               # the capture could not be made (see _capture_guarded): this step runs eagerly
                return self.__call__()
            slot.graphs[role].replay()
            if role in slot.tails:           # the step was cut at the tail trigger: exchange the tail beside graph B
                self.bucket.start_tail_allreduce(self.trigger.lo, self.group)
                slot.tails[role].replay()
            if last:
                if not self.dp:
                    self.opt.note_external_step()
                else:
                    with self.state:
                        self._reduce_and_update()
            loss = slot.losses[role]
        if last:
            self.sched.step()
        self.index += 1
        self.micro = 0 if last else self.micro + 1
        return loss

    def _capture_guarded(self, slot: _Slot, role: str) -> None:
        """`_capture`, with a way out for the one form that depends on the process group's behaviour under stream capture: if
        the TWO-graph capture (cut at the tail trigger, `split_capture`) raises, say so on stderr, capture the step as one
        graph (the whole bucket is then exchanged after backward, as before round 4), and if that fails too run without
        graphs.  `capture_fallback` records what happened (bench.py puts it in `config.launch_path`).  A failure of the
        ONE-graph capture with no split attempted is a defect and propagates."""
        attempted_split = self.trigger is not None and self.split_capture and role in ("full", "last")
        try:
            self._capture(slot, role)
            return
        except Exception as e:  # noqa: BLE001
            if not attempted_split:
                raise
            import sys
            print(f"[TrainStep] two-graph capture failed ({type(e).__name__}: {e}); falling back to one graph per step",
                  file=sys.stderr, flush=True)
            self.capture_fallback = f"two-graph capture failed ({type(e).__name__}); one graph per step"
        self.split_capture = False
        slot.drop_graphs()
        import gc
        gc.collect()                             # the half-made graph objects release their share of the pool
        try:
            torch.cuda.synchronize()
            self._pool_check()                   # (the first graph of the pool died with the attempt: start a new pool)
            self._capture(slot, role)
        except Exception as e:  # noqa: BLE001
            import sys
            print(f"[TrainStep] one-graph capture failed as well ({type(e).__name__}: {e}); running without graphs",
                  file=sys.stderr, flush=True)
            self.capture_fallback = f"graph capture failed ({type(e).__name__}); eager launches"
            for sl in self._slots.values():
                sl.drop_graphs()
            self._pool_check()
            self.use_graph = False

    def _capture(self, slot: _Slot, role: str) -> None:
        """Record one micro-batch of `role` over the static buffers of `slot` into a HIP graph.  Nothing executes during
        capture; the caller replays it right away."""
        self._ensure_planes()
        torch.cuda.synchronize()
        if role not in ("full", "first"):
            self._planes.mark_current()      # the window's first graph refreshes the planes at replay: record no re-splits here
        g = torch.cuda.CUDAGraph()
        step0 = self.opt._step
        prev = self._cur
        self._cur = slot
        if self._pool is None:
            self._pool = torch.cuda.graph_pool_handle()
        split = self.trigger is not None and self.split_capture and role in ("full", "last")
        g_tail = None
        try:
            if not split:
                with torch.cuda.graph(g, pool=self._pool, capture_error_mode=_CAPTURE_MODE):
                    with self.state:
                        slot.losses[role] = self._forward_backward(role, capturing=True)
                        if role in ("full", "last") and not self.dp:
                            self.opt.step()
            else:
                # Two graphs cut where the tail trigger fires.  The hook runs inside backward, so autograd must run on THIS thread
                # (a stream capture is ended by the thread that began it); everything is recorded on one side stream.
                tail = torch.cuda.CUDAGraph()
                cut = {"done": False}

                def cut_here():
                    g.capture_end()
                    tail.capture_begin(pool=self._pool, capture_error_mode=_CAPTURE_MODE)
                    cut["done"] = True
                cap = torch.cuda.Stream()
                cap.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(cap), torch.autograd.set_multithreading_enabled(False):
                    g.capture_begin(pool=self._pool, capture_error_mode=_CAPTURE_MODE)
                    self._cut = cut_here
                    completed = False
                    try:
                        with self.state:
                            slot.losses[role] = self._forward_backward(role, capturing=True)
                        completed = True
                    finally:
                        self._cut = None
                        try:
                            (tail if cut["done"] else g).capture_end()
                        except Exception:  # noqa: BLE001
                            if completed:    # (after a failure inside the pass its own exception is the one to report)
                                raise
                torch.cuda.current_stream().wait_stream(cap)
                if cut["done"]:
                    g_tail = tail
        finally:
            self._cur = prev
            self.opt._step = step0           # the capture pass ran the host bookkeeping of a step that did not execute
            self._planes.mark_stale()        # ... and the recorded plane refresh has not run either
        slot.graphs[role] = g
        if g_tail is not None:
            slot.tails[role] = g_tail

    def ensure_captured(self) -> None:
        """Capture the current shape's graph for the role of the NEXT micro-batch now (outside any timed region) instead
        of lazily inside the first step past the eager warm-up.  Needs two eager micro-batches before it: the first creates
        the weight-plane caches, the second their descriptor table."""
        if not self.use_graph:
            return
        slot, role = self._cur, self._role()
        if slot is None or role in slot.graphs:
            return
        if self.index < 2:
            raise RuntimeError("TrainStep.ensure_captured: run at least two (eager) steps first")
        self._push_state()                   # harmless: the next step pushes its own state again
        self._capture_guarded(slot, role)
        self.eager_warmup = min(self.eager_warmup, self.index)

    @property
    def graphed(self) -> bool:
        return self._cur is not None and bool(self._cur.graphs)

    @property
    def n_graphs(self) -> int:
        return sum(len(s.graphs) for s in self._slots.values())
