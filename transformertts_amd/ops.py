"""torch.autograd.Function wrappers over the C ABI (include/ttts_hip.h).

PyTorch is used here only as plumbing: it owns device memory (caching allocator), the current HIP
stream and the autograd graph.  Every arithmetic step on the path is a hand-written gfx950 kernel
called through `_lib`; nothing in this file computes with torch ops on the hot path, and there is
no CPU / eager fallback -- non-CUDA tensors are rejected.
"""
from __future__ import annotations

import os
import weakref
from ctypes import c_void_p
from typing import Optional

import torch

from . import _lib

ACT_NONE, ACT_RELU, ACT_TANH = 0, 1, 2
AMAX_SLOTS = 1024         # TTTS_AMAX_SLOTS of include/ttts_hip.h: floats per partial-maxima array


def _p(t: Optional[torch.Tensor]):
    return None if t is None else c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """hipStream_t of torch's current stream on the current device.  torch.cuda.current_stream() costs ~100 us per
    call here (it re-reads os.environ through is_available()), i.e. 25 ms per training step over ~700 launches; the raw
    accessor is the same one torch's own launchers use."""
    if _raw_stream is not None:
        return c_void_p(_raw_stream(torch.cuda.current_device()))
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(t: torch.Tensor, name: str, dtype=torch.float32) -> torch.Tensor:
    if not t.is_cuda:
        raise ValueError(f"{name}: expected a CUDA/HIP tensor (the HIP path has no CPU fallback), got {t.device}")
    if t.dtype != dtype:
        raise ValueError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


# ---- twin batches.  The two forwards of a training step encode the SAME phonemes (reference
# lightning_module.py:53-59 under no_grad, :77 with grad); the encoder is row-parallel, so both run as ONE batch of 2 B: the
# GRAD forward's utterances first, the no-grad forward's behind them.  Autograd only ever sees the first half -- a tensor `t`
# of B utterances that is the leading part of a (2 B, ...) buffer, `t._ttts_twin` -- and every forward kernel is launched on
# the buffer (same pointer, twice the rows); the backward kernels run on the halves they saved, unchanged, and regenerate the
# same dropout masks because the grad half's element indices start at zero.  BatchNorm statistics stay per forward.
TWIN_ENCODER = True


def _twin(t):
    """the (2 B, ...) buffer whose first half `t` is, or None"""
    f = getattr(t, "_ttts_twin", None) if t is not None else None
    if f is None:
        return None
    if f.data_ptr() != t.data_ptr() or f.shape[0] != 2 * t.shape[0] or f.shape[1:] != t.shape[1:] or not f.is_contiguous():
        raise RuntimeError("twin batch: the tensor no longer is the first half of its buffer")
    return f


def twin_pair(full: torch.Tensor):
    """(grad half, no-grad half) of a (2 B, ...) buffer; the grad half carries the buffer (`_ttts_twin`)"""
    B = full.shape[0] // 2
    a = full[:B]
    a._ttts_twin = full
    return a, full[B:]


def _ws(nbytes: int, device) -> torch.Tensor:
    return torch.empty((max(int(nbytes), 16) + 3) // 4, dtype=torch.float32, device=device)


# `torch.manual_seed(s)` with an unchanged s leaves torch.initial_seed() as it was, so a re-seed is invisible from the value
# alone; the reference's dropout draws restart with it (they come from torch's generator).  The two public spellings of the call
# are therefore counted: a thin wrapper that bumps a counter and calls the original (Lightning's seed_everything goes through
# torch.manual_seed).  Seeding a Generator object directly is not seen -- nor does the reference's module-level code do it.
_torch_seed_calls = [0]


def _count_torch_seeding() -> None:
    if getattr(torch.manual_seed, "_ttts_counted", False):
        return
    orig = torch.manual_seed

    def manual_seed(seed):
        _torch_seed_calls[0] += 1
        return orig(seed)
    manual_seed.__doc__ = orig.__doc__
    manual_seed._ttts_counted = True
    manual_seed._ttts_orig = orig
    torch.manual_seed = manual_seed
    if getattr(torch.random, "manual_seed", None) is orig:
        torch.random.manual_seed = manual_seed


_count_torch_seeding()


class _SeedStream:
    """64-bit dropout seeds: one per dropout site per forward call (the kernels hash seed + element index)."""

    def __init__(self, base: int = 0x5EED5EED):
        self.base = base
        self.counter = 0
        self.explicit = False          # manual_seed() was called: the stream no longer follows torch's seed
        self._derived_from = None      # (torch seed, rank) the base was last derived from

    def manual_seed(self, seed: int) -> None:
        self.base = int(seed) & 0xFFFFFFFFFFFF
        self.counter = 0
        self.explicit = True

    def seed_from_torch(self, rank: int = None) -> None:
        """Base = torch's process seed (so `torch.manual_seed` / Lightning's `seed_everything` decide the masks, as they
        do in the reference) folded with the data-parallel rank, so that replicas draw different masks."""
        if rank is None:
            import torch.distributed as dist
            rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        self.base = (torch.initial_seed() ^ (0x9E3779B9 * (int(rank) + 1))) & 0xFFFFFFFFFFFF
        self.counter = 0
        self._derived_from = (torch.initial_seed(), int(rank), _torch_seed_calls[0])

    def ensure_seeded(self) -> None:
        """Called at the first use of a step (training_step / TrainStep), i.e. when the process group exists -- Lightning
        builds the module BEFORE it initialises torch.distributed, so the rank read in a constructor is 0 on every replica.
        Derives the base from (torch seed, rank) whenever that pair changed OR torch was seeded again (`_torch_seed_calls`: seeding
        the same value twice in one process -- train.py:24-28 run twice -- reproduces the same masks, as the reference's draws
        from torch's generator do); does nothing after an explicit manual_seed().  Building a second module (an evaluation copy,
        an EMA / teacher model, load_from_checkpoint) does NOT restart the stream: only a (re-)seed does."""
        if self.explicit:
            return
        import torch.distributed as dist
        rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        if self._derived_from != (torch.initial_seed(), int(rank), _torch_seed_calls[0]):
            self.seed_from_torch(rank)

    def rearm(self) -> None:
        """The next `ensure_seeded` derives the base again and restarts the counter (explicit use; nothing on the product path
        calls it -- a re-seed is seen through `_torch_seed_calls`).  No effect after `manual_seed()`."""
        self._derived_from = None

    def follow_torch(self) -> None:
        """undo `manual_seed()`: the stream follows torch's process seed again (tests that seed explicitly call this on exit)"""
        self.explicit = False
        self._derived_from = None

    def next(self) -> int:
        self.counter += 1
        return ((self.base * 0x9E3779B97F4A7C15) ^ (self.counter * 0xD1B54A32D192ED03)) & 0xFFFFFFFFFFFFFFFF


seeds = _SeedStream()


class StepState:
    """Device-side per-step scalars (`ttts_step_state` of include/ttts_hip.h: seed word, lr, p_tf, step count).

    Kernel arguments are frozen when a HIP graph is captured, so whatever changes from step to step must be read from
    device memory by the kernels.  While a StepState is ACTIVE (`with state:` or `activate()`), every dropout site
    passes `&state.seed` as its `step_seed`, FlatAdam reads lr / step from it and the scheduled-sampling mix reads p_tf
    from it.  `push()` refreshes the block with ONE small host-to-device copy from a ring of pinned slots (no
    synchronisation unless the host gets `ring` steps ahead of the device)."""

    SIZE = 32

    def __init__(self, device, ring: int = 64):
        self.dev = torch.zeros(self.SIZE // 8, dtype=torch.int64, device=device)
        self._host = torch.zeros(ring, self.SIZE // 8, dtype=torch.int64).pin_memory()
        self._f32 = self._host.view(torch.float32)           # (ring, 8) float view of the same bytes
        self._events = [None] * ring
        self._i = 0
        self.ptr = c_void_p(self.dev.data_ptr())

    def push(self, seed: int, lr: float, p_tf: float, step: int) -> None:
        i = self._i % len(self._events)
        ev = self._events[i]
        if ev is not None:
            ev.synchronize()                                  # slot still in flight only if the host is `ring` steps ahead
        seed &= 0xFFFFFFFFFFFFFFFF
        self._host[i, 0] = seed - (1 << 64) if seed >= (1 << 63) else seed
        self._f32[i, 2] = float(lr)
        self._f32[i, 3] = float(p_tf)
        self._host[i, 2] = int(step)
        self._host[i, 3] = 0
        self.dev.copy_(self._host[i], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._events[i] = ev
        self._i += 1

    def __enter__(self):
        global _active_state
        self._prev = _active_state
        _active_state = self
        return self

    def __exit__(self, *exc):
        global _active_state
        _active_state = self._prev


_active_state: Optional[StepState] = None


def _ss():
    """`step_seed` / `ttts_step_state*` argument of the C ABI: the active StepState block, or NULL."""
    return None if _active_state is None else _active_state.ptr

# Forward / data-gradient GEMMs and convolutions: "x6" = fp32-accurate split-precision products on the bf16 MFMA
# (3-way bf16 split, six MFMA terms, fp32 accumulate; error 1.1e-7 vs 2.9e-7 for the plain fp32 MFMA chain),
# "f32" = the plain v_mfma_f32_32x32x2_f32 kernel.  Weight gradients always use the fp32 kernel.
GEMM_MODE = "x6"
ATTN_MODE = GEMM_MODE       # attention products (backward; forward unless ATTN_FWD_MODE): "x6" or "f32"
ATTN_FWD_MODE = "h3"   # forward: "h3", "x6", "f32"
ATTN_BWD_MODE = "h3"   # backward: "h3", "x6", "f32"
WGRAD_MODE = "h3"   # weight gradients: "h3", "x6", "f32"
# Forward GEMMs (nn.Linear / Conv1d forward) under GEMM_MODE "x6": "h3" = fp16x3 split (three f16 MFMA terms, both operands
# pre-scaled by powers of two taken from their MEASURED maxima, csrc/gemm_h3.hip -- no magnitude window), "x6" = bf16x6.
# Shapes the fp16 kernel cannot take (K or channels not a multiple of 4) go to bf16x6.
FWD_MODE = "h3"


# Data-gradient GEMMs: "h3" = fp16x3 with a dynamic pre-scale of the gradient operand (one amax pass over dy), "x6".
BWD_MODE = "h3"


class cross_check_forms:
    """TEST / TOOL API, not configuration: the product path always runs the fp16x3 forms and reads no environment
    variable.  `with ops.cross_check_forms(gemm="f32")` (or fwd= / bwd= / wgrad= / attn_fwd= / attn_bwd= "x6") routes the
    launches made inside the block to the bf16x6 / fp32-MFMA forms the library keeps as numerical cross-checks
    (tests/test_hip_modes.py holds all forms to the same fp64 references); `defer_reduce=False` runs every column reduction
    as it occurs instead of batching them at the end of backward."""

    _NAMES = {"gemm": "GEMM_MODE", "fwd": "FWD_MODE", "bwd": "BWD_MODE", "wgrad": "WGRAD_MODE", "attn": "ATTN_MODE",
              "attn_fwd": "ATTN_FWD_MODE", "attn_bwd": "ATTN_BWD_MODE", "defer_reduce": "DEFER_REDUCE",
              "wgrad_groups": "WGRAD_GROUPS"}

    def __init__(self, **forms):
        unknown = set(forms) - set(self._NAMES)
        if unknown:
            raise TypeError(f"cross_check_forms: unknown form(s) {sorted(unknown)}")
        if forms.get("gemm") == "f32":         # the fp32-MFMA form of everything, as the old process-wide switch meant it
            forms = dict({"attn": "f32", "attn_fwd": "f32", "attn_bwd": "f32", "wgrad": "f32"}, **forms)
        self.forms, self.saved = forms, {}

    def __enter__(self):
        g = globals()
        for k, v in self.forms.items():
            self.saved[k] = g[self._NAMES[k]]
            g[self._NAMES[k]] = v
        return self

    def __exit__(self, *exc):
        g = globals()
        for k, v in self.saved.items():
            g[self._NAMES[k]] = v


def _fwd_h3(K: int, N: int, channels: int = 0) -> bool:
    """fp16x3 takes reduction depths, conv channel counts and output widths that are multiples of 4 (the weight image pads
    the channels of a tap to a multiple of 32 with zeros: ttts_split_image_bytes)."""
    return GEMM_MODE == "x6" and FWD_MODE == "h3" and K % 4 == 0 and channels % 4 == 0 and N % 4 == 0


def _bwd_h3(K: int, N: int, channels: int = 0) -> bool:
    return GEMM_MODE == "x6" and BWD_MODE == "h3" and K % 4 == 0 and channels % 4 == 0 and N % 4 == 0


# ---- guard regions (TEST seam; tests/conftest.py switches it on around every -m gpu test).  The arrays kernels publish into
# through plain pointers (partial maxima: `amax_publish`, an atomic on `slots + (slot & 1023)`; the per-(head, row) inverse scales
# of a head image) then get one extra row behind them that holds a sentinel, and `check_guards()` asserts it is untouched: a
# publish one section past an array (round 5: `sec = hcol / c_amax_sec` for a head past column N, csrc/gemm_h3i.hip) lands there
# instead of in a neighbour's allocation.  (An atomic max of 0.0 -- what a head whose operands are all out of range publishes --
# changes no value anywhere; the guard sees every write that could have changed a result.)
GUARD = False
GUARD_VALUE = 1.0e-30
_guards: list = []


def _guarded(rows: int, cols: int, device, what: str, keep: bool = False) -> torch.Tensor:
    """(rows, cols) fp32, uninitialised, with a sentinel row behind it (GUARD mode only); `keep`: a long-lived array, checked by
    every later `check_guards()` too"""
    if torch.cuda.is_current_stream_capturing():
        # memory of a graph's private pool is handed out again when the graph is dropped (TrainStep's shape-keyed cache evicts):
        # a sentinel there would be "overwritten" by whoever gets the block next.  Captures run unguarded; the eager steps in
        # front of every capture launch the same kernels on the same shapes.
        return torch.empty(rows, cols, dtype=torch.float32, device=device)
    a = torch.empty(rows + 1, cols, dtype=torch.float32, device=device)
    a[rows].fill_(GUARD_VALUE)
    _guards.append((what, a, keep))
    return a[:rows]


def check_guards() -> None:
    """every sentinel row handed out since the last call is intact (raises AssertionError naming the arrays otherwise)"""
    bad = [what for what, a, _ in _guards if not bool((a[-1] == GUARD_VALUE).all())]
    _guards[:] = [g for g in _guards if g[2]]
    if bad:
        raise AssertionError(f"a kernel wrote behind: {sorted(set(bad))}")


def _amax(t: torch.Tensor) -> torch.Tensor:
    """AMAX_SLOTS partial maxima of |t| (device), the dynamic pre-scale input of the fp16x3 gradient GEMMs: the array the kernel
    that produced `t` left on it (`_ttts_amax`, see _amax_slots), or a separate pass over `t`."""
    ready = getattr(t, "_ttts_amax", None)
    if ready is not None:
        return ready
    if not t.is_contiguous():
        t = t.contiguous()
    out = _guarded(1, AMAX_SLOTS, t.device, "amax partials")[0] if GUARD else torch.empty(AMAX_SLOTS, dtype=torch.float32, device=t.device)
    _lib.check(_lib.load().ttts_amax_partials(_p(t), t.numel(), _p(out), _stream()), "ttts_amax_partials")
    return out


class _AmaxArena:
    """Per-device pool of AMAX_SLOTS-float partial-maxima arrays that ONE memset zeroes for a whole training step.  ~150 kernels
    of a step (every producer of an fp16x3 operand: GEMM epilogues, LayerNorm, BatchNorm, positional encoding, attention,
    the backward masks) fill such an array with atomic maxima and each would otherwise need a memset of its own.  Slices
    are handed out in call order and stay valid until the next reset (a forward's maxima are read again in backward)."""
    SLICES = 512

    def __init__(self, device):
        self.buf = _guarded(self.SLICES, AMAX_SLOTS, device, "amax arena", keep=True) if GUARD else \
            torch.empty(self.SLICES, AMAX_SLOTS, dtype=torch.float32, device=device)
        self.next = 0
        self.clean = False          # True between reset() and release(): unissued slices are known to be zero

    def reset(self) -> None:
        _lib.check(_lib.load().ttts_zero(_p(self.buf), self.buf.numel() * 4, _stream()), "ttts_zero")
        self.next, self.clean = 0, True

    def take(self, n: int = 1):
        """one zeroed slice, or (n > 1) n consecutive ones as an (n, AMAX_SLOTS) view"""
        if not self.clean or self.next + n > self.SLICES:
            return None
        self.next += n
        return self.buf[self.next - 1] if n == 1 else self.buf[self.next - n:self.next]


_amax_arenas = {}


_pass_id = [0]          # micro-batch passes begun (amax_arena_reset): lets per-pass host state tell one capture pass from the next


def amax_arena_reset(device) -> None:
    """Call once at the start of a step (step.TrainStep does); everything handed out before is dead by then."""
    _pass_id[0] += 1
    arena = _amax_arenas.get(device)
    if arena is None:
        arena = _amax_arenas[device] = _AmaxArena(device)
    arena.reset()


def amax_arena_release(device) -> None:
    """At the end of the step: arrays handed out from here on are zeroed individually again."""
    arena = _amax_arenas.get(device)
    if arena is not None:
        arena.clean = False


def _amax_slots(device, zero: bool) -> torch.Tensor:
    """An AMAX_SLOTS-float array for a producing kernel to leave the partial maxima of its output in (attached to that
    output as `_ttts_amax`; a tensor attribute survives the hop to the next autograd Function, and when it does not the
    consumer simply runs the separate pass).  `zero`: the producer fills it with atomic maxima."""
    if zero:
        arena = _amax_arenas.get(device)
        got = arena.take() if arena is not None else None
        if got is not None:
            return got
    a = _guarded(1, AMAX_SLOTS, device, "amax slots")[0] if GUARD else torch.empty(AMAX_SLOTS, dtype=torch.float32, device=device)
    if zero:
        _lib.check(_lib.load().ttts_zero(_p(a), AMAX_SLOTS * 4, _stream()), "ttts_zero")
    return a


def _amax_slots_n(device, n: int) -> torch.Tensor:
    """(n, AMAX_SLOTS) zeroed partial-maxima arrays, contiguous (the section maxima of a head-image output)"""
    arena = _amax_arenas.get(device)
    got = arena.take(n) if arena is not None else None
    if got is not None:
        return got.view(n, AMAX_SLOTS)
    a = _guarded(n, AMAX_SLOTS, device, "section maxima") if GUARD else torch.empty(n, AMAX_SLOTS, dtype=torch.float32, device=device)
    _lib.check(_lib.load().ttts_zero(_p(a), a.numel() * 4, _stream()), "ttts_zero")
    return a


# ----------------------------------------------------------------------------------------------- deferred reductions
# The parameter-gradient kernels end in a small reduction of their partials into the gradient sink.  With sinks (the
# results are not read before the pass ends) those ~90 launches are appended to a queue that belongs to the gradient
# bucket (`parallel.FlatGradBucket.queue`, a caller-owned `ttts_reduce_queue` of the C ABI) and run as two launches when
# the queue is flushed: at the end of the backward pass (the first deferring node of a pass registers a final callback with
# the autograd engine, so `p.grad` is complete when `backward()` returns -- Lightning clips gradients right there), and
# again, defensively, before a collective over the bucket and before the optimizer reads it.  Nothing is process-global:
# the queue and its "callback registered" flag belong to the bucket, and `FlatGradBucket.zero()` drops whatever a backward
# pass that raised (its callback never ran) left behind.
DEFER_REDUCE = True          # tests flip it through `cross_check_forms(defer_reduce=False)`


WGRAD_GROUPS = True          # small weight gradients of a backward pass run as grouped launches (ReduceQueue.defer_wgrad)
WGRAD_GROUP_MAX = 4          # WG_GROUP_MAX of csrc/gemm_common.h
# members per launch: class 1 (4-wave 128 x 128 tile: out-projections, cross-attention q, the encoder's linears) fills up; class 2
# (8-wave 256 x 256 LDS-DMA tile) takes the three big weights of ONE decoder layer -- FFN2, FFN1, the packed in-projection, 11
# tiles: 23 row splits each on 253 workgroups -- so that no layer's last member is left to run alone
WGRAD_GROUP_SIZE = {1: 4, 2: 3, 3: 2, 4: 2}     # (classes 3 / 4: the two 80-channel weights of each kind -- a linear and a convolution)
# The grouped launches depend on nothing later in backward and nothing depends on them before the optimizer: on a SIDE stream they
# run beside the data-gradient chain and fill the CUs its launches leave idle (partial last rounds of the persistent GEMMs, the
# memory-bound normalisation kernels between them); joined where the queue is flushed.  Captured as a parallel graph branch.
WGRAD_SIDE_STREAM = False
WGRAD_SIDE_MAX_ROWS = 0      # > 0: only groups over at most this many rows take the side stream (the small-batch / encoder groups)


class ReduceQueue:
    """Host-side queue of deferred second-stage reductions (include/ttts_hip.h, ttts_reduce_queue) plus the workspaces
    the queued reductions still read -- and, in front of it, the small weight gradients that wait to be launched as a GROUP
    (`defer_wgrad`): a parameter gradient has no reader before the optimizer, so the 256 x 256-class outputs of a layer
    (out-projections, the cross-attention query projection, the encoder's linears) are collected as backward produces their
    operands and run as ONE grid of up to WGRAD_GROUP_MAX problems (ttts_linear_bwd_weight_h3_group): 1/n of the partial sums,
    n times the rows per workgroup, 1/n of the launches."""

    def __init__(self):
        self._lib = _lib.load()
        self.handle = c_void_p(self._lib.ttts_reduce_queue_create())
        if not self.handle:
            raise MemoryError("ttts_reduce_queue_create failed")
        self.keep: list = []
        self.wg = {1: [], 2: [], 3: [], 4: []}    # pending members of the next grouped launch, per class (ttts_wgrad_group_ok)
        self._armed = False         # a final callback of the running backward pass will flush
        self._side = None           # side stream of the grouped launches (WGRAD_SIDE_STREAM) and whether it has unjoined work
        self._side_busy = False

    def _arm(self) -> None:
        if not self._armed:
            torch.autograd.Variable._execution_engine.queue_callback(self._final)
            self._armed = True

    def arg(self, ws: torch.Tensor):
        """`queue` argument for an entry point whose reduction may wait for the flush; keeps `ws` alive until then.
        Called from backward nodes only (the engine's final-callback queue is open there)."""
        self._arm()
        self.keep.append(ws)
        return self.handle

    def defer_wgrad(self, cls: int, dy, dy_am, x, x_am, dw, db, M: int, N: int, K: int, taps: int = 1, T: int = 0,
                    row_shift: int = 0) -> None:
        """dw (+)= dy^T x, db (+)= column sums of dy (taps > 1: a convolution's weight gradient over utterances of T rows) --
        later, in a grouped launch of class `cls` (backward nodes only).  Members of one group share a row count: their
        workgroups then walk equally long row ranges."""
        self._arm()
        pend = self.wg[cls]
        if pend and (pend[0][6] != M or pend[0][0].device != dy.device):
            self.launch_wgrads(cls)
            pend = self.wg[cls]
        ws = _ws(self._lib.ttts_wgrad_workspace_bytes(M, N, K, taps), x.device)
        pend.append((dy, dy_am, x, x_am, dw, db, M, N, K, ws, taps, T, row_shift))
        if len(pend) >= WGRAD_GROUP_SIZE[cls]:
            self.launch_wgrads(cls)

    def launch_wgrads(self, cls: int = 0) -> None:
        """launch the pending group(s) now, on the current stream (their reductions join the queue)"""
        if cls == 0:
            for c in (1, 2, 3, 4):
                self.launch_wgrads(c)
            return
        if not self.wg[cls]:
            return
        import ctypes
        wg, self.wg[cls] = self.wg[cls], []
        n = len(wg)
        PA, ZA, LA, IA = ctypes.c_void_p * n, ctypes.c_size_t * n, ctypes.c_int64 * n, ctypes.c_int * n
        col = lambda i: [m[i] for m in wg]      # noqa: E731
        ptr = lambda ts: PA(*[(t.data_ptr() if t is not None else None) for t in ts])      # noqa: E731
        def go(stream):
            _lib.check(self._lib.ttts_wgrad_group(
                n, ptr(col(0)), ptr(col(2)), ptr(col(4)), ptr(col(5)), ptr(col(9)), ZA(*[m[9].numel() * 4 for m in wg]), LA(*col(6)),
                IA(*col(7)), IA(*col(8)), IA(*col(10)), IA(*col(11)), IA(*col(12)), 1, ptr(col(1)), ptr(col(3)),
                self.handle if DEFER_REDUCE else None, stream), "ttts_wgrad_group")
        if WGRAD_SIDE_STREAM and DEFER_REDUCE and (WGRAD_SIDE_MAX_ROWS <= 0 or wg[0][6] <= WGRAD_SIDE_MAX_ROWS):
            dev = wg[0][0].device
            if self._side is None or self._side.device != dev:
                self._side = torch.cuda.Stream(device=dev)
            cur = torch.cuda.current_stream(dev)
            self._side.wait_stream(cur)                # the operands (and the zeroed sinks) are ready
            with torch.cuda.stream(self._side):
                go(c_void_p(self._side.cuda_stream))
            self._side_busy = True
            # the operands were allocated on the main stream: they must outlive the side stream's reads -- kept until the join
            self.keep.extend(t for m in wg for t in (m[0], m[1], m[2], m[3]))
        else:
            go(_stream())
        self.keep.extend(m[9] for m in wg)

    def join_side(self) -> None:
        if self._side_busy:
            torch.cuda.current_stream(self._side.device).wait_stream(self._side)
            self._side_busy = False

    def _final(self) -> None:
        self._armed = False
        self.flush()

    def pending(self) -> int:
        return int(self._lib.ttts_reduce_queue_pending(self.handle))

    def flush(self) -> None:
        """Run every queued reduction now, on the current stream (the pending grouped weight gradients first)."""
        self.launch_wgrads()
        self.join_side()            # (the queued reductions read the partial sums the side stream writes)
        if self.keep or self.pending():
            _lib.check(self._lib.ttts_reduce_queue_flush(self.handle, _stream()), "ttts_reduce_queue_flush")
            self.keep.clear()

    def clear(self) -> None:
        _lib.check(self._lib.ttts_reduce_queue_clear(self.handle), "ttts_reduce_queue_clear")
        self.join_side()
        self.keep.clear()
        self.wg = {1: [], 2: [], 3: [], 4: []}
        self._armed = False

    def __del__(self):
        try:
            if self.handle:
                self._lib.ttts_reduce_queue_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


def _qarg(queue: Optional[ReduceQueue], ws: torch.Tensor):
    return queue.arg(ws) if (queue is not None and DEFER_REDUCE) else None


def _wgrad_is_split(N: int, K: int) -> bool:
    """Mirror of wgrad_use_x6 (csrc/gemm.hip): shapes the split-precision weight-gradient kernels take (the others run on
    the fp32-MFMA kernel and need no maxima)."""
    return (N >= 128 or N in (80, 96)) and (K >= 128 or K in (80, 96)) and (N >= 128 or K >= 128)


def _attn_bwd(lib, do, dq_am, dkv_am, q_am, k_am, v_am, rowstat, *args):
    """Attention backward in the configured form; `args` = every C-ABI argument up to step_seed.  dq_am / dkv_am: zeroed
    AMAX_SLOTS-slot arrays in which the fp16x3 kernels leave max|dq| / max|dk, dv| (the in-projection gradients consume them);
    q_am / k_am / v_am: the partial maxima of the forward operands (their dynamic pre-scales)."""
    if ATTN_BWD_MODE == "h3":
        return lib.ttts_attention_bwd_h3(*args, _p(_amax(do)), _p(dq_am), _p(dkv_am), _p(q_am), _p(k_am), _p(v_am), _p(rowstat),
                                         _stream())
    return (lib.ttts_attention_bwd_x6 if ATTN_BWD_MODE == "x6" else lib.ttts_attention_bwd)(*args, _stream())


def _wgrad(lib, name: str, dy: torch.Tensor, amax, x: torch.Tensor, x_amax, split_ok: bool, queue, *args):
    """Weight-gradient entry point `name` in the configured form; `args` = everything between (dy, x) and `queue`.
    The fp16x3 form takes the partial maxima of |dy| and |x| (computed here unless the caller already has them); shapes
    the split kernels do not take (`split_ok` False) run on the fp32-MFMA kernel behind the _x6 entry point and need none."""
    if WGRAD_MODE == "h3" and split_ok:
        am = amax if amax is not None else _amax(dy)
        xm = x_amax if x_amax is not None else _amax(x)
        return getattr(lib, name + "_h3")(_p(dy), _p(x), *args, _p(am), _p(xm), queue, _stream())
    return getattr(lib, name + ("_x6" if WGRAD_MODE in ("x6", "h3") else ""))(_p(dy), _p(x), *args, queue, _stream())


_param_epoch = 0


def bump_param_epoch() -> None:
    """Invalidate every cached weight split.  Called by optimizers that update parameters through raw pointers
    (optim.FlatAdam), which autograd's per-tensor version counters cannot see."""
    global _param_epoch
    _param_epoch += 1


class _PlaneEntry:
    __slots__ = ("wref", "off", "mode", "rows", "cols", "c2", "taps", "planes", "tag")


_BATCHED_SPLIT = True
_plane_entries: list = []        # every (weight, mode) split so far, for the one-launch refresh after an optimizer step
_plane_tables: dict = {}         # flat-storage address -> (signature, device descriptor table, pinned host copy, total blocks)


def _split_units(rows: int, cols: int, mode: int, c2: int) -> int:
    """work units of one entry of the batched refresh; a LINEAR entry's c2 / taps may hold the geometry of a stacked image
    (64-bit packed, `_StackedPlanes`), which is no part of the count"""
    conv = mode >= 4 and ((mode - 4) & 3) >= 2 or mode in (2, 3)
    return int(_lib.load().ttts_weight_split_units(rows, cols, mode, c2 if conv else 0))


def _refresh_all_planes(storage: int) -> None:
    """Re-split every registered weight that lives in the flat parameter storage at address `storage` (one model under
    one FlatAdam) and whose storage and version are unchanged (only the parameter epoch moved, i.e. an optimizer stepped
    through raw pointers) with ONE launch instead of one per weight and mode.  Weights of other models are left alone:
    their owners may be using them on another stream."""
    global _plane_entries
    live, mine, sig = [], [], []
    for e in _plane_entries:
        w = e.wref()
        if w is None:
            continue
        live.append(e)
        if w.untyped_storage().data_ptr() != storage:
            continue
        mine.append(e)
        if e.tag[0] == w._version and e.tag[1] == w.data_ptr() + e.off and e.tag[2] != _param_epoch:
            sig.append((e.tag[1], e.planes.data_ptr(), e.rows, e.cols, e.mode, e.c2, e.taps))
    _plane_entries = live
    for k in [k for k in _plane_tables if not any(e.wref() is not None and e.wref().untyped_storage().data_ptr() == k for e in live)]:
        del _plane_tables[k]
    if not sig:
        return
    sig_t = tuple(sig)
    table = _plane_tables.get(storage)
    if table is None or table[0] != sig_t:
        rows, blk = [], 0
        for s in sig:
            rows.append(list(s) + [blk])
            blk += _split_units(s[2], s[3], s[4], s[5])
        host = torch.tensor(rows, dtype=torch.int64).pin_memory()      # page-locked: the upload does not synchronise
        table = _plane_tables[storage] = (sig_t, host.to(mine[0].planes.device, non_blocking=True), host, blk)
    lib = _lib.load()
    _lib.check(lib.ttts_weight_split_batched(_p(table[1]), len(sig), table[3], _stream()), "ttts_weight_split_batched")
    refreshed = {s[1] for s in sig}
    for e in mine:
        if e.planes.data_ptr() in refreshed:
            e.tag = (e.tag[0], e.tag[1], _param_epoch)


class PlaneTable:
    """Every weight-plane image of ONE module's parameters and the descriptor table of their batched refresh.

    `step.TrainStep` owns one: it refreshes the planes explicitly at the start of a step (one launch sequence, recorded in
    the step's HIP graph) instead of relying on the lazy, process-wide refresh of `_planes`.  The table tensor, its pinned
    host copy and (through the entries) the plane buffers are referenced from here, so a captured graph that reads them
    stays valid for as long as its TrainStep lives, whatever other models in the process do."""

    def __init__(self, module: torch.nn.Module):
        self.entries = []
        seen = set()
        for prm in module.parameters():
            for ent in (getattr(prm, "_ttts_planes", None) or {}).values():
                if id(ent) not in seen:
                    seen.add(id(ent))
                    self.entries.append((prm, ent))
        if not self.entries:
            raise RuntimeError("PlaneTable: the module has no weight planes yet (run one forward + backward first)")
        rows, blk = [], 0
        self.sig = []
        for prm, e in self.entries:
            src = prm.data_ptr() + e.off
            self.sig.append((src, e.planes.data_ptr(), prm._version))
            rows.append([src, e.planes.data_ptr(), e.rows, e.cols, e.mode, e.c2, e.taps, blk])
            blk += _split_units(e.rows, e.cols, e.mode, e.c2)
        self.blocks = blk
        self.host = torch.tensor(rows, dtype=torch.int64).pin_memory()
        self.dev = self.host.to(self.entries[0][1].planes.device, non_blocking=True)

    def valid(self) -> bool:
        """The parameters still live where the table says (no `.to()`, no re-allocated planes, no in-place autograd edit)."""
        return all((prm.data_ptr() + e.off, e.planes.data_ptr(), prm._version) == sg
                   for (prm, e), sg in zip(self.entries, self.sig))

    def stale(self) -> bool:
        return any(e.tag[2] != _param_epoch for _, e in self.entries)

    def refresh(self) -> None:
        """Re-split every weight of the module with one batched launch sequence and mark the planes current."""
        _lib.check(_lib.load().ttts_weight_split_batched(_p(self.dev), len(self.entries), self.blocks, _stream()),
                   "ttts_weight_split_batched")
        for prm, e in self.entries:
            e.tag = (prm._version, prm.data_ptr() + e.off, _param_epoch)

    def mark_current(self) -> None:
        """Before capturing a micro-batch that does NOT begin an accumulation window: the window's first graph refreshes the
        planes when it replays, so the launches recorded here must use them as they are (no per-weight re-split nodes)."""
        for prm, e in self.entries:
            e.tag = (prm._version, prm.data_ptr() + e.off, _param_epoch)

    def mark_stale(self) -> None:
        """After a capture: the recorded refresh has not run, so the host must not believe the planes are current."""
        for _, e in self.entries:
            e.tag = (e.tag[0], e.tag[1], -1)


def _planes(w: torch.Tensor, mode: int, rows: int, cols: int, c2: int = 0, taps: int = 0) -> torch.Tensor:
    """hi/mid/lo bf16 planes of a weight, re-laid as the K-contiguous B operand (ttts_weight_split).  Cached on the
    parameter object and keyed by its version counter + storage address + parameter epoch, so the two forwards and the
    backward of a step split each weight once; after an optimizer step all weights are re-split in one launch.
    Row slices made by `param_rows` are cached on the parameter they were cut from."""
    owner = getattr(w, "_ttts_planes_owner", None)
    holder, sub = owner if owner is not None else (w, 0)
    cache = getattr(holder, "_ttts_planes", None)
    tag = (holder._version, w.data_ptr(), _param_epoch)
    key = (mode, sub)
    ent = cache.get(key) if cache is not None else None
    if ent is not None:
        if ent.tag == tag:
            return ent.planes
        if _BATCHED_SPLIT and ent.tag[0] == tag[0] and ent.tag[1] == tag[1] and not torch.cuda.is_current_stream_capturing():
            _refresh_all_planes(holder.untyped_storage().data_ptr())   # only the epoch moved: refresh this model at once (never inside a capture:
            #                                       the process-wide table is not owned by the graph, see PlaneTable)
            if ent.tag == tag:
                return ent.planes
    lib = _lib.load()
    nwords = (int(lib.ttts_split_image_bytes(rows, cols, mode, c2, taps)) + 1) // 2      # (fp16x3 images pad channels to 32)
    if ent is None or ent.planes.numel() != nwords or ent.planes.device != w.device:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("weight planes must exist before a HIP-graph capture: run the eager warm-up steps first")
        planes = torch.empty(nwords, dtype=torch.int16, device=w.device)
    else:
        planes = ent.planes
    _lib.check(lib.ttts_weight_split(_p(w), _p(planes), rows, cols, mode, c2, taps, _stream()), "ttts_weight_split")
    try:
        if cache is None:
            cache = {}
            holder._ttts_planes = cache
        if ent is None:
            ent = _PlaneEntry()
            ent.wref = weakref.ref(holder)
            _plane_entries.append(ent)
            cache[key] = ent
        ent.off = w.data_ptr() - holder.data_ptr()
        ent.mode, ent.rows, ent.cols, ent.c2, ent.taps, ent.planes, ent.tag = mode, rows, cols, c2, taps, planes, tag
    except (AttributeError, TypeError):
        pass
    return planes


class _ParamRowsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p, r0, r1):
        ctx.cfg = (p.shape, r0, r1)
        ctx.set_materialize_grads(False)
        return p.detach()[r0:r1]

    @staticmethod
    def backward(ctx, g):
        if g is None:                      # the consumer added its gradient straight into the parameter's sink
            return None, None, None
        shape, r0, r1 = ctx.cfg
        full = torch.zeros(shape, dtype=g.dtype, device=g.device)
        full[r0:r1] = g
        return full, None, None


def param_rows(p: torch.Tensor, r0: int, r1: int) -> torch.Tensor:
    """Rows r0..r1 of a parameter (e.g. the q or k/v part of a packed in-projection) as an operand of `linear`: unlike a
    plain slice, the result keeps the parameter's gradient sink (so the weight-gradient kernel adds in place and no
    slice-backward kernels run) and its weight planes are cached on the parameter."""
    v = _ParamRowsFn.apply(p, r0, r1)
    sk = _sink(p)
    if sk is not None:
        v._ttts_grad_sink = sk[r0:r1]
        v._ttts_reduce_queue = getattr(p, "_ttts_reduce_queue", None)
    if p.dim() == 2:
        v._ttts_planes_owner = (p, r0)
    return v


def _sink(t: Optional[torch.Tensor]):
    """Gradient sink of a parameter: a slice of the flat data-parallel gradient bucket (parallel.FlatGradBucket).
    When every parameter of an op has one, the backward kernels ADD their result straight into it
    (`accumulate=1`) and autograd gets None -- no per-tensor accumulation kernels, no flatten copy."""
    return None if t is None else getattr(t, "_ttts_grad_sink", None)


def _sinks(*params):
    """(list of destinations or None, accumulate flag, reduction queue or None): sinks are used only if all given
    parameters have one; the queue is the one their bucket attached (`_ttts_reduce_queue`), when they share it."""
    live = [p for p in params if p is not None]
    sk = [_sink(p) for p in live]
    if live and all(x is not None for x in sk):
        qs = {id(getattr(p, "_ttts_reduce_queue", None)) for p in live}
        q = getattr(live[0], "_ttts_reduce_queue", None) if len(qs) == 1 else None
        return [(_sink(p) if p is not None else None) for p in params], 1, q
    return None, 0, None


# ----------------------------------------------------------------------------------------------- linear
# Image operands (csrc/gemm_h3i.hip): a producer whose kernel holds whole rows (LayerNorm forward / backward) leaves, next to its
# fp32 output, the activation already split into f16 hi / lo with a per-row power-of-two scale -- `t._ttts_image = (image,
# row_inv)` -- and the GEMM that reads it stages both operands by LDS-DMA on 128 x 256 tiles, two workgroups per CU.  Pays from
# a few thousand rows on (512 resident workgroups) and for outputs at least half a tile wide.
IMAGE_MIN_ROWS = 8192
# Measured in the training step (same-box A/B, profiles/r04_image_ab.txt): per launch the image kernel is 1.06-1.26x faster than
# gemm_h3, but the image is 4 more bytes per element for its producer to write (LayerNorm forward 16.6 -> 26.1 us, backward 26.9 ->
# 38.4 us at 55 680 x 256) and every LayerNorm of this model feeds exactly ONE large GEMM: the step came out 0.3 ms SLOWER.  So
# LayerNorm leaves images only when asked (tests, `layer_norm(..., emit_image=True)`); what the step uses by default is the same
# kernel on the plain fp32 operand (`_dma_shape_ok`), where it needs no producer and still wins.
LAYERNORM_IMAGES = False


def _image_rows_ok(M: int, d: int) -> bool:
    return M >= IMAGE_MIN_ROWS and d in (256, 512, 1024)


def _image_shape_ok(M: int, red: int, out: int) -> bool:
    """M rows, reduction depth `red`, output width `out`"""
    return M >= IMAGE_MIN_ROWS and red % 32 == 0 and red >= 32 and out >= 128 and out % 4 == 0


def _dma_shape_ok(M: int, red: int, out: int, gate: bool) -> bool:
    """shapes on which the LDS-DMA kernel beats gemm_h3 with a plain fp32 activation (ttts_linear_*_h3d; tools/h3i_bench.py,
    h3 time / DMA time): data gradients with a relu gate, 256 -> 1024 columns: 1.09x at M = 13 920, 1.12x at 27 840, 1.15x at
    55 680 (512 -> 2048 at 27 840: 0.99x, not taken); square projections when their 128 x 256 tiles fill the chip: 256 -> 256
    0.72x at M = 13 920, 0.95x at 27 840, 1.04x at 41 760 (326 tiles), 1.07x at 55 680; 512 -> 512 1.11x at 27 840 (436 tiles);
    forward 256 -> 1024 on the 256-ROW tile of the same kernel (one 8-wave workgroup per CU, chosen inside the library when 256
    of its tiles exist): 1.06x at 55 680, step -0.07 ms same-box (the 128-row tile is level there; 512 -> 2048 at 27 840 level on
    either; no difference at M = 13 920); the gated 256 -> 1024 data gradient on it too (round 6: step -0.02 ms same-box).  Level or
    slower elsewhere (K = 1024: 0.9x)."""
    if not (DMA_GEMMS and M >= IMAGE_MIN_ROWS and red % 32 == 0 and out % 4 == 0):
        return False
    if gate and red <= 256 and out >= 512:
        return True
    if DMA_BIG_FWD and not gate and red <= 256 and out >= 1024 and -(-M // 256) * -(-out // 256) >= 256:
        return True
    return red == out and out in (256, 512) and -(-M // 128) * -(-out // 256) >= DMA_MIN_TILES


DMA_MIN_TILES = 320


def _dma_dims(red: int, out: int) -> bool:
    """could `_dma_shape_ok` say yes for SOME row count?  The choice between the two kernels (and so between the two weight
    images, modes 4 / 8 resp. 5 / 9) depends on the batch's row count, and a captured graph can only use images that exist:
    `_both_images` makes both in the eager steps whenever the dimensions are eligible, whatever the current row count."""
    return DMA_GEMMS and red % 32 == 0 and out % 4 == 0 and ((red <= 256 and out >= 512) or (red == out and out in (256, 512)))


def _image_dims(red: int, out: int) -> bool:
    """could `_image_shape_ok` say yes for SOME row count while producers leave images (LAYERNORM_IMAGES)?  Same reason as
    `_dma_dims`: the k16 weight image must exist before a capture that lands on the other side of IMAGE_MIN_ROWS."""
    return LAYERNORM_IMAGES and red % 32 == 0 and red >= 32 and out >= 128 and out % 4 == 0


def _both_images(w: torch.Tensor, used: int, other: int, rows: int, cols: int) -> torch.Tensor:
    """planes of `w` in mode `used`; outside a capture also make sure its sibling image (`other`) exists"""
    if not torch.cuda.is_current_stream_capturing():
        _planes(w, other, rows, cols)
    return _planes(w, used, rows, cols)
DMA_GEMMS = True
DMA_BIG_FWD = True          # 256 -> 1024 forward GEMMs (FFN1) on the 256-row LDS-DMA tile when its tiles fill the chip (tools/h3i_bench.py)


def _new_image(M: int, d: int, device):
    return (torch.empty(M * d * 2, dtype=torch.int16, device=device), torch.empty(M, dtype=torch.float32, device=device))


class LinearFn(torch.autograd.Function):
    """y = drop(act(x @ w.T + b)) + residual, rows optionally shifted by `row_shift` inside each utterance."""

    @staticmethod
    def forward(ctx, x, w, b, residual, act, drop_p, seed, row_shift, T, tok_out=None, tok_in=None, skip_in=None,
                skip_out=None, tok_drop=None, x_amax=None, y_amax=None, x_image=None, y_himg=None, twin=None):
        """x_amax: partial maxima of |x| (fp16x3 forms; None: measured here); y_amax: None, or a zeroed AMAX_SLOTS-slot array
        that receives max|y| (the wrapper attaches it to y for the next fp16x3 consumer); x_image: None, or (image, row_inv) of
        x left by its producer -- the GEMM then takes the image-operand kernel (ttts_linear_fwd_h3i)."""
        lib = _lib.load()
        x = _chk(x, "linear.x")
        w = _chk(w, "linear.weight")
        N, K = w.shape
        if x.shape[-1] != K:
            raise ValueError(f"linear: x has {x.shape[-1]} features, weight expects {K}")
        if act == ACT_RELU and residual is not None:
            raise ValueError("linear: relu epilogue cannot be combined with a residual")
        M = x.numel() // K
        b_ = _chk(b, "linear.bias") if b is not None else None
        r_ = _chk(residual, "linear.residual") if residual is not None else None
        if twin is not None:
            # x (and the residual) are the first halves of 2 B-utterance buffers: the kernel runs on the buffers -- same pointers,
            # twice the rows --, autograd gets the first half of the output (`twin`: [x buffer, residual buffer or None, out list])
            if x_image is not None or row_shift != 0:
                raise ValueError("linear: a twin batch takes plain fp32 operands without a row shift")
            M = 2 * M
            y_full = torch.empty(2 * x.shape[0], *x.shape[1:-1], N, dtype=torch.float32, device=x.device)
            y = y_full[:x.shape[0]]
            twin[2].append(y_full)
        else:
            y = torch.empty(*x.shape[:-1], N, dtype=torch.float32, device=x.device)
        if r_ is not None and r_.shape != y.shape:
            raise ValueError("linear: residual shape mismatch")
        if y_himg is not None:
            # the output leaves as a HEAD IMAGE (attention in-projections: ttts_linear_fwd_h3d_img); y_himg = (row_inv, section maxima,
            # columns per section).  `y` keeps its fp32 geometry and dtype, but its cells hold f16 hi / lo pairs: only the attention
            # kernels may read it (`linear` hands it out wrapped in a HeadImage, which is not a Tensor).
            if x_amax is None:
                x_amax = _amax(x)
            row_inv, sec_amax, sec_cols = y_himg
            _lib.check(lib.ttts_linear_fwd_h3d_img(_p(x), _p(_both_images(w, 8, 4, N, K)), _p(b_), _p(y), _p(row_inv), M, N, K, _p(x_amax),
                                                   _p(sec_amax), sec_cols, _stream()), "ttts_linear_fwd_h3d_img")
        elif x_image is not None and _fwd_h3(K, N) and row_shift == 0:
            _lib.check(lib.ttts_linear_fwd_h3i(_p(x_image[0]), _p(x_image[1]), _p(_both_images(w, 8, 4, N, K)), _p(b_), _p(r_), _p(y), M, N, K,
                                               act, float(drop_p), seed, _ss(), _p(y_amax), _stream()), "ttts_linear_fwd_h3i")
        elif _fwd_h3(K, N):
            if x_amax is None:
                x_amax = _amax(x)
            both = row_shift == 0 and (_dma_dims(K, N) or _image_dims(K, N))
            if row_shift == 0 and _dma_shape_ok(M, K, N, False):
                _lib.check(lib.ttts_linear_fwd_h3d(_p(x), _p(_both_images(w, 8, 4, N, K) if both else _planes(w, 8, N, K)), _p(b_), _p(r_),
                                                   _p(y), M, N, K, act,
                                                   float(drop_p), seed, _ss(), _p(x_amax), _p(y_amax), _stream()), "ttts_linear_fwd_h3d")
            else:
                _lib.check(lib.ttts_linear_fwd_h3(_p(x), _p(_both_images(w, 4, 8, N, K) if both else _planes(w, 4, N, K)), _p(b_), _p(r_),
                                                  _p(y), M, N, K, act,
                                                  float(drop_p), seed, _ss(), row_shift, T, _p(x_amax), _p(y_amax), _stream()),
                           "ttts_linear_fwd_h3")
        elif GEMM_MODE == "x6":
            _lib.check(lib.ttts_linear_fwd_x6(_p(x), _p(_planes(w, 0, N, K)), _p(b_), _p(r_), _p(y), M, N, K, act,
                                              float(drop_p), seed, _ss(), row_shift, T, _stream()), "ttts_linear_fwd_x6")
        else:
            _lib.check(lib.ttts_linear_fwd(_p(x), _p(w), _p(b_), _p(r_), _p(y), M, N, K, act, float(drop_p), seed,
                                           _ss(), row_shift, T, _stream()), "ttts_linear_fwd")
        ctx.save_for_backward(x, w, y if act == ACT_RELU else None)
        ctx.cfg = (act, float(drop_p), seed, row_shift, T, b is not None, residual is not None)
        ctx.sinks = _sinks(w, b)
        ctx.x_amax = x_amax         # the weight gradient reads x again (same pre-scale)
        ctx.ss = _ss()              # backward regenerates the dropout mask under the step-state word of ITS forward
        ctx.toks = (tok_out, tok_in, skip_in, skip_out)
        ctx.tok_drop = tok_drop
        # x is the output of an attention kernel and feeds nothing but this Linear: the gradient this backward returns for it
        # reaches the attention backward as it is (never summed with another one), so its maxima can ride on the tensor
        ctx.sole_consumer = bool(getattr(x, "_ttts_sole_consumer", False))
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, w, y = ctx.saved_tensors
        act, drop_p, seed, row_shift, T, has_b, has_r = ctx.cfg
        tok_out, tok_in, skip_in, skip_out = ctx.toks
        N, K = w.shape
        M = x.numel() // K
        dy = _chk(dy, "linear.dy")
        # partial maxima of |dacc| (dynamic pre-scale of the fp16x3 data / weight gradients): emitted by the mask kernel
        # that produces dacc when there is one, by a separate pass otherwise
        want_am = (ctx.needs_input_grad[0] and _bwd_h3(N, K)) or \
                  (ctx.needs_input_grad[1] and WGRAD_MODE == "h3" and _wgrad_is_split(N, K))
        am = None
        dacc_image = None
        if act == ACT_RELU and tok_out is not None and tok_out.premasked:
            dacc = dy                    # the consumer's data-gradient epilogue already applied the relu / dropout mask
            tok_out.premasked = False
        elif act == ACT_RELU:
            dacc = torch.empty_like(dy)
            am = _amax_slots(dy.device, True) if want_am else None
            _lib.check(lib.ttts_relu_dropout_bwd(_p(dy), _p(y), _p(dacc), dy.numel(), drop_p, _p(am), _stream()),
                       "ttts_relu_dropout_bwd")
        elif drop_p > 0.0:
            td = ctx.tok_drop
            if td is not None and td.dacc is not None and td.dx is dy:
                dacc, am = td.dacc, td.amax          # written by the LayerNorm backward that produced dy
                dacc_image = td.image                # ... with its image operand, when the shape takes the image kernel
                td.dx = td.dacc = td.amax = td.image = None
            else:
                dacc = torch.empty_like(dy)
                am = _amax_slots(dy.device, True) if want_am else None
                _lib.check(lib.ttts_dropout_bwd(_p(dy), _p(dacc), dy.numel(), drop_p, seed, ctx.ss, _p(am), _stream()),
                           "ttts_dropout_bwd")
        else:
            dacc = dy
            dacc_image = getattr(dy, "_ttts_image", None)     # a LayerNorm backward without residual dropout left it on its dx
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            if row_shift != 0:
                raise RuntimeError("linear: input gradient through a shifted loader is not needed on this path")
            dx = torch.empty_like(x)
            gate, gscale = (x, tok_in.scale) if tok_in is not None else (None, 1.0)
            skip = None                    # gradient of the block's skip connection, left here by the block's last Linear
            if skip_in is not None and skip_in.grad is not None:
                skip, skip_in.grad = skip_in.grad, None
                if skip.shape != x.shape or not skip.is_contiguous():
                    raise RuntimeError("linear: skip-connection gradient does not match the block input")
            if _bwd_h3(N, K):
                # dx with the producer's relu mask applied here is exactly the `dacc` of that producer's backward: leave its
                # maxima on it
                dx_am = _amax_slots(dx.device, True) if (tok_in is not None or ctx.sole_consumer) else None
                if dacc_image is not None and _image_shape_ok(M, N, K):
                    _lib.check(lib.ttts_linear_bwd_data_h3i(_p(dacc_image[0]), _p(dacc_image[1]), _p(_both_images(w, 9, 5, K, N)), _p(skip),
                                                            _p(dx), M, N, K, _p(gate), gscale, _p(dx_am), _stream()),
                               "ttts_linear_bwd_data_h3i")
                else:
                    am = am if am is not None else _amax(dacc)
                    both = _dma_dims(N, K) or _image_dims(N, K)
                    if _dma_shape_ok(M, N, K, gate is not None) and not (gate is not None and skip is not None):
                        _lib.check(lib.ttts_linear_bwd_data_h3d(_p(dacc), _p(_both_images(w, 9, 5, K, N) if both else _planes(w, 9, K, N)),
                                                                _p(skip), _p(dx), M, N, K,
                                                                _p(gate), gscale, _p(am), _p(dx_am), _stream()),
                                   "ttts_linear_bwd_data_h3d")
                    else:
                        _lib.check(lib.ttts_linear_bwd_data_h3(_p(dacc), _p(_both_images(w, 5, 9, K, N) if both else _planes(w, 5, K, N)),
                                                               _p(skip), _p(dx), M, N, K,
                                                               _p(gate), gscale, _p(am), _p(dx_am), _stream()),
                                   "ttts_linear_bwd_data_h3")
                if dx_am is not None:
                    dx._ttts_amax = dx_am
            elif GEMM_MODE == "x6":
                _lib.check(lib.ttts_linear_bwd_data_x6(_p(dacc), _p(_planes(w, 1, K, N)), _p(skip), _p(dx), M, N, K,
                                                       _p(gate), gscale, _stream()), "ttts_linear_bwd_data_x6")
            else:
                _lib.check(lib.ttts_linear_bwd_data(_p(dacc), _p(w), _p(skip), _p(dx), M, N, K, _p(gate), gscale, _stream()),
                           "ttts_linear_bwd_data")
            if tok_in is not None:
                tok_in.premasked = True
        if ctx.needs_input_grad[1]:
            sk, acc, queue = ctx.sinks
            cls = lib.ttts_wgrad_group_ok(M, N, K, 1) if (WGRAD_GROUPS and DEFER_REDUCE and sk is not None and queue is not None and
                                                       WGRAD_MODE == "h3" and _wgrad_is_split(N, K)) else 0
            if cls == 2 and row_shift != 0:
                cls = 0                      # (the LDS-DMA tile clips utterances of at least 16 rows only: keep the checked single launch)
            if cls:
                # an output with a gradient sink: nobody reads it before the optimizer, so it waits for its group
                queue.defer_wgrad(cls, dacc, am if am is not None else _amax(dacc), x,
                                  ctx.x_amax if ctx.x_amax is not None else _amax(x), sk[0], sk[1], M, N, K, 1, T if row_shift else 0,
                                  row_shift)
            else:
                nbytes = lib.ttts_wgrad_workspace_bytes(M, N, K, 1)
                ws = _ws(nbytes, x.device)
                if sk is not None:
                    dw_t, db_t = sk
                else:
                    dw_t = dw = torch.empty_like(w)
                    db_t = db = torch.empty(N, dtype=torch.float32, device=x.device) if has_b else None
                _lib.check(_wgrad(lib, "ttts_linear_bwd_weight", dacc, am, x, ctx.x_amax, _wgrad_is_split(N, K), _qarg(queue, ws),
                                  _p(dw_t), _p(db_t), _p(ws), ws.numel() * 4, M, N, K, row_shift, T, acc), "ttts_linear_bwd_weight")
        dres = dy if has_r else None
        if has_r and skip_out is not None:      # hand the skip gradient to the block's first Linear instead of autograd
            skip_out.grad, dres = dy, None
        return dx, dw, db, dres, None, None, None, None, None, None, None, None, None, None, None, None, None, None, None


# Test seam: called with the output of every relu-epilogue Linear, in call order (which units the HIP path gated off).
_relu_observer = None


class _ReluToken:
    """Handshake between a Linear with a relu(+dropout) epilogue and the ONE Linear that consumes its output: the
    consumer's data-gradient kernel applies the producer's backward mask in its epilogue and says so here."""
    __slots__ = ("scale", "premasked")

    def __init__(self, scale: float):
        self.scale, self.premasked = scale, False


class _DropToken:
    """Handshake between a Linear with a residual-dropout epilogue (y = drop(x W + b) + residual) and the LayerNorm that is
    the ONLY reader of y: the LayerNorm's backward kernel writes, next to its dx, the dropped copy the Linear's backward
    would compute from dx in a pass of its own (`ttts_dropout_bwd`), with its partial maxima."""
    __slots__ = ("p", "seed", "ss", "dx", "dacc", "amax", "image")

    def __init__(self, p: float, seed: int, ss):
        self.p, self.seed, self.ss = p, seed, ss
        self.dx = self.dacc = self.amax = self.image = None


class SkipToken:
    """Residual block `y = last(f(first(x))) + x` whose first and last ops are Linears: the last one's backward parks
    the skip-connection gradient here and the first one adds it in its data-gradient epilogue, instead of autograd
    summing two full-size tensors with a separate kernel.  Pass the same token as `skip_in` to the first Linear (its
    input must be the block input x) and as `skip_out` to the last one (its `residual` must be the same x)."""
    __slots__ = ("grad",)

    def __init__(self):
        self.grad = None


HEAD_IMAGES = True          # attention in-projections write head images and attention runs on them (csrc/attention_img.hip)
FUSED_CROSS_KV = True       # ONE K/V projection of the encoder memory for all decoder layers (cross_kv_projection)


class HeadImage:
    """What an attention in-projection returns when its output leaves as a HEAD IMAGE (`linear(..., head_image_sections=n)`,
    `cross_kv_projection`): per row and 64-column head {64 f16 hi, 64 f16 lo} of (x W^T + b) 2^e(row, head) in the 256 bytes
    the fp32 columns would occupy, 2^-e in `row_inv`.  Deliberately NOT a Tensor: the cells are typed float32 and hold f16 pairs,
    so anything but the attention kernels that read them (`self_attention`, `cross_attention`) would compute on garbage -- a
    slice, a clone, a hook or a finite-check cannot be taken of this object by accident.
      cells     the (B, T, width) tensor autograd tracks (row stride `ld` cells: a layer's window of a stacked projection)
      row_inv   (all heads of the projection, B * T) inverse scales; this handle's heads start at head `col0 // 64`
      sec_amax  (sections, AMAX_SLOTS) partial maxima per section of `sec_cols` columns of the projection
      col0      first column of this handle inside the projection its row_inv / sec_amax describe
      slab      None, or the gradient slab the producer wants this handle's gradient written into (`_KVGradSlab`)"""
    __slots__ = ("cells", "row_inv", "sec_amax", "sec_cols", "col0", "slab", "index")

    def __init__(self, cells, row_inv, sec_amax, sec_cols: int, col0: int = 0, slab=None, index: int = 0):
        self.cells, self.row_inv, self.sec_amax, self.sec_cols, self.col0, self.slab, self.index = \
            cells, row_inv, sec_amax, int(sec_cols), int(col0), slab, int(index)

    @property
    def shape(self):
        return self.cells.shape

    @property
    def device(self):
        return self.cells.device

    @property
    def ld(self) -> int:
        return self.cells.stride(-2)

    def inv_of(self, col: int):
        """pointer to the inverse scales of the head that starts at column `col` of this handle"""
        rows = self.row_inv.shape[1]
        return _off(self.row_inv, ((self.col0 + col) // 64) * rows)

    def amax_of(self, col: int) -> torch.Tensor:
        """partial maxima of the section that holds column `col` of this handle"""
        return self.sec_amax[(self.col0 + col) // self.sec_cols]

    def __repr__(self):
        return f"HeadImage(shape={tuple(self.shape)}, ld={self.ld}, col0={self.col0})"


def head_image_ok(x: torch.Tensor, w: torch.Tensor, n_head: int, sections: int) -> bool:
    """can the in-projection x @ w.T (+ b) leave as a head image for `ops.self_attention` / `cross_attention`?  64-column heads,
    reduction depth a multiple of 32, the fp16x3 forms selected, operands below 4 GiB."""
    N, K = w.shape
    M = x.numel() // K
    return (HEAD_IMAGES and x.is_cuda and _fwd_h3(K, N) and ATTN_FWD_MODE == "h3" and ATTN_BWD_MODE == "h3" and K % 32 == 0
            and N == sections * n_head * 64 and M * K * 4 < (1 << 32) and (M + 256) * N * 4 < (1 << 32))


def linear(x, w, b=None, residual=None, act=ACT_NONE, drop_p=0.0, seed=0, row_shift=0, T=0, sole_consumer=False,
           skip_in=None, skip_out=None, publish_amax=False, head_image_sections: int = 0):
    """`sole_consumer=True` is the caller's promise that nothing but this Linear reads `x`; if `x` came out of a
    relu(+dropout) Linear, its backward mask is then fused into this Linear's data-gradient epilogue.
    `skip_in` / `skip_out`: see SkipToken.
    `head_image_sections` = n > 0: the output leaves as a head image of n sections (q | k | v) and a `HeadImage` is returned.
    `publish_amax`: the output feeds another fp16x3 GEMM / attention kernel, so the epilogue leaves its partial maxima on
    it (`y._ttts_amax`) and that consumer needs no pass of its own over y.  True, or a zeroed AMAX_SLOTS-float array to add the
    maxima to (a running maximum over several calls: the K/V cache of `inference`)."""
    grad_on = torch.is_grad_enabled()
    N, K = w.shape
    h3 = x.is_cuda and _fwd_h3(K, N)
    x_full = _twin(x)
    twin = None
    if x_full is not None:
        r_full = _twin(residual)
        if residual is not None and r_full is None:
            raise ValueError("linear: the residual of a twin batch must be a twin batch too")
        twin = [x_full, r_full, []]
    # (a twin batch's maxima cover BOTH halves: its producer published them over the whole buffer)
    x_am = (_amax(x) if getattr(x, "_ttts_amax", None) is not None or x_full is None else _amax(x_full)) if h3 else None
    y_am = None
    if h3 and publish_amax is not False and publish_amax is not None:
        y_am = publish_amax if isinstance(publish_amax, torch.Tensor) else _amax_slots(x.device, True)
    tok_in = getattr(x, "_ttts_relu_token", None) if (sole_consumer and grad_on) else None
    tok_out = _ReluToken(1.0 / (1.0 - float(drop_p))) if act == ACT_RELU else None
    if not (grad_on and x.requires_grad):
        skip_in = None                      # nobody will pick the gradient up: leave it to autograd
    if skip_out is not None and (residual is None or not grad_on):
        skip_out = None
    tok_drop = None
    if grad_on and act == ACT_NONE and float(drop_p) > 0.0 and residual is not None and x.is_cuda:
        tok_drop = _DropToken(float(drop_p), seed, _ss())
    x_img = getattr(x, "_ttts_image", None) if (h3 and row_shift == 0) else None
    if x_img is not None and not _image_shape_ok(x.numel() // K, K, N):
        x_img = None
    y_himg = None
    if head_image_sections:
        # (rows of inverse scales [N / 64][M], one partial-maxima array per section, columns per section)
        if residual is not None or act != ACT_NONE or float(drop_p) > 0.0 or row_shift != 0 or N % (64 * head_image_sections) != 0:
            raise ValueError("linear: a head-image output takes a bias-only epilogue and whole 64-column heads per section")
        M = x.numel() // K * (2 if twin is not None else 1)
        sec = _amax_slots_n(x.device, head_image_sections)
        inv = _guarded(N // 64, M, x.device, "head-image inverse scales") if GUARD else \
            torch.empty(N // 64, M, dtype=torch.float32, device=x.device)
        y_himg = (inv, sec, N // head_image_sections)
        y_am = None
    if twin is not None:
        x_img = None
    y = LinearFn.apply(x, w, b, residual, act, drop_p, seed, row_shift, T, tok_out, tok_in, skip_in, skip_out, tok_drop,
                       x_am, y_am, x_img, y_himg, twin)
    if twin is not None:
        y._ttts_twin = twin[2][0]
    if y_himg is not None:
        return HeadImage(y, y_himg[0], y_himg[1], y_himg[2])
    if y_am is not None:
        y._ttts_amax = y_am
    if tok_drop is not None:
        y._ttts_drop_token = tok_drop           # picked up by layer_norm(y, ..., sole_consumer=True)
    if tok_out is not None:
        y._ttts_relu_token = tok_out
        if _relu_observer is not None:
            _relu_observer(y)
    return y


# ----------------------------------------------------------------------------------------------- fused cross-attention K/V
def _pack2(hi: int, lo: int) -> int:
    return (int(hi) << 32) | int(lo)


class _StackedPlanes:
    """ONE fp16x3 weight image of several parameter row slices (all (rows, cols)) STACKED: along the image's rows for the forward
    image (mode 8: the slices' output columns side by side, N = n * rows) or along its reduction index for the data-gradient
    image (mode 5: w^T of the slices one behind the other).  Each slice is an ordinary entry of the batched weight refresh
    (`_plane_entries`, `PlaneTable`) whose descriptor names its window of the shared image (include/ttts_hip.h,
    ttts_weight_split_batched); all windows share the image's tail, i.e. ONE scale from the maximum over every slice.  The
    entries are cached on the parameters the slices were cut from, so the refresh after an optimizer step and the table a
    captured graph replays cover them like any other weight."""

    def __init__(self, slices, mode: int):
        lib = _lib.load()
        self.mode, self.n = mode, len(slices)
        rows, cols = slices[0].shape
        self.rows, self.cols = rows, cols
        base = (mode - 4) & 3
        if base == 0:
            self.Rimg, self.Cimg = self.n * rows, cols          # forward image: (n rows, cols)
            wins = [(_pack2(self.Rimg, self.Cimg), _pack2(i * rows, 0), rows, cols) for i in range(self.n)]
        else:
            if rows % 32 != 0:
                raise ValueError("stacked data-gradient image: the slices' row counts must be multiples of 32")
            self.Rimg, self.Cimg = cols, self.n * rows          # image of w^T: (cols, n rows)
            wins = [(_pack2(self.Rimg, self.Cimg), _pack2(0, i * rows), cols, rows) for i in range(self.n)]
        nwords = (int(lib.ttts_split_image_bytes(self.Rimg, self.Cimg, mode, 0, 0)) + 1) // 2
        self.planes = torch.empty(nwords, dtype=torch.int16, device=slices[0].device)
        self.entries = []
        key = ("stack", mode, tuple(id(getattr(sl, "_ttts_planes_owner", (sl, 0))[0]) for sl in slices))
        for sl, (geo, off, R, C) in zip(slices, wins):
            holder, sub = getattr(sl, "_ttts_planes_owner", (sl, 0))
            cache = getattr(holder, "_ttts_planes", None)
            if cache is None:
                cache = {}
                holder._ttts_planes = cache
            ent = _PlaneEntry()
            ent.wref = weakref.ref(holder)
            ent.off = sl.data_ptr() - holder.data_ptr()
            ent.mode, ent.rows, ent.cols, ent.c2, ent.taps, ent.planes = mode, R, C, geo, off, self.planes
            ent.tag = (holder._version, sl.data_ptr(), -1)
            cache[(key, sub)] = ent
            _plane_entries.append(ent)
            self.entries.append((holder, ent))
        # its own small refresh table (first build, and whenever the process-wide / TrainStep refresh has not covered it)
        tab, blk = [], 0
        for holder, e in self.entries:
            tab.append([holder.data_ptr() + e.off, self.planes.data_ptr(), e.rows, e.cols, e.mode, e.c2, e.taps, blk])
            blk += _split_units(e.rows, e.cols, e.mode, 0)
        self.blocks = blk
        self.sig = [(holder.data_ptr() + e.off) for holder, e in self.entries]
        self.table = torch.tensor(tab, dtype=torch.int64).to(self.planes.device)

    def alive(self, slices) -> bool:
        return len(slices) == self.n and all(h.data_ptr() + e.off == sl.data_ptr() and e.planes is self.planes
                                             for (h, e), sl in zip(self.entries, slices))

    def current(self) -> bool:
        return all(e.tag == (h._version, h.data_ptr() + e.off, _param_epoch) for h, e in self.entries)

    def get(self) -> torch.Tensor:
        if not self.current():
            h0 = self.entries[0][0]
            only_epoch = all(e.tag[0] == h._version and e.tag[1] == h.data_ptr() + e.off for h, e in self.entries)
            if _BATCHED_SPLIT and only_epoch and not torch.cuda.is_current_stream_capturing():
                _refresh_all_planes(h0.untyped_storage().data_ptr())       # the optimizer stepped: this model's images at once
            if not self.current():
                _lib.check(_lib.load().ttts_weight_split_batched(_p(self.table), self.n, self.blocks, _stream()),
                           "ttts_weight_split_batched (stacked image)")
                for h, e in self.entries:
                    e.tag = (h._version, h.data_ptr() + e.off, _param_epoch)
        return self.planes


def _stacked_planes(owner, slices, mode: int) -> torch.Tensor:
    """the stacked image of `slices` in `mode`, cached on `owner` (the module that owns the fused op)"""
    cache = owner.__dict__.setdefault("_ttts_stacked", {})
    st = cache.get(mode)
    if st is None or not st.alive(slices):
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("stacked weight planes must exist before a HIP-graph capture: run the eager warm-up steps first")
        st = cache[mode] = _StackedPlanes(slices, mode)
    return st.get()


def _stacked_bias(owner, bs) -> torch.Tensor:
    """the slices' biases one behind the other (L 2d floats), gathered again only when a parameter changed: one 6 KB copy per
    optimizer step, recorded in the step's graph with the forward that first asks"""
    cache = owner.__dict__.setdefault("_ttts_stacked", {})
    # During a capture the gather must be RECORDED by whichever forward of the pass comes first, whatever an earlier capture pass
    # that was abandoned (TrainStep._capture_guarded's fallback) left in this host-side tag: the pass number is part of it.
    tag = tuple((b._version, b.data_ptr()) for b in bs) + (_param_epoch, _pass_id[0] if torch.cuda.is_current_stream_capturing() else -1)
    ent = cache.get("bias")
    if ent is None or ent[0].numel() != sum(b.numel() for b in bs) or ent[0].device != bs[0].device:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("stacked bias must exist before a HIP-graph capture: run the eager warm-up steps first")
        ent = cache["bias"] = [torch.empty(sum(b.numel() for b in bs), dtype=torch.float32, device=bs[0].device), None]
    if ent[1] != tag:
        torch.cat([b.detach() for b in bs], out=ent[0])
        ent[1] = tag
    return ent[0]


class CrossKVProjFn(torch.autograd.Function):
    """(k_l | v_l) = mem W_l[d:3d]^T + b_l[d:3d] for EVERY decoder layer l in one GEMM (N = L 2d), written as a head image: the
    reference projects the same encoder memory once per layer (model/layers.py:54-74 -> F.multi_head_attention_forward's
    in-projection of `key` / `value`, torch/nn/functional.py:6206+).  One launch instead of L per forward, and in backward ONE
    data-gradient GEMM with reduction depth L 2d (which also adds the layers' memory gradients: no fan-out, no add kernel) and
    ONE weight-gradient GEMM whose row blocks are reduced straight into each layer's gradient sink.
    Outputs: L windows (B, Tk, 2d) of the image (B, Tk, L 2d); their gradients arrive in `slab` (see _KVGradSlab)."""

    @staticmethod
    def forward(ctx, mem, owner, slab, row_inv, sec_amax, mem_amax, *wb):
        lib = _lib.load()
        mem = _chk(mem, "cross_kv_projection.memory")
        L = len(wb) // 2
        ws, bs = wb[:L], wb[L:]
        rows, K = ws[0].shape                     # 2d, d
        if mem.shape[-1] != K:
            raise ValueError("cross_kv_projection: memory width does not match the in-projections")
        B, Tk = mem.shape[0], mem.shape[1]
        M, N = B * Tk, L * rows
        y = torch.empty(B, Tk, N, dtype=torch.float32, device=mem.device)
        bias = _stacked_bias(owner, bs)
        if mem_amax is None:
            mem_amax = _amax(mem)
        planes = _stacked_planes(owner, ws, 8)
        _stacked_planes(owner, ws, 5)             # the backward's image exists before any capture
        _lib.check(lib.ttts_linear_fwd_h3d_img(_p(mem), _p(planes), _p(bias), _p(y), _p(row_inv), M, N, K, _p(mem_amax), _p(sec_amax),
                                               rows // 2, _stream()), "ttts_linear_fwd_h3d_img (stacked K/V)")
        ctx.save_for_backward(mem)
        ctx.params = (ws, bs)
        ctx.owner, ctx.slab, ctx.mem_amax = owner, slab, mem_amax
        ctx.sinks = _sinks(*ws, *bs)
        ctx.set_materialize_grads(False)
        return tuple(y[:, :, l * rows:(l + 1) * rows] for l in range(L))

    @staticmethod
    def backward(ctx, *grads):
        lib = _lib.load()
        (mem,) = ctx.saved_tensors
        ws, bs = ctx.params
        L = len(ws)
        rows, K = ws[0].shape
        B, Tk = mem.shape[0], mem.shape[1]
        M, N = B * Tk, L * rows
        dy, am, written = ctx.slab.take()
        if dy is None:
            if all(g is None for g in grads):
                return (None,) * (6 + 2 * L)
            dy, am, written = torch.empty(B, Tk, N, dtype=torch.float32, device=mem.device), None, set()
        for l, g in enumerate(grads):             # whatever did not arrive through the slab (a layer that ran another path)
            win = dy[:, :, l * rows:(l + 1) * rows]
            if g is None:
                if l * rows not in written:
                    win.zero_()
            elif g.data_ptr() != win.data_ptr() or g.stride() != win.stride():
                win.copy_(g)
                am = None
        if am is None:
            am = _amax(dy)
        dmem = None
        if ctx.needs_input_grad[0]:
            dmem = torch.empty_like(mem)
            _lib.check(lib.ttts_linear_bwd_data_h3(_p(dy), _p(_stacked_planes(ctx.owner, ws, 5)), None, _p(dmem), M, N, K, None, 1.0,
                                                   _p(am), None, _stream()), "ttts_linear_bwd_data_h3 (stacked K/V)")
        out_w = [None] * L
        out_b = [None] * L
        if any(ctx.needs_input_grad[6:6 + L]):
            import ctypes
            sk, acc, queue = ctx.sinks
            if sk is not None:
                dws, dbs = sk[:L], sk[L:]
            else:
                dws = [torch.empty(rows, K, dtype=torch.float32, device=mem.device) for _ in range(L)]
                dbs = [torch.empty(rows, dtype=torch.float32, device=mem.device) if bs[0] is not None else None for _ in range(L)]
                out_w, out_b = list(dws), list(dbs)
            nbytes = lib.ttts_wgrad_workspace_bytes(M, N, K, 1)
            wsp = _ws(nbytes, mem.device)
            PA = ctypes.c_void_p * L
            dwp = PA(*[t.data_ptr() for t in dws])
            dbp = PA(*[t.data_ptr() for t in dbs]) if dbs[0] is not None else None
            _lib.check(lib.ttts_linear_bwd_weight_h3_parts(_p(dy), _p(mem), dwp, dbp, L, _p(wsp), wsp.numel() * 4, M, N, K, acc, _p(am),
                                                           _p(ctx.mem_amax), _qarg(queue, wsp), _stream()),
                       "ttts_linear_bwd_weight_h3_parts")
        return (dmem, None, None, None, None, None, *out_w, *out_b)


def cross_kv_ok(mem: torch.Tensor, attns, n_head: int) -> bool:
    """can the K/V projections of `attns` (the decoder layers' MultiheadAttention modules) run as one head-image GEMM?"""
    if not (FUSED_CROSS_KV and HEAD_IMAGES and len(attns) > 1 and mem.is_cuda and WGRAD_MODE == "h3" and BWD_MODE == "h3"):
        return False
    w0 = attns[0].in_proj_weight
    d = w0.shape[1]
    same = all(a.in_proj_weight.shape == w0.shape and a.num_heads == n_head and (a.in_proj_bias is not None) for a in attns)
    return (same and d == n_head * 64 and d % 32 == 0 and _wgrad_is_split(len(attns) * 2 * d, d) and
            head_image_ok(mem, w0.detach()[d:], n_head, 2) and mem.numel() // d * (len(attns) * 2 * d + 256) * 4 < (1 << 32))


def cross_kv_projection(mem: torch.Tensor, attns, owner):
    """-> one HeadImage (k | v) per module of `attns` from ONE GEMM over `mem` (see CrossKVProjFn); `owner`: the module the
    stacked weight images are cached on (the decoder stack)."""
    L = len(attns)
    d = attns[0].in_proj_weight.shape[1]
    B, Tk = mem.shape[0], mem.shape[1]
    M, N = B * Tk, L * 2 * d
    ws = [param_rows(a.in_proj_weight, d, 3 * d) for a in attns]
    bs = [param_rows(a.in_proj_bias, d, 3 * d) for a in attns]
    sec = _amax_slots_n(mem.device, 2 * L)
    inv = _guarded(N // 64, M, mem.device, "stacked K/V inverse scales") if GUARD else torch.empty(N // 64, M, dtype=torch.float32, device=mem.device)
    slab = _KVGradSlab(B, Tk, N) if torch.is_grad_enabled() else None
    outs = CrossKVProjFn.apply(mem, owner, slab, inv, sec, _amax(mem), *ws, *bs)
    return [HeadImage(o, inv, sec, d, col0=l * 2 * d, slab=slab, index=l) for l, o in enumerate(outs)]


# ----------------------------------------------------------------------------------------------- heads
def _mel_head(lib, x, w_mel, b_mel, mel, M, N, K, x_amax, mel_amax):
    """mel = x w_mel^T + b_mel into the given buffer, in the configured form; -> the partial maxima of x it used (or None)"""
    if _fwd_h3(K, N):
        if x_amax is None:
            x_amax = _amax(x)
        _lib.check(lib.ttts_linear_fwd_h3(_p(x), _p(_planes(w_mel, 4, N, K)), _p(b_mel), None, _p(mel), M, N, K,
                                          ACT_NONE, 0.0, 0, None, 0, 0, _p(x_amax), _p(mel_amax), _stream()),
                   "ttts_linear_fwd_h3")
    elif GEMM_MODE == "x6":
        _lib.check(lib.ttts_linear_fwd_x6(_p(x), _p(_planes(w_mel, 0, N, K)), _p(b_mel), None, _p(mel), M, N, K,
                                          ACT_NONE, 0.0, 0, None, 0, 0, _stream()), "ttts_linear_fwd_x6")
    else:
        _lib.check(lib.ttts_linear_fwd(_p(x), _p(w_mel), _p(b_mel), None, _p(mel), M, N, K, ACT_NONE, 0.0,
                                       0, None, 0, 0, _stream()), "ttts_linear_fwd")
    return x_amax


class PostnetTwin:
    """The mel predictions of BOTH forwards of a training step as one twin batch (see `_twin`): the no-grad forward, which runs
    first, leaves its prediction in the SECOND half of `full` (2 B, T, n_mels) and does not run the post-net; the grad forward
    writes its own into the first half and runs the post-net once over both.  The reference runs the post-net in the no-grad
    forward too (model/model.py:310) and drops the result (lightning_module.py:53-59 keeps `pred_melspec`): what stays of it
    are the updates of its BatchNorm running statistics, which the twin pass makes per half, the no-grad forward's first.
    `amax`: partial maxima of |prediction| over both halves (each forward's mel head adds its own)."""
    __slots__ = ("full", "amax")

    def __init__(self):
        self.full = self.amax = None


TWIN_POSTNET = True


class HeadsFn(torch.autograd.Function):
    """mel = x @ w_mel.T + b_mel  (B,T,n_mels);  stop = x @ w_stop.T + b_stop  (B,T)  -- one read of dx."""

    @staticmethod
    def forward(ctx, x, w_mel, b_mel, w_stop, b_stop, x_amax=None, mel_amax=None, box=None):
        lib = _lib.load()
        x = _chk(x, "heads.x")
        N, K = w_mel.shape
        M = x.numel() // K
        # box: the mel prediction is written as the first half of the twin batch whose second half the no-grad forward left
        mel = box.full[:x.shape[0]] if box is not None else torch.empty(*x.shape[:-1], N, dtype=torch.float32, device=x.device)
        stop = torch.empty(x.shape[:-1], dtype=torch.float32, device=x.device)
        w_mel = _chk(w_mel, "w_mel")
        x_amax = _mel_head(lib, x, w_mel, b_mel, mel, M, N, K, x_amax, mel_amax)
        _lib.check(lib.ttts_rowdot_fwd(_p(x), _p(_chk(w_stop, "w_stop")), _p(b_stop), _p(stop), M, K, _stream()),
                   "ttts_rowdot_fwd")
        ctx.save_for_backward(x, w_mel, w_stop)
        ctx.sinks = _sinks(w_mel, b_mel, w_stop, b_stop)
        ctx.x_amax = x_amax
        return mel, stop

    @staticmethod
    def backward(ctx, dmel, dstop):
        lib = _lib.load()
        x, w_mel, w_stop = ctx.saved_tensors
        N, K = w_mel.shape
        M = x.numel() // K
        dmel = _chk(dmel, "heads.dmel")
        dstop = _chk(dstop, "heads.dstop")
        dx = torch.empty_like(x)
        dmel_am = None
        if _bwd_h3(N, K):
            dmel_am = _amax(dmel)                  # (shared with the weight gradient below)
            _lib.check(lib.ttts_linear_bwd_data_h3(_p(dmel), _p(_planes(w_mel, 5, K, N)), None, _p(dx), M, N, K, None, 1.0,
                                                   _p(dmel_am), None, _stream()), "ttts_linear_bwd_data_h3")
        elif GEMM_MODE == "x6":
            _lib.check(lib.ttts_linear_bwd_data_x6(_p(dmel), _p(_planes(w_mel, 1, K, N)), None, _p(dx), M, N, K, None, 1.0,
                                                   _stream()),
                       "ttts_linear_bwd_data_x6")
        else:
            _lib.check(lib.ttts_linear_bwd_data(_p(dmel), _p(w_mel), None, _p(dx), M, N, K, None, 1.0, _stream()),
                       "ttts_linear_bwd_data")
        ws = _ws(lib.ttts_wgrad_workspace_bytes(M, N, K, 1), x.device)
        sk, acc, queue = ctx.sinks
        ws2 = _ws(lib.ttts_rowdot_bwd_workspace_bytes(K), x.device)
        if sk is not None:
            t_wm, t_bm, t_ws, t_bs = sk
            dw_mel = db_mel = dw_stop = db_stop = None
        else:
            t_wm = dw_mel = torch.empty_like(w_mel)
            t_bm = db_mel = torch.empty(N, dtype=torch.float32, device=x.device)
            t_ws = dw_stop = torch.empty_like(w_stop)
            t_bs = db_stop = torch.empty(1, dtype=torch.float32, device=x.device)
        cls = lib.ttts_wgrad_group_ok(M, N, K, 1) if (WGRAD_GROUPS and DEFER_REDUCE and sk is not None and queue is not None and
                                                       WGRAD_MODE == "h3" and _wgrad_is_split(N, K)) else 0
        if cls:
            queue.defer_wgrad(cls, dmel, dmel_am if dmel_am is not None else _amax(dmel), x,
                              ctx.x_amax if ctx.x_amax is not None else _amax(x), t_wm, t_bm, M, N, K)
        else:
            _lib.check(_wgrad(lib, "ttts_linear_bwd_weight", dmel, dmel_am, x, ctx.x_amax, _wgrad_is_split(N, K), _qarg(queue, ws),
                              _p(t_wm), _p(t_bm), _p(ws), ws.numel() * 4, M, N, K, 0, 0, acc), "ttts_linear_bwd_weight")
        _lib.check(lib.ttts_rowdot_bwd(_p(dstop), _p(x), _p(w_stop), _p(dx), _p(t_ws), _p(t_bs), _p(ws2),
                                       ws2.numel() * 4, M, K, acc, _qarg(queue, ws2), _stream()), "ttts_rowdot_bwd")
        return dx, dw_mel, db_mel, dw_stop, db_stop, None, None, None


def heads(x, w_mel, b_mel, w_stop, b_stop, need_stop: bool = True, box: Optional[PostnetTwin] = None):
    """(mel, stop) heads; x's partial maxima ride on it (LayerNorm left them), and the mel output leaves with its own for
    the post-net's first convolution (its weight gradient reads pred_melspec as an fp16x3 operand).
    `need_stop=False` (no-grad callers only): the stop logits have no reader -- the first forward of training_step keeps
    `pred_melspec` alone (reference lightning_module.py:53-59) and the stop head has no state -- so they are not computed and
    None is returned for them."""
    N, K = w_mel.shape
    if box is not None and box.full is None:
        # the no-grad forward of a twin post-net: the prediction goes behind the place of the grad forward's
        if torch.is_grad_enabled() or need_stop or x.dim() != 3:
            raise ValueError("heads(box=<empty PostnetTwin>) is the no-grad forward's call (need_stop=False, (B, T, d) input)")
        lib = _lib.load()
        x = _chk(x, "heads.x")
        B = x.shape[0]
        box.full = torch.empty(2 * B, x.shape[1], N, dtype=torch.float32, device=x.device)
        box.amax = _amax_slots(x.device, True) if _fwd_h3(K, N) else None
        _mel_head(lib, x, _chk(w_mel, "w_mel"), b_mel, box.full[B:], x.numel() // K, N, K, None, box.amax)
        return box.full[B:], None
    if not need_stop:
        if torch.is_grad_enabled() and (x.requires_grad or w_stop.requires_grad):
            raise ValueError("heads(need_stop=False) is for no-grad forwards: the stop head's gradients would be lost")
        return linear(x, w_mel, b_mel, publish_amax=True), None
    h3 = x.is_cuda and _fwd_h3(K, N)
    x_am = _amax(x) if h3 else None
    if box is not None:
        if box.full.shape != (2 * x.shape[0], x.shape[1], N) or box.full.device != x.device:
            raise ValueError("heads: the twin post-net's buffer was left by a forward over another batch")
        mel_am = box.amax
    else:
        mel_am = _amax_slots(x.device, True) if h3 else None
    mel, stop = HeadsFn.apply(x, w_mel, b_mel, w_stop, b_stop, x_am, mel_am, box)
    if mel_am is not None:
        mel._ttts_amax = mel_am
    if box is not None:
        mel._ttts_twin = box.full
    return mel, stop


# ----------------------------------------------------------------------------------------------- conv + BN
class ConvBNFn(torch.autograd.Function):
    """z = drop(act(BatchNorm1d(Conv1d_same(x)))) on (B,T,C); running stats updated in place when training."""

    @staticmethod
    def forward(ctx, x, conv_w, conv_b, gamma, beta, running_mean, running_var, nbt, training, momentum, eps, act,
                drop_p, seed, x_amax=None, z_amax=None, twin=None):
        lib = _lib.load()
        x = _chk(x, "conv_bn.x")
        B, T, cin = x.shape
        cout, cin_w, taps = conv_w.shape
        if cin != cin_w:
            raise ValueError(f"conv_bn: x has {cin} channels, weight expects {cin_w}")
        dev = x.device
        conv_w = _chk(conv_w, "conv.weight")
        if twin is not None:
            return ConvBNFn._forward_twin(ctx, lib, x, conv_w, conv_b, gamma, beta, running_mean, running_var, nbt, training, momentum,
                                          eps, act, drop_p, seed, x_amax, z_amax, twin)
        y = torch.empty(B, T, cout, dtype=torch.float32, device=dev)
        M = B * T
        bn_nblk, bn_ws = 0, None
        if _fwd_h3(taps * cin, cout, cin):
            if x_amax is None:
                x_amax = _amax(x)
            if training and M > 1:
                # the convolution's epilogue leaves BatchNorm's row-chunk partials behind: no statistics pass over y
                bn_nblk = lib.ttts_conv1d_fwd_h3_bn_blocks(B, T, cin, cout, taps)
                if bn_nblk > 0:
                    bn_ws = _ws(lib.ttts_bn_workspace_bytes(M, cout), dev)
            _lib.check(lib.ttts_conv1d_fwd_h3(_p(x), _p(_planes(conv_w, 6, cout, taps * cin, cin, taps)), _p(conv_b), _p(y),
                                              B, T, cin, cout, taps, _p(x_amax), _p(bn_ws), _stream()), "ttts_conv1d_fwd_h3")
        elif GEMM_MODE == "x6":
            _lib.check(lib.ttts_conv1d_fwd_x6(_p(x), _p(_planes(conv_w, 2, cout, taps * cin, cin, taps)), _p(conv_b), _p(y),
                                              B, T, cin, cout, taps, _stream()), "ttts_conv1d_fwd_x6")
        else:
            w_fwd = torch.empty(cout * taps * cin, dtype=torch.float32, device=dev)
            _lib.check(lib.ttts_conv1d_pack_weight(_p(conv_w), _p(w_fwd), None, cout, cin, taps, _stream()),
                       "ttts_conv1d_pack_weight")
            _lib.check(lib.ttts_conv1d_fwd(_p(x), _p(w_fwd), _p(conv_b), _p(y), B, T, cin, cout, taps, _stream()),
                       "ttts_conv1d_fwd")
        mean = torch.empty(cout, dtype=torch.float32, device=dev)
        invstd = torch.empty(cout, dtype=torch.float32, device=dev)
        if training and bn_ws is not None:
            _lib.check(lib.ttts_bn_train_stats_from_partials(_p(bn_ws), bn_nblk, _p(mean), _p(invstd), _p(running_mean),
                                                             _p(running_var), _p(nbt), cout, float(momentum), float(eps),
                                                             _stream()), "ttts_bn_train_stats_from_partials")
        elif training:
            ws = _ws(lib.ttts_bn_workspace_bytes(M, cout), dev)
            _lib.check(lib.ttts_bn_train_stats(_p(y), _p(mean), _p(invstd), _p(running_mean), _p(running_var), _p(nbt),
                                               _p(ws), ws.numel() * 4, M, cout, float(momentum), float(eps), _stream()),
                       "ttts_bn_train_stats")
        else:
            _lib.check(lib.ttts_bn_eval_stats(_p(running_mean), _p(running_var), _p(mean), _p(invstd), cout, float(eps),
                                              _stream()), "ttts_bn_eval_stats")
        z = torch.empty_like(y)
        _lib.check(lib.ttts_bn_apply_fwd(_p(y), _p(mean), _p(invstd), _p(gamma), _p(beta), _p(z), M, cout, act,
                                         float(drop_p), seed, _ss(), _p(z_amax), _stream()), "ttts_bn_apply_fwd")
        ctx.save_for_backward(x, conv_w, y, mean, invstd, gamma, beta)
        ctx.x_amax = x_amax
        ctx.cfg = (training, act, float(drop_p), seed, conv_b is not None)
        ctx.ss = _ss()
        ctx.sinks = _sinks(conv_w, conv_b, gamma, beta)
        return z

    @staticmethod
    def _forward_twin(ctx, lib, x, conv_w, conv_b, gamma, beta, running_mean, running_var, nbt, training, momentum, eps, act, drop_p,
                      seed, x_amax, z_amax, twin):
        """x is the first half of a twin batch (`_twin`): ONE convolution over all 2 B utterances; BatchNorm stays per forward --
        statistics of each half from its own row chunks of the epilogue's partials (or its own pass), the no-grad forward's
        first (its update of the running statistics precedes the grad forward's, as the reference's two forwards order them),
        normalisation + activation + dropout per half with a mask stream of its own."""
        B, T, cin = x.shape
        cout, _, taps = conv_w.shape
        dev = x.device
        if not _fwd_h3(taps * cin, cout, cin):
            raise ValueError("conv_bn: a twin batch runs on the fp16x3 form only")
        Bk, M = 2 * B, B * T
        y_k = torch.empty(Bk, T, cout, dtype=torch.float32, device=dev)
        z_k = torch.empty(Bk, T, cout, dtype=torch.float32, device=dev)
        if x_amax is None:
            x_amax = _amax(twin[0])
        bn_ws, runs = None, None
        if training and M > 1:
            nblk = lib.ttts_conv1d_fwd_h3_bn_blocks(Bk, T, cin, cout, taps)
            chunk = lib.ttts_conv1d_fwd_h3_bn_chunk_rows(Bk, T, cin, cout, taps)
            if nblk > 0 and 0 < chunk <= 256 and nblk >= -(-2 * M // chunk):      # (chunks past the last row are (0, 0, 0))
                # each half takes its whole row chunks from the epilogue's partials; the chunk the halves share (M is no multiple
                # of the chunk) is read from y itself, row by row: (first partial, partials, first row, rows) per half
                bn_ws = _ws(lib.ttts_bn_workspace_bytes(2 * M, cout), dev)
                s_, cut = divmod(M, chunk)
                if cut == 0:
                    runs = {0: (0, s_, 0, 0), 1: (s_, nblk - s_, 0, 0)}
                else:
                    runs = {0: (0, s_, s_ * chunk, cut), 1: (s_ + 1, nblk - s_ - 1, M, min((s_ + 1) * chunk, 2 * M) - M)}
        _lib.check(lib.ttts_conv1d_fwd_h3(_p(x), _p(_planes(conv_w, 6, cout, taps * cin, cin, taps)), _p(conv_b), _p(y_k),
                                          Bk, T, cin, cout, taps, _p(x_amax), _p(bn_ws), _stream()), "ttts_conv1d_fwd_h3")
        mi = torch.empty(2, 2, cout, dtype=torch.float32, device=dev)          # [half][mean, invstd]
        halves = (1, 0)                           # the no-grad forward's half first
        if training and bn_ws is not None:        # both halves' statistics in one launch (running statistics: first set first)
            sets = []
            for h in halves:
                b0, nb, r0, nr = runs[h]
                sets += [_off(bn_ws, b0 * 3 * cout) if nb else None, nb, _off(y_k, r0 * cout) if nr else None, nr, _p(mi[h, 0]), _p(mi[h, 1])]
            _lib.check(lib.ttts_bn_train_stats_twin(*sets, _p(running_mean), _p(running_var), _p(nbt), cout, float(momentum),
                                                    float(eps), _stream()), "ttts_bn_train_stats_twin")
        for h in halves:
            y_h, z_h = y_k[h * B:(h + 1) * B], z_k[h * B:(h + 1) * B]
            mean, invstd = mi[h, 0], mi[h, 1]
            if training and bn_ws is None:        # (a tile that leaves no partials: a statistics pass per half)
                ws = _ws(lib.ttts_bn_workspace_bytes(M, cout), dev)
                _lib.check(lib.ttts_bn_train_stats(_p(y_h), _p(mean), _p(invstd), _p(running_mean), _p(running_var), _p(nbt),
                                                   _p(ws), ws.numel() * 4, M, cout, float(momentum), float(eps), _stream()),
                           "ttts_bn_train_stats")
            elif not training:
                _lib.check(lib.ttts_bn_eval_stats(_p(running_mean), _p(running_var), _p(mean), _p(invstd), cout, float(eps),
                                                  _stream()), "ttts_bn_eval_stats")
            if h == 1 and len(twin) > 2 and twin[2]:
                continue                      # the no-grad half of the LAST layer of a twin pass: its statistics were all anybody wanted
            # (each half is a launch of its own whose element indices start at zero: the no-grad half draws from another seed)
            seed_h = seed if (h == 0 or seed == 0) else ((seed * 0x9E3779B97F4A7C15 + 0x632BE59BD9B4E019) & 0xFFFFFFFFFFFFFFFF)
            _lib.check(lib.ttts_bn_apply_fwd(_p(y_h), _p(mean), _p(invstd), _p(gamma), _p(beta), _p(z_h), M, cout, act,
                                             float(drop_p), seed_h, _ss(), _p(z_amax), _stream()), "ttts_bn_apply_fwd")
        twin[1].append(z_k)
        ctx.save_for_backward(x, conv_w, y_k[:B], mi[0, 0], mi[0, 1], gamma, beta)
        ctx.x_amax = x_amax
        ctx.cfg = (training, act, float(drop_p), seed, conv_b is not None)
        ctx.ss = _ss()
        ctx.sinks = _sinks(conv_w, conv_b, gamma, beta)
        return z_k[:B]

    @staticmethod
    def backward(ctx, dz):
        lib = _lib.load()
        x, conv_w, y, mean, invstd, gamma, beta = ctx.saved_tensors
        training, act, drop_p, seed, has_b = ctx.cfg
        B, T, cin = x.shape
        cout, _, taps = conv_w.shape
        dev = x.device
        M = B * T
        dz = _chk(dz, "conv_bn.dz")
        dy = torch.empty_like(y)
        sk, acc, queue = ctx.sinks
        dw = db = dgamma = dbeta = None
        if sk is not None:
            t_w, t_b, t_g, t_be = sk
        else:
            t_w = dw = torch.empty_like(conv_w)
            t_b = db = torch.empty(cout, dtype=torch.float32, device=dev) if has_b else None
            t_g = dgamma = torch.empty_like(gamma)
            t_be = dbeta = torch.empty_like(beta)
        ws = _ws(lib.ttts_bn_workspace_bytes(M, cout), dev)
        # the BatchNorm backward writes dy, the gradient both conv GEMMs below consume: it leaves dy's partial maxima too
        want_am = (ctx.needs_input_grad[0] and _bwd_h3(taps * cout, cin, cout)) or \
                  (WGRAD_MODE == "h3" and _wgrad_is_split(cout, cin))
        am = _amax_slots(dev, True) if want_am else None
        _lib.check(lib.ttts_bn_bwd(_p(dz), _p(y), _p(mean), _p(invstd), _p(gamma), _p(beta), _p(dy), _p(t_g), _p(t_be),
                                   _p(ws), ws.numel() * 4, M, cout, act, drop_p, seed, ctx.ss, acc, _p(am), 1 if training else 0,
                                   _stream()), "ttts_bn_bwd")
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            if _bwd_h3(taps * cout, cin, cout):
                _lib.check(lib.ttts_conv1d_bwd_data_h3(_p(dy), _p(_planes(conv_w, 7, cin, taps * cout, cout, taps)), _p(dx),
                                                       B, T, cin, cout, taps, _p(am), _stream()), "ttts_conv1d_bwd_data_h3")
            elif GEMM_MODE == "x6":
                _lib.check(lib.ttts_conv1d_bwd_data_x6(_p(dy), _p(_planes(conv_w, 3, cin, taps * cout, cout, taps)), _p(dx),
                                                       B, T, cin, cout, taps, _stream()), "ttts_conv1d_bwd_data_x6")
            else:
                w_bwd = torch.empty(cin * taps * cout, dtype=torch.float32, device=dev)
                _lib.check(lib.ttts_conv1d_pack_weight(_p(conv_w), None, _p(w_bwd), cout, cin, taps, _stream()),
                           "ttts_conv1d_pack_weight")
                _lib.check(lib.ttts_conv1d_bwd_data(_p(dy), _p(w_bwd), _p(dx), B, T, cin, cout, taps, _stream()),
                           "ttts_conv1d_bwd_data")
        cls = lib.ttts_wgrad_group_ok(M, cout, cin, taps) if (WGRAD_GROUPS and DEFER_REDUCE and sk is not None and queue is not None and
                                                              WGRAD_MODE == "h3" and _wgrad_is_split(cout, cin)) else 0
        if cls:       # a gradient sink: nobody reads the result before the optimizer, so it waits for its group
            queue.defer_wgrad(cls, dy, am if am is not None else _amax(dy), x, ctx.x_amax if ctx.x_amax is not None else _amax(x),
                              t_w, t_b, M, cout, cin, taps, T)
        else:
            ws2 = _ws(lib.ttts_wgrad_workspace_bytes(M, cout, cin, taps), dev)
            _lib.check(_wgrad(lib, "ttts_conv1d_bwd_weight", dy, am, x, ctx.x_amax, _wgrad_is_split(cout, cin),
                              _qarg(queue, ws2) if sk is not None else None, _p(t_w), _p(t_b), _p(ws2), ws2.numel() * 4, B, T, cin,
                              cout, taps, acc), "ttts_conv1d_bwd_weight")
        return dx, dw, db, dgamma, dbeta, None, None, None, None, None, None, None, None, None, None, None, None


def conv_bn(x, conv_w, conv_b, gamma, beta, running_mean, running_var, nbt, training, momentum=0.1, eps=1e-5,
            act=ACT_NONE, drop_p=0.0, seed=0, publish_amax=True, twin_last=False):
    """`publish_amax`: leave the partial maxima of the output on it (`z._ttts_amax`) for the fp16x3 GEMM that reads it.
    `twin_last`: if x is the first half of a twin batch (`_twin`), nobody reads the other half's output: the no-grad forward's
    rows still enter the convolution and update the running statistics, but are not normalised, and the result is an ordinary
    tensor of B utterances."""
    cout, cin, taps = conv_w.shape
    x_full = _twin(x)
    twin = [x_full, [], bool(twin_last)] if x_full is not None else None
    x_am = None
    if x.is_cuda and _fwd_h3(taps * cin, cout, cin):
        x_am = _amax(x) if (getattr(x, "_ttts_amax", None) is not None or x_full is None) else _amax(x_full)
    z_am = _amax_slots(x.device, True) if (publish_amax and x.is_cuda) else None
    z = ConvBNFn.apply(x, conv_w, conv_b, gamma, beta, running_mean, running_var, nbt, training, momentum, eps, act,
                       drop_p, seed, x_am, z_am, twin)
    if twin is not None and not twin_last:
        z._ttts_twin = twin[1][0]
    if z_am is not None:
        z._ttts_amax = z_am
    return z


# ----------------------------------------------------------------------------------------------- layer norm
class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps, tok_drop=None, y_amax=None, y_image=None, bwd_image=False, twin=None):
        """y_image: None, or (image, row_inv) buffers the kernel fills with the image operand of y; bwd_image: the backward leaves
        the image of the gradient it hands to the Linear that produced x (on the drop token, or on dx)."""
        lib = _lib.load()
        ctx.tok_drop = tok_drop
        ctx.bwd_image = bool(bwd_image)
        x = _chk(x, "layernorm.x")
        d = x.shape[-1]
        M = x.numel() // d
        Mk = 2 * M if twin is not None else M              # (twin batch, see `_twin`: the kernel normalises both halves)
        y_k = torch.empty(2 * x.shape[0], *x.shape[1:], dtype=torch.float32, device=x.device) if twin is not None else torch.empty_like(x)
        mean = torch.empty(Mk, dtype=torch.float32, device=x.device)
        rstd = torch.empty(Mk, dtype=torch.float32, device=x.device)
        yi, yv = y_image if (y_image is not None and twin is None) else (None, None)
        _lib.check(lib.ttts_layernorm_fwd(_p(x), _p(gamma), _p(beta), _p(y_k), _p(mean), _p(rstd), Mk, d, float(eps),
                                          _p(y_amax), _p(yi), _p(yv), _stream()), "ttts_layernorm_fwd")
        y = y_k
        if twin is not None:
            twin[1].append(y_k)
            y, mean, rstd = y_k[:x.shape[0]], mean[:M], rstd[:M]
        ctx.save_for_backward(x, gamma, mean, rstd)
        ctx.sinks = _sinks(gamma, beta)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, gamma, mean, rstd = ctx.saved_tensors
        d = x.shape[-1]
        M = x.numel() // d
        dy = _chk(dy, "layernorm.dy")
        dx = torch.empty_like(x)
        sk, acc, queue = ctx.sinks
        dgamma = dbeta = None
        if sk is not None:
            t_g, t_b = sk
        else:
            t_g = dgamma = torch.empty_like(gamma)
            t_b = dbeta = torch.empty_like(gamma)
        ws = _ws(lib.ttts_layernorm_bwd_workspace_bytes(d), x.device)
        qa = _qarg(queue, ws) if sk is not None else None
        td = ctx.tok_drop
        img = _new_image(M, d, x.device) if (ctx.bwd_image and _image_rows_ok(M, d)) else (None, None)
        if td is not None and d in (256, 512, 1024):
            # x is the output of a Linear with residual dropout and feeds nothing but this LayerNorm: dx is that Linear's dy
            dacc = torch.empty_like(x)
            am = _amax_slots(x.device, True)
            _lib.check(lib.ttts_layernorm_bwd_drop(_p(dy), _p(x), _p(mean), _p(rstd), _p(gamma), _p(dx), _p(t_g), _p(t_b),
                                                   _p(ws), ws.numel() * 4, M, d, acc, _p(dacc), td.p, td.seed, td.ss, _p(am),
                                                   _p(img[0]), _p(img[1]), qa, _stream()), "ttts_layernorm_bwd_drop")
            td.dx, td.dacc, td.amax = dx, dacc, am
            td.image = img if img[0] is not None else None
        else:
            _lib.check(lib.ttts_layernorm_bwd(_p(dy), _p(x), _p(mean), _p(rstd), _p(gamma), _p(dx), _p(t_g), _p(t_b),
                                              _p(ws), ws.numel() * 4, M, d, acc, _p(img[0]), _p(img[1]), qa, _stream()),
                       "ttts_layernorm_bwd")
            if img[0] is not None:
                dx._ttts_image = img
        return dx, dgamma, dbeta, None, None, None, None, None, None


def layer_norm(x, gamma, beta, eps=1e-5, sole_consumer=False, publish_amax=True, emit_image=None):
    """`sole_consumer=True` is the caller's promise that nothing but this LayerNorm reads `x`; if `x` came out of a Linear
    with a residual-dropout epilogue, that Linear's dropout backward is then written by this LayerNorm's backward kernel.
    `publish_amax`: leave the partial maxima of the output on it for the fp16x3 GEMMs that read it."""
    tok = getattr(x, "_ttts_drop_token", None) if (sole_consumer and torch.is_grad_enabled()) else None
    y_am = _amax_slots(x.device, True) if (publish_amax and x.is_cuda) else None
    d = x.shape[-1]
    M = x.numel() // d
    # the image operand of y for the GEMMs that read it (in-projections, FFN1), and -- when x is the output of a Linear nobody
    # else reads -- of the gradient that Linear's backward consumes
    if emit_image is None:
        emit_image = LAYERNORM_IMAGES
    x_full = _twin(x)
    twin = [x_full, []] if x_full is not None else None
    y_img = _new_image(M, d, x.device) if (emit_image and x.is_cuda and _image_rows_ok(M, d) and twin is None) else None
    y = LayerNormFn.apply(x, gamma, beta, eps, tok, y_am, y_img, bool(emit_image and sole_consumer and twin is None), twin)
    if twin is not None:
        y._ttts_twin = twin[1][0]
    if y_am is not None:
        y._ttts_amax = y_am
    if y_img is not None:
        y._ttts_image = y_img
    return y


# ----------------------------------------------------------------------------------------------- attention
def _attn_fwd(q, k, v, ldq, ldk, ldv, B, H, Tq, Tk, lens, causal, drop_p, seed, need_weights, q_am=None, k_am=None,
              v_am=None, o_am=None, q_scale: float = 0.125):
    """q_am / k_am / v_am: partial maxima of the operands (fp16x3 form; required there); o_am: None, or a zeroed
    AMAX_SLOTS-slot array that receives max|o|."""
    lib = _lib.load()
    dev = lens.device
    o = torch.empty(B, Tq, H * 64, dtype=torch.float32, device=dev)
    # lse (natural units) and, in the fp16x3 form, the row statistics in the kernel's own units behind it: rows 1-3 of `stat`
    stat = torch.empty(4 if ATTN_FWD_MODE == "h3" else 1, B, H, Tq, dtype=torch.float32, device=dev)
    lse = stat[0]
    attn = torch.empty(B, H, Tq, Tk, dtype=torch.float32, device=dev) if need_weights else None
    args = (q, k, v, _p(o), _p(lse), _p(attn), _p(lens), B, H, Tq, Tk, ldq, ldk, ldv, H * 64, 1 if causal else 0,
            float(q_scale), float(drop_p), seed, _ss())
    if ATTN_FWD_MODE == "h3":
        if q_am is None or k_am is None or v_am is None:
            raise ValueError("attention (fp16x3 form): the partial maxima of q, k and v are required")
        _lib.check(lib.ttts_attention_fwd_h3(*args, _p(q_am), _p(k_am), _p(v_am), _p(o_am), _p(stat[1:]), _stream()),
                   "ttts_attention_fwd_h3")
    else:
        fwd = lib.ttts_attention_fwd_x6 if ATTN_FWD_MODE == "x6" else lib.ttts_attention_fwd
        _lib.check(fwd(*args, _stream()), "ttts_attention_fwd")
    return o, stat, attn


def _attn_h3() -> bool:
    return ATTN_FWD_MODE == "h3" or ATTN_BWD_MODE == "h3"


def _off(t: torch.Tensor, col: int):
    return c_void_p(t.data_ptr() + 4 * col)


def _head_dim(d: int, n_head: int) -> int:
    """Columns per head; the kernels work on 64-column heads and narrower ones are zero-padded to 64 (`_pad_heads`); wider
    heads do not reach them (`_attention_wide_heads`)."""
    if n_head <= 0 or d % n_head != 0:
        raise ValueError(f"attention: d_model {d} is not divisible by {n_head} heads")
    hd = d // n_head
    if hd > 64:
        raise ValueError(f"attention kernels take head_dim <= 64 (d_model {d}, heads {n_head}: head_dim {hd})")
    return hd


def _wide_heads(d: int, n_head: int) -> bool:
    if n_head <= 0 or d % n_head != 0:
        raise ValueError(f"attention: d_model {d} is not divisible by {n_head} heads")
    return d // n_head > 64


def _attention_wide_heads(q, k, v, lens, n_head: int, causal: bool, drop_p: float, dead=None, add_mask=None):
    """Attention as plain tensor algebra on the library's fp32 GEMMs, for the two cases the hand-written kernels do not take:
      * heads wider than 64 columns (the reference takes any `nhead`, model/model.py:139-161; no BASELINE configuration has them):
        the kernels hold a 64-column head per fragment set;
      * masks that are not "keys past a length" / causal (`masked_attention`: a key-padding mask with holes, `memory_mask`, an
        arbitrary `tgt_mask` / `mask` -- arguments of the reference's layers, model/layers.py:29-74, that its model never passes).
    Same conventions as the kernels (weights returned AFTER dropout, rows without an allowed key give zeros), differentiated by
    autograd.  `dead` (B, Tk) bool replaces the keys-past-`lens` mask; `add_mask` (broadcastable to (B, H, Tq, Tk), finite) is
    added to the scaled scores as torch adds a float `attn_mask`.  Correct, not tuned."""
    B, Tq, d = q.shape
    Tk = k.shape[1]
    hd = d // n_head
    qh = q.reshape(B, Tq, n_head, hd).transpose(1, 2)
    kh = k.reshape(B, Tk, n_head, hd).transpose(1, 2)
    vh = v.reshape(B, Tk, n_head, hd).transpose(1, 2)
    s = torch.matmul(qh * hd ** -0.5, kh.transpose(-1, -2))                      # (B, H, Tq, Tk)
    key = torch.arange(Tk, device=q.device)
    if dead is None:
        dead = key[None, :] >= lens.to(q.device)[:, None]                         # (B, Tk)
    big = torch.finfo(torch.float32).min
    if add_mask is not None:
        s = s + add_mask
    s = s.masked_fill(dead[:, None, None, :], big)                                # finite: no NaN in either direction
    if causal:
        s = s.masked_fill((key[None, :] > torch.arange(Tq, device=q.device)[:, None])[None, None], big)
    # a row without an allowed key (an utterance without keys, a padded query whose band holds dead keys only): zeros, as the
    # kernels and torch's scaled_dot_product_attention give -- not the uniform row a finite fill would leave
    p = torch.softmax(s, dim=-1) * (s > 0.5 * big).any(dim=-1, keepdim=True).to(s.dtype)
    if drop_p > 0:
        p = torch.nn.functional.dropout(p, drop_p, True)
    o = torch.matmul(p, vh).transpose(1, 2).reshape(B, Tq, d)
    return o, p


def masked_attention(q, k, v, lens, n_head: int, causal: bool, drop_p: float, dead=None, add_mask=None):
    """(context, per-head weights) under masks the kernels do not derive from lengths: see `_attention_wide_heads`."""
    if n_head <= 0 or q.shape[-1] % n_head != 0:
        raise ValueError(f"attention: d_model {q.shape[-1]} is not divisible by {n_head} heads")
    return _attention_wide_heads(q, k, v, lens, n_head, causal, drop_p, dead, add_mask)


def _pad_heads(src: torch.Tensor, col0: int, ld: int, rows: int, H: int, hd: int) -> torch.Tensor:
    """(rows, H*64): the H heads of `src` (row stride ld floats, head h at column col0 + h*hd) zero-padded to 64 columns."""
    dst = torch.empty(rows, H * 64, dtype=torch.float32, device=src.device)
    _lib.check(_lib.load().ttts_heads_pad(_off(src, col0), ld, _p(dst), rows, H, hd, _stream()), "ttts_heads_pad")
    return dst


def _unpad_heads(src: torch.Tensor, dst: torch.Tensor, col0: int, ld: int, rows: int, H: int, hd: int) -> None:
    _lib.check(_lib.load().ttts_heads_unpad(_p(src), _off(dst, col0), ld, rows, H, hd, _stream()), "ttts_heads_unpad")


class SelfAttentionImgFn(torch.autograd.Function):
    """o = softmax(mask(q k^T / 8)) v on a packed in-projection output that arrived as a HEAD IMAGE (`linear(...,
    head_image_sections=3)`): K / V tiles are staged by LDS-DMA, no split arithmetic (csrc/attention_img.hip).
    `twin` = [image buffer, lengths buffer, out list]: qkv / lens are the first halves of a twin batch (see `_twin`): the forward
    runs on all 2 B utterances, the backward on the first B (the inverse scales' head planes and the row statistics' planes keep the
    buffer's strides: q_inv_rows / k_inv_rows / stat_plane of the C ABI)."""

    @staticmethod
    def forward(ctx, qkv, row_inv, sec_amax, lens, n_head, causal, drop_p, seed, o_amax=None, twin=None):
        lib = _lib.load()
        qkv = _chk(qkv, "self_attention.qkv")
        lens = _chk(lens, "self_attention.lens", torch.int64)
        B, T, d3 = qkv.shape
        Bk = 2 * B if twin is not None else B             # utterances the kernel sees
        d, M = d3 // 3, Bk * T
        if row_inv.shape[1] != M or (twin is not None and (twin[1] is None or twin[1].shape[0] != Bk)):
            raise ValueError("self_attention: inverse scales / lengths do not match the image")
        o_k = torch.empty(Bk, T, d, dtype=torch.float32, device=qkv.device)
        stat = torch.empty(6, Bk, n_head, T, dtype=torch.float32, device=qkv.device)     # lse, then the five row-statistic planes
        HM = n_head * M
        _lib.check(lib.ttts_attention_fwd_img(_off(qkv, 0), _off(qkv, d), _off(qkv, 2 * d), _off(row_inv, 0), _off(row_inv, HM),
                                              _off(row_inv, 2 * HM), _p(o_k), _p(stat[0]), None, _p(lens), Bk, n_head, T, T, d3, d3, d3, d,
                                              1 if causal else 0, 0.125, float(drop_p), seed, _ss(), _p(sec_amax[2]), _p(o_amax),
                                              _p(stat[1:]), 0, 0, 0, _stream()), "ttts_attention_fwd_img")
        o = o_k[:B] if twin is not None else o_k
        if twin is not None:
            twin[2].append(o_k)
        ctx.save_for_backward(qkv, row_inv, o, stat, lens)
        ctx.cfg = (n_head, causal, float(drop_p), seed, Bk)
        ctx.ss = _ss()
        return o

    @staticmethod
    def backward(ctx, do):
        lib = _lib.load()
        qkv, row_inv, o, stat, lens = ctx.saved_tensors
        n_head, causal, drop_p, seed, Bk = ctx.cfg
        B, T, d3 = qkv.shape
        d, Mk = d3 // 3, Bk * T
        HM = n_head * Mk
        do = _chk(do, "self_attention.do")
        dqkv = torch.empty_like(qkv)
        delta = torch.empty(B, n_head, T, dtype=torch.float32, device=qkv.device)
        am = _amax_slots(qkv.device, True)                                     # max|dqkv| for the in-projection gradients
        _lib.check(lib.ttts_attention_bwd_img(_off(qkv, 0), _off(qkv, d), _off(qkv, 2 * d), _off(row_inv, 0), _off(row_inv, HM),
                                              _off(row_inv, 2 * HM), _p(o), _p(do), _p(stat[1:]), _p(delta), _off(dqkv, 0), _off(dqkv, d),
                                              _off(dqkv, 2 * d), _p(lens), B, n_head, T, T, d3, d3, d3, d, d3, d3, d3, 1 if causal else 0,
                                              0.125, drop_p, seed, ctx.ss, _p(_amax(do)), _p(am), _p(am), None, 1,
                                              Mk, Mk, Bk * n_head * T, _stream()), "ttts_attention_bwd_img")
        dqkv._ttts_amax = am
        return dqkv, None, None, None, None, None, None, None, None, None


def _dkv_query_splits(key_blocks: int, Tq: int) -> int:
    """query-range splits of the dK / dV kernel: enough workgroups for two per CU (512), each with at least four 32-query stages"""
    if key_blocks >= 512 or Tq < 256:
        return 1
    return max(1, min(8, 512 // key_blocks, (Tq // 32) // 4))


def _chk_window(t: torch.Tensor, name: str) -> torch.Tensor:
    """a (B, T, w) fp32 CUDA tensor whose rows are `ld` cells apart (ld >= w: contiguous, or a column window of a wider
    contiguous tensor) -- never copied: the cells of a head image mean nothing to a copy kernel's consumer but are legal to move,
    yet a window must stay inside the tensor its inverse scales describe"""
    if not t.is_cuda or t.dtype != torch.float32 or t.dim() != 3:
        raise ValueError(f"{name}: expected a (B, T, w) fp32 CUDA tensor")
    B, T, w = t.shape
    if t.stride(2) != 1 or t.stride(1) < w or t.stride(0) != T * t.stride(1) or t.stride(1) % 4 != 0 or t.data_ptr() % 16 != 0:
        raise ValueError(f"{name}: rows must be equally spaced, 16-byte aligned windows of one tensor (strides {t.stride()})")
    return t


class _KVGradSlab:
    """Where the gradients of the L windows of a stacked K/V projection meet: ONE (B, Tk, L 2d) buffer allocated by the first
    cross-attention backward of a pass and ONE partial-maxima array, both consumed by CrossKVProjFn.backward.  Each layer's
    dK / dV kernel writes its window in place (row stride L 2d), so no copy, no zero-fill and no accumulation kernel runs."""
    __slots__ = ("shape", "buf", "amax", "written")

    def __init__(self, B: int, Tk: int, width: int):
        self.shape = (B, Tk, width)
        self.buf = self.amax = None
        self.written = set()

    def window(self, device, col0: int, w: int) -> torch.Tensor:
        if self.buf is None:
            self.buf = torch.empty(self.shape, dtype=torch.float32, device=device)
            self.amax = _amax_slots(device, True)
            self.written = set()
        self.written.add(col0)
        return self.buf[:, :, col0:col0 + w]

    def take(self):
        buf, am, wr = self.buf, self.amax, self.written
        self.buf = self.amax = None
        self.written = set()
        return buf, am, wr


class CrossAttentionImgFn(torch.autograd.Function):
    """encoder-decoder attention on head images: q (B,Tq,d) and kv (B,Tk,2d: k | v), the cells of two HeadImages (`qh`, `kvh`
    carry their inverse scales and maxima; kv may be a window of a stacked projection with row stride ld >= 2d)"""

    @staticmethod
    def forward(ctx, q, qh, kv, kvh, lens, n_head, drop_p, seed, need_weights=True, o_amax=None):
        lib = _lib.load()
        q = _chk_window(q, "cross_attention.q")
        kv = _chk_window(kv, "cross_attention.kv")
        lens = _chk(lens, "cross_attention.lens", torch.int64)
        B, Tq, d = q.shape
        Tk = kv.shape[1]
        if kv.shape[2] != 2 * d or qh.row_inv.shape[1] < B * Tq or kvh.row_inv.shape[1] < B * Tk:
            raise ValueError("cross_attention: q / kv head images do not match")
        ldq, ldk = q.stride(1), kv.stride(1)
        o = torch.empty(B, Tq, d, dtype=torch.float32, device=q.device)
        stat = torch.empty(6, B, n_head, Tq, dtype=torch.float32, device=q.device)
        attn = torch.empty(B, n_head, Tq, Tk, dtype=torch.float32, device=q.device) if need_weights else None
        _lib.check(lib.ttts_attention_fwd_img(_p(q), _off(kv, 0), _off(kv, d), qh.inv_of(0), kvh.inv_of(0), kvh.inv_of(d), _p(o),
                                              _p(stat[0]), _p(attn), _p(lens), B, n_head, Tq, Tk, ldq, ldk, ldk, d, 0, 0.125,
                                              float(drop_p), seed, _ss(), _p(kvh.amax_of(d)), _p(o_amax), _p(stat[1:]),
                                              qh.row_inv.shape[1], kvh.row_inv.shape[1], 0, _stream()),
                   "ttts_attention_fwd_img")
        ctx.save_for_backward(q, kv, o, stat, lens)
        ctx.himg = (qh, kvh)
        ctx.cfg = (n_head, float(drop_p), seed)
        ctx.ss = _ss()
        if attn is None:
            attn = torch.empty(0, dtype=torch.float32, device=q.device)
        ctx.mark_non_differentiable(attn)
        ctx.set_materialize_grads(False)
        return o, attn

    @staticmethod
    def backward(ctx, do, _dattn):
        if do is None:
            return (None,) * 10
        lib = _lib.load()
        q, kv, o, stat, lens = ctx.saved_tensors
        qh, kvh = ctx.himg
        n_head, drop_p, seed = ctx.cfg
        B, Tq, d = q.shape
        Tk = kv.shape[1]
        ldq, ldk = q.stride(1), kv.stride(1)
        do = _chk(do, "cross_attention.do")
        dq = torch.empty(B, Tq, d, dtype=torch.float32, device=q.device)
        delta = torch.empty(B, n_head, Tq, dtype=torch.float32, device=q.device)
        am_q = _amax_slots(q.device, True)
        slab = kvh.slab
        if slab is not None:         # the producer gathers its windows' gradients in one buffer (and one maxima array)
            dkv = slab.window(q.device, kvh.col0, 2 * d)
            am_kv = slab.amax
        else:
            dkv, am_kv = torch.empty(B, Tk, 2 * d, dtype=torch.float32, device=q.device), _amax_slots(q.device, True)
        ldg = dkv.stride(1)
        # few key blocks and many queries (cross-attention: 256 workgroups of one 128-key block each): split the query range
        nsp = _dkv_query_splits(B * n_head * -(-Tk // 128), Tq)
        part = torch.empty(nsp, B, Tk, 2 * d, dtype=torch.float32, device=q.device) if nsp > 1 else None
        _lib.check(lib.ttts_attention_bwd_img(_p(q), _off(kv, 0), _off(kv, d), qh.inv_of(0), kvh.inv_of(0), kvh.inv_of(d), _p(o), _p(do),
                                              _p(stat[1:]), _p(delta), _p(dq), _off(dkv, 0), _off(dkv, d), _p(lens), B, n_head, Tq, Tk,
                                              ldq, ldk, ldk, d, d, ldg, ldg, 0, 0.125, drop_p, seed, ctx.ss, _p(_amax(do)),
                                              _p(am_q), _p(am_kv), _p(part), nsp, qh.row_inv.shape[1], kvh.row_inv.shape[1], 0,
                                              _stream()), "ttts_attention_bwd_img")
        dq._ttts_amax = am_q
        if slab is None:
            dkv._ttts_amax = am_kv
        return dq, None, dkv, None, None, None, None, None, None, None


class SelfAttentionFn(torch.autograd.Function):
    """o = softmax(mask(q k^T / sqrt(head_dim))) v over a packed in-proj output qkv (B,T,3d); head_dim <= 64 (heads of 64 are
    read in place, narrower ones through zero-padded copies)."""

    @staticmethod
    def forward(ctx, qkv, lens, n_head, causal, drop_p, seed, qkv_amax=None, o_amax=None):
        qkv = _chk(qkv, "self_attention.qkv")
        lens = _chk(lens, "self_attention.lens", torch.int64)
        B, T, d3 = qkv.shape
        d = d3 // 3
        hd = _head_dim(d, n_head)
        if qkv_amax is None and _attn_h3():
            qkv_amax = _amax(qkv)
        pads = None
        if hd == 64:
            ptrs, ld = (_off(qkv, 0), _off(qkv, d), _off(qkv, 2 * d)), d3
        else:
            pads = tuple(_pad_heads(qkv, c, d3, B * T, n_head, hd) for c in (0, d, 2 * d))
            ptrs, ld = tuple(_p(t) for t in pads), n_head * 64
        o64, stat, _ = _attn_fwd(*ptrs, ld, ld, ld, B, n_head, T, T, lens, causal, drop_p, seed, False, qkv_amax, qkv_amax,
                                 qkv_amax, o_amax, q_scale=hd ** -0.5)
        if hd == 64:
            o = o64
        else:
            o = torch.empty(B, T, d, dtype=torch.float32, device=qkv.device)
            _unpad_heads(o64, o, 0, d, B * T, n_head, hd)
        ctx.save_for_backward(qkv, o64, stat, lens, *(pads or ()))
        ctx.qkv_amax = qkv_amax
        ctx.cfg = (n_head, causal, float(drop_p), seed, hd)
        ctx.ss = _ss()
        return o

    @staticmethod
    def backward(ctx, do):
        lib = _lib.load()
        qkv, o64, stat, lens, *pads = ctx.saved_tensors
        lse, rowstat = stat[0], (stat[1:] if stat.shape[0] == 4 else None)
        n_head, causal, drop_p, seed, hd = ctx.cfg
        B, T, d3 = qkv.shape
        d = d3 // 3
        do = _chk(do, "self_attention.do")
        dqkv = torch.empty_like(qkv)
        delta = torch.empty(lse.shape, dtype=torch.float32, device=lse.device)
        am = _amax_slots(qkv.device, True) if ATTN_BWD_MODE == "h3" else None     # max|dqkv| for the in-projection gradients
        qa = ctx.qkv_amax
        if hd == 64:
            ins, ld, do64 = (_off(qkv, 0), _off(qkv, d), _off(qkv, 2 * d)), d3, do
            outs, ldg = (_off(dqkv, 0), _off(dqkv, d), _off(dqkv, 2 * d)), d3
        else:
            ins, ld = tuple(_p(t) for t in pads), n_head * 64
            do64 = _pad_heads(do, 0, d, B * T, n_head, hd)
            grads = tuple(torch.empty(B * T, n_head * 64, dtype=torch.float32, device=qkv.device) for _ in range(3))
            outs, ldg = tuple(_p(t) for t in grads), n_head * 64
        _lib.check(_attn_bwd(lib, do64, am, am, qa, qa, qa, rowstat, *ins, _p(o64), _p(do64), _p(lse), _p(delta), *outs, _p(lens),
                             B, n_head, T, T, ld, ld, ld, n_head * 64, ldg, ldg, ldg, 1 if causal else 0, hd ** -0.5, drop_p,
                             seed, ctx.ss), "ttts_attention_bwd")
        if hd != 64:
            for g, c in zip(grads, (0, d, 2 * d)):
                _unpad_heads(g, dqkv, c, d3, B * T, n_head, hd)
        if am is not None:
            dqkv._ttts_amax = am
        return dqkv, None, None, None, None, None, None, None


class CrossAttentionFn(torch.autograd.Function):
    """Encoder-decoder attention: q (B,Tq,d), packed kv (B,Tk,2d) -> o (B,Tq,d), weights (B,H,Tq,Tk) post-dropout."""

    @staticmethod
    def forward(ctx, q, kv, lens, n_head, drop_p, seed, need_weights=True, q_amax=None, kv_amax=None, o_amax=None):
        q = _chk(q, "cross_attention.q")
        kv = _chk(kv, "cross_attention.kv")
        lens = _chk(lens, "cross_attention.lens", torch.int64)
        B, Tq, d = q.shape
        Tk = kv.shape[1]
        hd = _head_dim(d, n_head)
        if _attn_h3():
            q_amax = _amax(q) if q_amax is None else q_amax
            kv_amax = _amax(kv) if kv_amax is None else kv_amax
        pads = None
        if hd == 64:
            ptrs, lds = (_off(q, 0), _off(kv, 0), _off(kv, d)), (d, 2 * d, 2 * d)
        else:
            pads = (_pad_heads(q, 0, d, B * Tq, n_head, hd), _pad_heads(kv, 0, 2 * d, B * Tk, n_head, hd),
                    _pad_heads(kv, d, 2 * d, B * Tk, n_head, hd))
            ptrs, lds = tuple(_p(t) for t in pads), (n_head * 64,) * 3
        o64, stat, attn = _attn_fwd(*ptrs, *lds, B, n_head, Tq, Tk, lens, False, drop_p, seed, need_weights, q_amax, kv_amax,
                                    kv_amax, o_amax, q_scale=hd ** -0.5)
        if hd == 64:
            o = o64
        else:
            o = torch.empty(B, Tq, d, dtype=torch.float32, device=q.device)
            _unpad_heads(o64, o, 0, d, B * Tq, n_head, hd)
        ctx.save_for_backward(q, kv, o64, stat, lens, *(pads or ()))
        ctx.amax = (q_amax, kv_amax)
        ctx.cfg = (n_head, float(drop_p), seed, hd)
        ctx.ss = _ss()
        if attn is None:       # weights not requested: single-pass online softmax, nothing written
            attn = torch.empty(0, dtype=torch.float32, device=q.device)
        ctx.mark_non_differentiable(attn)
        ctx.set_materialize_grads(False)     # or the engine fills a zero "gradient" the size of the weights every backward
        return o, attn

    @staticmethod
    def backward(ctx, do, _dattn):
        if do is None:
            return None, None, None, None, None, None, None, None, None, None
        lib = _lib.load()
        q, kv, o64, stat, lens, *pads = ctx.saved_tensors
        lse, rowstat = stat[0], (stat[1:] if stat.shape[0] == 4 else None)
        n_head, drop_p, seed, hd = ctx.cfg
        B, Tq, d = q.shape
        Tk = kv.shape[1]
        do = _chk(do, "cross_attention.do")
        dq = torch.empty_like(q)
        dkv = torch.empty_like(kv)
        delta = torch.empty(lse.shape, dtype=torch.float32, device=lse.device)
        am_q = am_kv = None
        if ATTN_BWD_MODE == "h3":
            am_q, am_kv = _amax_slots(q.device, True), _amax_slots(q.device, True)
        qa, kva = ctx.amax
        if hd == 64:
            ins, lds, do64 = (_off(q, 0), _off(kv, 0), _off(kv, d)), (d, 2 * d, 2 * d), do
            outs, ldg = (_off(dq, 0), _off(dkv, 0), _off(dkv, d)), (d, 2 * d, 2 * d)
        else:
            ins, lds = tuple(_p(t) for t in pads), (n_head * 64,) * 3
            do64 = _pad_heads(do, 0, d, B * Tq, n_head, hd)
            grads = (torch.empty(B * Tq, n_head * 64, dtype=torch.float32, device=q.device),
                     torch.empty(B * Tk, n_head * 64, dtype=torch.float32, device=q.device),
                     torch.empty(B * Tk, n_head * 64, dtype=torch.float32, device=q.device))
            outs, ldg = tuple(_p(t) for t in grads), (n_head * 64,) * 3
        _lib.check(_attn_bwd(lib, do64, am_q, am_kv, qa, kva, kva, rowstat, *ins, _p(o64), _p(do64), _p(lse), _p(delta), *outs,
                             _p(lens), B, n_head, Tq, Tk, *lds, n_head * 64, *ldg, 0, hd ** -0.5, drop_p, seed, ctx.ss),
                   "ttts_attention_bwd")
        if hd != 64:
            _unpad_heads(grads[0], dq, 0, d, B * Tq, n_head, hd)
            _unpad_heads(grads[1], dkv, 0, 2 * d, B * Tk, n_head, hd)
            _unpad_heads(grads[2], dkv, d, 2 * d, B * Tk, n_head, hd)
        if am_q is not None:
            dq._ttts_amax, dkv._ttts_amax = am_q, am_kv
        return dq, dkv, None, None, None, None, None, None, None, None


def self_attention(qkv, lens, n_head: int, causal: bool, drop_p: float, seed: int):
    """Self-attention over a packed in-projection output (a Tensor, or the HeadImage `linear(..., head_image_sections=3)`
    returned); the partial maxima of a fp32 `qkv` ride on it when its producer left them (`linear(..., publish_amax=True)`), and
    the context leaves with its own for the out-projection."""
    d = qkv.shape[-1] // 3
    if _wide_heads(d, n_head):
        if isinstance(qkv, HeadImage):
            raise ValueError("self_attention: head images hold 64-column heads")
        return _attention_wide_heads(qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:], lens, n_head, causal, drop_p)[0]
    if isinstance(qkv, HeadImage):             # the in-projection left a head image: the LDS-DMA kernels
        o_am = _amax_slots(qkv.device, True)
        q_full = _twin(qkv.cells)
        twin = [q_full, _twin(lens), []] if q_full is not None else None
        o = SelfAttentionImgFn.apply(qkv.cells, qkv.row_inv, qkv.sec_amax, lens, n_head, causal, drop_p, seed, o_am, twin)
        if twin is not None:
            o._ttts_twin = twin[2][0]
        o._ttts_amax = o_am
        return o
    if _twin(qkv) is not None:
        raise ValueError("self_attention: a twin batch runs on head images only")
    h3 = qkv.is_cuda and _attn_h3()
    am = _amax(qkv) if h3 else None
    o_am = _amax_slots(qkv.device, True) if (qkv.is_cuda and ATTN_FWD_MODE == "h3") else None
    o = SelfAttentionFn.apply(qkv, lens, n_head, causal, drop_p, seed, am, o_am)
    if o_am is not None:
        o._ttts_amax = o_am
    return o


def cross_attention(q, kv, lens, n_head: int, drop_p: float, seed: int, need_weights: bool = True):
    """q (B,Tq,d), kv (B,Tk,2d): both fp32 tensors, or both HeadImages (kv may be one layer's window of `cross_kv_projection`)"""
    d = q.shape[-1]
    if _wide_heads(d, n_head):
        if isinstance(q, HeadImage) or isinstance(kv, HeadImage):
            raise ValueError("cross_attention: head images hold 64-column heads")
        o, attn = _attention_wide_heads(q, kv[..., :d], kv[..., d:], lens, n_head, False, drop_p)
        return o, (attn if need_weights else None)
    if isinstance(q, HeadImage) != isinstance(kv, HeadImage):
        raise ValueError("cross_attention: q and kv must both be head images or both fp32")
    if isinstance(q, HeadImage):
        o_am = _amax_slots(q.device, True)
        o, attn = CrossAttentionImgFn.apply(q.cells, q, kv.cells, kv, lens, n_head, drop_p, seed, need_weights, o_am)
        o._ttts_amax = o_am
        return o, attn
    h3 = q.is_cuda and _attn_h3()
    q_am, kv_am = (_amax(q), _amax(kv)) if h3 else (None, None)
    o_am = _amax_slots(q.device, True) if (q.is_cuda and ATTN_FWD_MODE == "h3") else None
    o, attn = CrossAttentionFn.apply(q, kv, lens, n_head, drop_p, seed, need_weights, q_am, kv_am, o_am)
    if o_am is not None:
        o._ttts_amax = o_am
    return o, attn


# ----------------------------------------------------------------------------------------------- small pieces
class EmbeddingFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ids, table, out_amax=None, twin=None):
        lib = _lib.load()
        ids = _chk(ids, "embedding.ids", torch.int64)
        table = _chk(table, "embedding.weight")
        vocab, d = table.shape
        k = 2 if twin is not None else 1                   # (twin batch, see `_twin`)
        out_k = torch.empty(k * ids.shape[0], *ids.shape[1:], d, dtype=torch.float32, device=table.device)
        _lib.check(lib.ttts_embedding_fwd(_p(ids), _p(table), _p(out_k), k * ids.numel(), vocab, d, _p(out_amax), _stream()),
                   "ttts_embedding_fwd")
        out = out_k[:ids.shape[0]] if twin is not None else out_k
        if twin is not None:
            twin[1].append(out_k)
        ctx.save_for_backward(ids)
        ctx.shape = (vocab, d)
        ctx.sinks = _sinks(table)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        (ids,) = ctx.saved_tensors
        vocab, d = ctx.shape
        dout = _chk(dout, "embedding.dout")
        sk, acc, _ = ctx.sinks
        dtable = None
        if sk is not None:
            t = sk[0]
        else:
            t = dtable = torch.empty(vocab, d, dtype=torch.float32, device=dout.device)
        _lib.check(lib.ttts_embedding_bwd(_p(ids), _p(dout), _p(t), ids.numel(), vocab, d, acc, _stream()),
                   "ttts_embedding_bwd")
        return None, dtable, None, None


def embedding(ids, table):
    out_am = _amax_slots(table.device, True) if table.is_cuda else None
    i_full = _twin(ids)
    twin = [i_full, []] if i_full is not None else None
    out = EmbeddingFn.apply(ids, table, out_am, twin)
    if twin is not None:
        out._ttts_twin = twin[1][0]
    if out_am is not None:
        out._ttts_amax = out_am
    return out


class PosEncFn(torch.autograd.Function):
    """y = drop(x + alpha * pe[:T])"""

    @staticmethod
    def forward(ctx, x, pe, alpha, drop_p, seed, y_amax=None, twin=None):
        lib = _lib.load()
        x = _chk(x, "posenc.x")
        B, T, d = x.shape
        if T > pe.shape[0] or d != pe.shape[1]:
            raise ValueError("posenc: sequence longer than the table or width mismatch")
        k = 2 if twin is not None else 1                   # (twin batch, see `_twin`)
        y_k = torch.empty(k * B, T, d, dtype=torch.float32, device=x.device)
        _lib.check(lib.ttts_posenc_fwd(_p(x), _p(pe), _p(alpha), _p(y_k), k * B, T, d, float(drop_p), seed, _ss(), _p(y_amax),
                                       _stream()), "ttts_posenc_fwd")
        y = y_k[:B] if twin is not None else y_k
        if twin is not None:
            twin[1].append(y_k)
        ctx.save_for_backward(pe)
        ctx.cfg = (float(drop_p), seed)
        ctx.ss = _ss()
        ctx.sinks = _sinks(alpha)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        (pe,) = ctx.saved_tensors
        drop_p, seed = ctx.cfg
        dy = _chk(dy, "posenc.dy")
        B, T, d = dy.shape
        dx = torch.empty_like(dy)
        sk, acc, _ = ctx.sinks
        dalpha = None
        if sk is not None:
            t = sk[0]
        else:
            t = dalpha = torch.empty(1, dtype=torch.float32, device=dy.device)
        ws = _ws(lib.ttts_posenc_bwd_workspace_bytes(), dy.device)
        _lib.check(lib.ttts_posenc_bwd(_p(dy), _p(pe), _p(dx), _p(t), _p(ws), ws.numel() * 4, B, T, d, drop_p, seed, ctx.ss, acc,
                                       _stream()), "ttts_posenc_bwd")
        return dx, None, dalpha, None, None, None, None


def posenc(x, pe, alpha, drop_p: float, seed: int):
    y_am = _amax_slots(x.device, True) if x.is_cuda else None
    x_full = _twin(x)
    twin = [x_full, []] if x_full is not None else None
    y = PosEncFn.apply(x, pe, alpha, drop_p, seed, y_am, twin)
    if twin is not None:
        y._ttts_twin = twin[1][0]
    if y_am is not None:
        y._ttts_amax = y_am
    return y


class AddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y):
        lib = _lib.load()
        x = _chk(x, "add.x")
        y = _chk(y, "add.y")
        z = torch.empty_like(x)
        _lib.check(lib.ttts_add(_p(x), _p(y), _p(z), x.numel(), _stream()), "ttts_add")
        return z

    @staticmethod
    def backward(ctx, dz):
        return dz, dz


class FanoutFn(torch.autograd.Function):
    """n aliases of x for n consumers.  Autograd would add their gradients with one stock `aten::add_` per extra consumer;
    here they meet in ONE launch (two or three consumers: the encoder memory under three decoder layers, the mel prediction
    under the loss, the post-net and its residual)."""

    @staticmethod
    def forward(ctx, x, n):
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *gs):
        lib = _lib.load()
        gs = [_chk(g, "fanout.grad") for g in gs if g is not None]
        if not gs:
            return None, None
        while len(gs) > 1:
            out = torch.empty_like(gs[0])
            if len(gs) >= 3 and out.numel() % 4 == 0:
                _lib.check(lib.ttts_add3(_p(gs[0]), _p(gs[1]), _p(gs[2]), _p(out), out.numel(), _stream()), "ttts_add3")
                gs = [out] + gs[3:]
            elif out.numel() % 4 == 0:
                _lib.check(lib.ttts_add(_p(gs[0]), _p(gs[1]), _p(out), out.numel(), _stream()), "ttts_add")
                gs = [out] + gs[2:]
            else:
                gs = [gs[0] + gs[1]] + gs[2:]
        return gs[0], None


def fanout(x: torch.Tensor, n: int):
    """n handles on x, one per consumer (see FanoutFn); plain aliases when no gradient will flow."""
    if n <= 1 or not (torch.is_grad_enabled() and x.requires_grad):
        return (x,) * max(n, 1)
    outs = FanoutFn.apply(x, n)
    full = getattr(x, "_ttts_twin", None)
    if full is not None:               # aliases of the first half of a twin batch are first halves of it
        for o in outs:
            o._ttts_twin = full
    amax = getattr(x, "_ttts_amax", None)
    if amax is not None:               # the producer's partial maxima describe every alias: consumers need no amax pass
        for o in outs:
            o._ttts_amax = amax
    return outs


# ----------------------------------------------------------------------------------------------- loss / mix
class TTSLossFn(torch.autograd.Function):
    """(total, pred_mel, post_mel, stop) of TransformerTTSLoss in one streaming reduction (no boolean-index gathers).  The
    four scalars are separate outputs (views of one 4-vector), so taking `total` costs no select / select_backward
    kernels and an unused output costs nothing in backward."""

    @staticmethod
    def forward(ctx, pred, post, stop, mel, lens, pos_weight):
        lib = _lib.load()
        pred, post, stop, mel = (_chk(pred, "loss.pred"), _chk(post, "loss.post"), _chk(stop, "loss.stop"),
                                 _chk(mel, "loss.mel"))
        lens = _chk(lens, "loss.lengths", torch.int64)
        B, T, C = pred.shape
        if post.shape != pred.shape or mel.shape != pred.shape or stop.shape != (B, T):
            raise ValueError("loss: shape mismatch between pred / post / mel / stop")
        out = torch.empty(4, dtype=torch.float32, device=pred.device)
        ws = _ws(lib.ttts_loss_workspace_bytes(), pred.device)
        _lib.check(lib.ttts_loss_fwd(_p(pred), _p(post), _p(stop), _p(mel), _p(lens), _p(out), _p(ws), ws.numel() * 4, B, T, C,
                                     float(pos_weight), _stream()), "ttts_loss_fwd")
        ctx.save_for_backward(pred, post, stop, mel, lens, ws)
        ctx.pos_weight = float(pos_weight)
        ctx.set_materialize_grads(False)
        return out[0], out[1], out[2], out[3]

    @staticmethod
    def backward(ctx, g_total, g_pred, g_post, g_stop):
        lib = _lib.load()
        pred, post, stop, mel, lens, ws = ctx.saved_tensors
        B, T, C = pred.shape
        gs = [None if g is None else _chk(g, "loss.grad") for g in (g_total, g_pred, g_post, g_stop)]
        dpred, dpost, dstop = torch.empty_like(pred), torch.empty_like(post), torch.empty_like(stop)
        _lib.check(lib.ttts_loss_bwd(_p(pred), _p(post), _p(stop), _p(mel), _p(lens), _p(ws), _p(gs[0]), _p(gs[1]), _p(gs[2]),
                                     _p(gs[3]), _p(dpred), _p(dpost), _p(dstop), B, T, C, ctx.pos_weight, _stream()),
                   "ttts_loss_bwd")
        return dpred, dpost, dstop, None, None, None


def sched_sampling_mix(pred, mel, u, lens, p_tf: float, l_bar: int = 8, seed: int = 0):
    """Block-wise scheduled-sampling mix on the device.  `u` is the (B,T) uniform draw, or None to draw it inside the
    kernel from `seed` (and the active StepState's seed word); with an active StepState p_tf is read from it."""
    lib = _lib.load()
    pred, mel = _chk(pred, "mix.pred"), _chk(mel, "mix.mel")
    if u is not None:
        u = _chk(u, "mix.u")
    lens = _chk(lens, "mix.lens", torch.int64)
    B, T, C = pred.shape
    out = torch.empty_like(mel)
    out_am = _amax_slots(mel.device, True)       # the decoder pre-net's weight gradient reads the mix as an fp16x3 operand
    _lib.check(lib.ttts_sched_sampling_mix(_p(pred), _p(mel), _p(u), _p(lens), _p(out), B, T, C, float(p_tf), l_bar, seed,
                                           _ss(), _p(out_am), _stream()), "ttts_sched_sampling_mix")
    out._ttts_amax = out_am
    return out
