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


def _ws(nbytes: int, device) -> torch.Tensor:
    return torch.empty((max(int(nbytes), 16) + 3) // 4, dtype=torch.float32, device=device)


class _SeedStream:
    """64-bit dropout seeds: one per dropout site per forward call (the kernels hash seed + element index)."""

    def __init__(self, base: int = 0x5EED5EED):
        self.base = base
        self.counter = 0

    def manual_seed(self, seed: int) -> None:
        self.base = int(seed) & 0xFFFFFFFFFFFF
        self.counter = 0

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
GEMM_MODE = os.environ.get("TTTS_GEMM_MODE", "x6")
ATTN_MODE = os.environ.get("TTTS_ATTN_MODE", GEMM_MODE)       # attention products (backward; forward unless ATTN_FWD_MODE): "x6" or "f32"
ATTN_FWD_MODE = os.environ.get("TTTS_ATTN_FWD_MODE", "h3" if ATTN_MODE == "x6" else ATTN_MODE)   # forward: "h3", "x6", "f32"
ATTN_BWD_MODE = os.environ.get("TTTS_ATTN_BWD_MODE", "h3" if ATTN_MODE == "x6" else ATTN_MODE)   # backward: "h3", "x6", "f32"
WGRAD_MODE = os.environ.get("TTTS_WGRAD_MODE", "h3" if GEMM_MODE == "x6" else GEMM_MODE)   # weight gradients: "h3", "x6", "f32"
# Forward GEMMs (nn.Linear / Conv1d forward) under GEMM_MODE "x6": "h3" = fp16x3 split (three f16 MFMA terms on operands
# pre-scaled into f16's range: O(1) activations, O(1/sqrt(fan_in)) weights; csrc/gemm_h3.hip), "x6" = bf16x6 as the
# gradients use.  Shapes the fp16 kernel cannot take (K or channels not a multiple of 32) go to bf16x6.
FWD_MODE = os.environ.get("TTTS_FWD_MODE", "h3")


# Data-gradient GEMMs: "h3" = fp16x3 with a dynamic pre-scale of the gradient operand (one amax pass over dy), "x6".
BWD_MODE = os.environ.get("TTTS_BWD_MODE", "h3")


def _fwd_h3(K: int, N: int, channels: int = 0) -> bool:
    """fp16x3 takes reduction depths (and conv channel counts) that are multiples of 32 and output widths % 4 == 0."""
    return GEMM_MODE == "x6" and FWD_MODE == "h3" and K % 32 == 0 and channels % 32 == 0 and N % 4 == 0


def _bwd_h3(K: int, N: int, channels: int = 0) -> bool:
    return GEMM_MODE == "x6" and BWD_MODE == "h3" and K % 32 == 0 and channels % 32 == 0 and N % 4 == 0


def _amax(t: torch.Tensor) -> torch.Tensor:
    """1024 partial maxima of |t| (device), the dynamic pre-scale input of the fp16x3 gradient GEMMs: the array the kernel
    that produced `t` left on it (`_ttts_amax`, see _amax_slots), or a separate pass over `t`."""
    ready = getattr(t, "_ttts_amax", None)
    if ready is not None:
        return ready
    out = torch.empty(1024, dtype=torch.float32, device=t.device)
    _lib.check(_lib.load().ttts_amax_partials(_p(t), t.numel(), _p(out), _stream()), "ttts_amax_partials")
    return out


class _AmaxArena:
    """Per-device pool of 1024-float partial-maxima arrays that ONE memset zeroes for a whole backward pass.  ~20 kernels
    of a backward pass fill such an array with atomic maxima and each would otherwise need a memset of its own."""
    SLICES = 64

    def __init__(self, device):
        self.buf = torch.empty(self.SLICES, 1024, dtype=torch.float32, device=device)
        self.next = 0
        self.clean = False          # True between reset() and release(): unissued slices are known to be zero

    def reset(self) -> None:
        _lib.check(_lib.load().ttts_zero(_p(self.buf), self.buf.numel() * 4, _stream()), "ttts_zero")
        self.next, self.clean = 0, True

    def take(self):
        if not self.clean or self.next >= self.SLICES:
            return None
        self.next += 1
        return self.buf[self.next - 1]


_amax_arenas = {}


def amax_arena_reset(device) -> None:
    """Call once per step in front of backward (step.TrainStep does)."""
    arena = _amax_arenas.get(device)
    if arena is None:
        arena = _amax_arenas[device] = _AmaxArena(device)
    arena.reset()


def amax_arena_release(device) -> None:
    """After backward: arrays handed out from here on are zeroed individually again."""
    arena = _amax_arenas.get(device)
    if arena is not None:
        arena.clean = False


def _amax_slots(device, zero: bool) -> torch.Tensor:
    """A 1024-float array for a gradient-producing kernel to leave the partial maxima of its output in (attached to that
    output as `_ttts_amax`; a tensor attribute survives the hop to the next autograd Function, and when it does not the
    consumer simply runs the separate pass).  `zero`: the producer fills it with atomic maxima."""
    if zero:
        arena = _amax_arenas.get(device)
        got = arena.take() if arena is not None else None
        if got is not None:
            return got
    a = torch.empty(1024, dtype=torch.float32, device=device)
    if zero:
        _lib.check(_lib.load().ttts_zero(_p(a), 4096, _stream()), "ttts_zero")
    return a


# ----------------------------------------------------------------------------------------------- deferred reductions
# The parameter-gradient kernels end in a small reduction of their partials into the gradient sink.  With sinks (the
# results are not read before the pass ends) those ~90 launches are queued in the library and run as two at the end of
# the backward pass: the first deferring backward node registers a final callback with the autograd engine.
DEFER_REDUCE = os.environ.get("TTTS_DEFER_REDUCE", "1") == "1"
_defer_armed = False
_defer_keep: list = []           # workspaces the queued reductions still read


def _defer(acc: int, ws: torch.Tensor) -> int:
    """`accumulate` argument for a parameter-gradient entry point writing into sinks: adds the "may defer" bit and keeps
    `ws` alive until the flush.  Only called from backward nodes (the engine's final-callback queue is open there)."""
    global _defer_armed
    if not DEFER_REDUCE:
        return acc
    if not _defer_armed:
        _lib.check(_lib.load().ttts_reduce_defer_begin(), "ttts_reduce_defer_begin")
        torch.autograd.Variable._execution_engine.queue_callback(_flush_deferred_final)
        _defer_armed = True
    _defer_keep.append(ws)
    return acc | 2


def flush_deferred(final: bool = False) -> None:
    """Run every queued reduction now (on the current stream).  Mid-pass (`final` False: e.g. before a collective over the
    gradients finished so far) the pass keeps deferring afterwards."""
    global _defer_armed
    if not _defer_armed:
        return
    if final:
        _defer_armed = False
    _lib.check(_lib.load().ttts_reduce_defer_flush(0 if final else 1, _stream()), "ttts_reduce_defer_flush")
    _defer_keep.clear()


def _flush_deferred_final() -> None:
    flush_deferred(True)


def abort_deferred() -> None:
    """Drop whatever an interrupted backward pass left queued (its final callback never ran)."""
    global _defer_armed
    if _defer_armed or _defer_keep:
        _lib.check(_lib.load().ttts_reduce_defer_abort(), "ttts_reduce_defer_abort")
        _defer_armed = False
        _defer_keep.clear()


def _wgrad_is_split(N: int, K: int) -> bool:
    """Mirror of wgrad_use_x6 (csrc/gemm.hip): shapes the split-precision weight-gradient kernels take (the others run on
    the fp32-MFMA kernel and need no maxima)."""
    return (N >= 128 or N in (80, 96)) and (K >= 128 or K in (80, 96)) and (N >= 128 or K >= 128)


def _attn_bwd(lib, do, dq_am, dkv_am, *args):
    """Attention backward in the configured form; `args` = every C-ABI argument up to step_seed.  dq_am / dkv_am: zeroed
    1024-slot arrays in which the fp16x3 kernels leave max|dq| / max|dk, dv| (the in-projection gradients consume them)."""
    if ATTN_BWD_MODE == "h3":
        return lib.ttts_attention_bwd_h3(*args, _p(_amax(do)), _p(dq_am), _p(dkv_am), _stream())
    return (lib.ttts_attention_bwd_x6 if ATTN_BWD_MODE == "x6" else lib.ttts_attention_bwd)(*args, _stream())


def _wgrad(lib, name: str, dy: torch.Tensor, amax, split_ok: bool, *args):
    """Weight-gradient entry point `name` in the configured form; `args` = everything after (dy ...) up to `accumulate`.
    The fp16x3 form takes the partial maxima of |dy| (computed here unless the caller already has them); shapes the split
    kernels do not take (`split_ok` False) run on the fp32-MFMA kernel behind the _x6 entry point and need none."""
    if WGRAD_MODE == "h3" and split_ok:
        am = amax if amax is not None else _amax(dy)
        return getattr(lib, name + "_h3")(_p(dy), *args, _p(am), _stream())
    return getattr(lib, name + ("_x6" if WGRAD_MODE in ("x6", "h3") else ""))(_p(dy), *args, _stream())


_param_epoch = 0


def bump_param_epoch() -> None:
    """Invalidate every cached weight split.  Called by optimizers that update parameters through raw pointers
    (optim.FlatAdam), which autograd's per-tensor version counters cannot see."""
    global _param_epoch
    _param_epoch += 1


class _PlaneEntry:
    __slots__ = ("wref", "off", "mode", "rows", "cols", "c2", "taps", "planes", "tag")


_BATCHED_SPLIT = os.environ.get("TTTS_BATCHED_SPLIT", "1") == "1"
_plane_entries: list = []        # every (weight, mode) split so far, for the one-launch refresh after an optimizer step
_plane_table = None              # (signature, device descriptor table, pinned host copy, total blocks)


def _refresh_all_planes() -> None:
    """Re-split every registered weight whose storage and version are unchanged (only the parameter epoch moved, i.e. an
    optimizer stepped through raw pointers) with ONE launch instead of one per weight and mode."""
    global _plane_table, _plane_entries
    live, sig = [], []
    for e in _plane_entries:
        w = e.wref()
        if w is None:
            continue
        live.append(e)
        if e.tag[0] == w._version and e.tag[1] == w.data_ptr() + e.off and e.tag[2] != _param_epoch:
            sig.append((e.tag[1], e.planes.data_ptr(), e.rows, e.cols, e.mode, e.c2, e.taps))
    _plane_entries = live
    if not sig:
        return
    sig_t = tuple(sig)
    if _plane_table is None or _plane_table[0] != sig_t:
        rows, blk = [], 0
        for s in sig:
            rows.append(list(s) + [blk])
            blk += (s[2] * s[3] + 255) // 256
        host = torch.tensor(rows, dtype=torch.int64).pin_memory()      # page-locked: the upload does not synchronise
        _plane_table = (sig_t, host.to(live[0].planes.device, non_blocking=True), host, blk)
    lib = _lib.load()
    _lib.check(lib.ttts_weight_split_batched(_p(_plane_table[1]), len(sig), _plane_table[3], _stream()),
               "ttts_weight_split_batched")
    refreshed = {s[1] for s in sig}
    for e in live:
        if e.planes.data_ptr() in refreshed:
            e.tag = (e.tag[0], e.tag[1], _param_epoch)


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
        if _BATCHED_SPLIT and ent.tag[0] == tag[0] and ent.tag[1] == tag[1]:   # only the epoch moved: refresh everything at once
            _refresh_all_planes()
            if ent.tag == tag:
                return ent.planes
    lib = _lib.load()
    if ent is None or ent.planes.numel() != 3 * rows * cols or ent.planes.device != w.device:
        planes = torch.empty(3 * rows * cols, dtype=torch.int16, device=w.device)
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
    if p.dim() == 2:
        v._ttts_planes_owner = (p, r0)
    return v


def _sink(t: Optional[torch.Tensor]):
    """Gradient sink of a parameter: a slice of the flat data-parallel gradient bucket (parallel.FlatGradBucket).
    When every parameter of an op has one, the backward kernels ADD their result straight into it
    (`accumulate=1`) and autograd gets None -- no per-tensor accumulation kernels, no flatten copy."""
    return None if t is None else getattr(t, "_ttts_grad_sink", None)


def _sinks(*params):
    """(list of destinations or None, accumulate flag): sinks are used only if all given parameters have one."""
    live = [p for p in params if p is not None]
    sk = [_sink(p) for p in live]
    if live and all(x is not None for x in sk):
        return [(_sink(p) if p is not None else None) for p in params], 1
    return None, 0


# ----------------------------------------------------------------------------------------------- linear
class LinearFn(torch.autograd.Function):
    """y = drop(act(x @ w.T + b)) + residual, rows optionally shifted by `row_shift` inside each utterance."""

    @staticmethod
    def forward(ctx, x, w, b, residual, act, drop_p, seed, row_shift, T, tok_out=None, tok_in=None, skip_in=None,
                skip_out=None, tok_drop=None):
        lib = _lib.load()
        x = _chk(x, "linear.x")
        w = _chk(w, "linear.weight")
        N, K = w.shape
        if x.shape[-1] != K:
            raise ValueError(f"linear: x has {x.shape[-1]} features, weight expects {K}")
        if act == ACT_RELU and residual is not None:
            raise ValueError("linear: relu epilogue cannot be combined with a residual")
        M = x.numel() // K
        y = torch.empty(*x.shape[:-1], N, dtype=torch.float32, device=x.device)
        b_ = _chk(b, "linear.bias") if b is not None else None
        r_ = _chk(residual, "linear.residual") if residual is not None else None
        if r_ is not None and r_.shape != y.shape:
            raise ValueError("linear: residual shape mismatch")
        if _fwd_h3(K, N):
            _lib.check(lib.ttts_linear_fwd_h3(_p(x), _p(_planes(w, 4, N, K)), _p(b_), _p(r_), _p(y), M, N, K, act,
                                              float(drop_p), seed, _ss(), row_shift, T, _stream()), "ttts_linear_fwd_h3")
        elif GEMM_MODE == "x6":
            _lib.check(lib.ttts_linear_fwd_x6(_p(x), _p(_planes(w, 0, N, K)), _p(b_), _p(r_), _p(y), M, N, K, act,
                                              float(drop_p), seed, _ss(), row_shift, T, _stream()), "ttts_linear_fwd_x6")
        else:
            _lib.check(lib.ttts_linear_fwd(_p(x), _p(w), _p(b_), _p(r_), _p(y), M, N, K, act, float(drop_p), seed,
                                           _ss(), row_shift, T, _stream()), "ttts_linear_fwd")
        ctx.save_for_backward(x, w, y if act == ACT_RELU else None)
        ctx.cfg = (act, float(drop_p), seed, row_shift, T, b is not None, residual is not None)
        ctx.sinks = _sinks(w, b)
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
        if act == ACT_RELU and tok_out is not None and tok_out.premasked:
            dacc = dy                    # the consumer's data-gradient epilogue already applied the relu / dropout mask
            tok_out.premasked = False
        elif act == ACT_RELU:
            dacc = torch.empty_like(dy)
            am = torch.empty(1024, dtype=torch.float32, device=dy.device) if want_am else None
            _lib.check(lib.ttts_relu_dropout_bwd(_p(dy), _p(y), _p(dacc), dy.numel(), drop_p, _p(am), _stream()),
                       "ttts_relu_dropout_bwd")
        elif drop_p > 0.0:
            td = ctx.tok_drop
            if td is not None and td.dacc is not None and td.dx is dy:
                dacc, am = td.dacc, td.amax          # written by the LayerNorm backward that produced dy
                td.dx = td.dacc = td.amax = None
            else:
                dacc = torch.empty_like(dy)
                am = torch.empty(1024, dtype=torch.float32, device=dy.device) if want_am else None
                _lib.check(lib.ttts_dropout_bwd(_p(dy), _p(dacc), dy.numel(), drop_p, seed, ctx.ss, _p(am), _stream()),
                           "ttts_dropout_bwd")
        else:
            dacc = dy
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
                am = am if am is not None else _amax(dacc)
                # dx with the producer's relu mask applied here is exactly the `dacc` of that producer's backward: leave its
                # maxima on it
                dx_am = _amax_slots(dx.device, True) if (tok_in is not None or ctx.sole_consumer) else None
                _lib.check(lib.ttts_linear_bwd_data_h3(_p(dacc), _p(_planes(w, 5, K, N)), _p(skip), _p(dx), M, N, K,
                                                       _p(gate), gscale, _p(am), _p(dx_am), _stream()), "ttts_linear_bwd_data_h3")
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
            nbytes = lib.ttts_wgrad_workspace_bytes(M, N, K, 1)
            ws = _ws(nbytes, x.device)
            sk, acc = ctx.sinks
            if sk is not None:
                dw_t, db_t = sk
                acc = _defer(acc, ws)
            else:
                dw_t = dw = torch.empty_like(w)
                db_t = db = torch.empty(N, dtype=torch.float32, device=x.device) if has_b else None
            _lib.check(_wgrad(lib, "ttts_linear_bwd_weight", dacc, am, _wgrad_is_split(N, K), _p(x), _p(dw_t), _p(db_t), _p(ws),
                              ws.numel() * 4, M, N, K, row_shift, T, acc), "ttts_linear_bwd_weight")
        dres = dy if has_r else None
        if has_r and skip_out is not None:      # hand the skip gradient to the block's first Linear instead of autograd
            skip_out.grad, dres = dy, None
        return dx, dw, db, dres, None, None, None, None, None, None, None, None, None, None


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
    __slots__ = ("p", "seed", "ss", "dx", "dacc", "amax")

    def __init__(self, p: float, seed: int, ss):
        self.p, self.seed, self.ss = p, seed, ss
        self.dx = self.dacc = self.amax = None


class SkipToken:
    """Residual block `y = last(f(first(x))) + x` whose first and last ops are Linears: the last one's backward parks
    the skip-connection gradient here and the first one adds it in its data-gradient epilogue, instead of autograd
    summing two full-size tensors with a separate kernel.  Pass the same token as `skip_in` to the first Linear (its
    input must be the block input x) and as `skip_out` to the last one (its `residual` must be the same x)."""
    __slots__ = ("grad",)

    def __init__(self):
        self.grad = None


def linear(x, w, b=None, residual=None, act=ACT_NONE, drop_p=0.0, seed=0, row_shift=0, T=0, sole_consumer=False,
           skip_in=None, skip_out=None):
    """`sole_consumer=True` is the caller's promise that nothing but this Linear reads `x`; if `x` came out of a
    relu(+dropout) Linear, its backward mask is then fused into this Linear's data-gradient epilogue.
    `skip_in` / `skip_out`: see SkipToken."""
    grad_on = torch.is_grad_enabled()
    tok_in = getattr(x, "_ttts_relu_token", None) if (sole_consumer and grad_on) else None
    tok_out = _ReluToken(1.0 / (1.0 - float(drop_p))) if act == ACT_RELU else None
    if not (grad_on and x.requires_grad):
        skip_in = None                      # nobody will pick the gradient up: leave it to autograd
    if skip_out is not None and (residual is None or not grad_on):
        skip_out = None
    tok_drop = None
    if grad_on and act == ACT_NONE and float(drop_p) > 0.0 and residual is not None and x.is_cuda:
        tok_drop = _DropToken(float(drop_p), seed, _ss())
    y = LinearFn.apply(x, w, b, residual, act, drop_p, seed, row_shift, T, tok_out, tok_in, skip_in, skip_out, tok_drop)
    if tok_drop is not None:
        y._ttts_drop_token = tok_drop           # picked up by layer_norm(y, ..., sole_consumer=True)
    if tok_out is not None:
        y._ttts_relu_token = tok_out
        if _relu_observer is not None:
            _relu_observer(y)
    return y


# ----------------------------------------------------------------------------------------------- heads
class HeadsFn(torch.autograd.Function):
    """mel = x @ w_mel.T + b_mel  (B,T,n_mels);  stop = x @ w_stop.T + b_stop  (B,T)  -- one read of dx."""

    @staticmethod
    def forward(ctx, x, w_mel, b_mel, w_stop, b_stop):
        lib = _lib.load()
        x = _chk(x, "heads.x")
        N, K = w_mel.shape
        M = x.numel() // K
        mel = torch.empty(*x.shape[:-1], N, dtype=torch.float32, device=x.device)
        stop = torch.empty(x.shape[:-1], dtype=torch.float32, device=x.device)
        w_mel = _chk(w_mel, "w_mel")
        if _fwd_h3(K, N):
            _lib.check(lib.ttts_linear_fwd_h3(_p(x), _p(_planes(w_mel, 4, N, K)), _p(b_mel), None, _p(mel), M, N, K,
                                              ACT_NONE, 0.0, 0, None, 0, 0, _stream()), "ttts_linear_fwd_h3")
        elif GEMM_MODE == "x6":
            _lib.check(lib.ttts_linear_fwd_x6(_p(x), _p(_planes(w_mel, 0, N, K)), _p(b_mel), None, _p(mel), M, N, K,
                                              ACT_NONE, 0.0, 0, None, 0, 0, _stream()), "ttts_linear_fwd_x6")
        else:
            _lib.check(lib.ttts_linear_fwd(_p(x), _p(w_mel), _p(b_mel), None, _p(mel), M, N, K, ACT_NONE, 0.0,
                                           0, None, 0, 0, _stream()), "ttts_linear_fwd")
        _lib.check(lib.ttts_rowdot_fwd(_p(x), _p(_chk(w_stop, "w_stop")), _p(b_stop), _p(stop), M, K, _stream()),
                   "ttts_rowdot_fwd")
        ctx.save_for_backward(x, w_mel, w_stop)
        ctx.sinks = _sinks(w_mel, b_mel, w_stop, b_stop)
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
        if GEMM_MODE == "x6":
            _lib.check(lib.ttts_linear_bwd_data_x6(_p(dmel), _p(_planes(w_mel, 1, K, N)), None, _p(dx), M, N, K, None, 1.0,
                                                   _stream()),
                       "ttts_linear_bwd_data_x6")
        else:
            _lib.check(lib.ttts_linear_bwd_data(_p(dmel), _p(w_mel), None, _p(dx), M, N, K, None, 1.0, _stream()),
                       "ttts_linear_bwd_data")
        ws = _ws(lib.ttts_wgrad_workspace_bytes(M, N, K, 1), x.device)
        sk, acc = ctx.sinks
        ws2 = _ws(lib.ttts_rowdot_bwd_workspace_bytes(K), x.device)
        if sk is not None:
            t_wm, t_bm, t_ws, t_bs = sk
            dw_mel = db_mel = dw_stop = db_stop = None
            acc = _defer(_defer(acc, ws), ws2)
        else:
            t_wm = dw_mel = torch.empty_like(w_mel)
            t_bm = db_mel = torch.empty(N, dtype=torch.float32, device=x.device)
            t_ws = dw_stop = torch.empty_like(w_stop)
            t_bs = db_stop = torch.empty(1, dtype=torch.float32, device=x.device)
        _lib.check(_wgrad(lib, "ttts_linear_bwd_weight", dmel, None, _wgrad_is_split(N, K), _p(x), _p(t_wm), _p(t_bm), _p(ws),
                          ws.numel() * 4, M, N, K, 0, 0, acc), "ttts_linear_bwd_weight")
        _lib.check(lib.ttts_rowdot_bwd(_p(dstop), _p(x), _p(w_stop), _p(dx), _p(t_ws), _p(t_bs), _p(ws2),
                                       ws2.numel() * 4, M, K, acc, _stream()), "ttts_rowdot_bwd")
        return dx, dw_mel, db_mel, dw_stop, db_stop


# ----------------------------------------------------------------------------------------------- conv + BN
class ConvBNFn(torch.autograd.Function):
    """z = drop(act(BatchNorm1d(Conv1d_same(x)))) on (B,T,C); running stats updated in place when training."""

    @staticmethod
    def forward(ctx, x, conv_w, conv_b, gamma, beta, running_mean, running_var, nbt, training, momentum, eps, act,
                drop_p, seed):
        lib = _lib.load()
        x = _chk(x, "conv_bn.x")
        B, T, cin = x.shape
        cout, cin_w, taps = conv_w.shape
        if cin != cin_w:
            raise ValueError(f"conv_bn: x has {cin} channels, weight expects {cin_w}")
        dev = x.device
        conv_w = _chk(conv_w, "conv.weight")
        y = torch.empty(B, T, cout, dtype=torch.float32, device=dev)
        if _fwd_h3(taps * cin, cout, cin):
            _lib.check(lib.ttts_conv1d_fwd_h3(_p(x), _p(_planes(conv_w, 6, cout, taps * cin, cin, taps)), _p(conv_b), _p(y),
                                              B, T, cin, cout, taps, _stream()), "ttts_conv1d_fwd_h3")
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
        M = B * T
        if training:
            ws = _ws(lib.ttts_bn_workspace_bytes(M, cout), dev)
            _lib.check(lib.ttts_bn_train_stats(_p(y), _p(mean), _p(invstd), _p(running_mean), _p(running_var), _p(nbt),
                                               _p(ws), ws.numel() * 4, M, cout, float(momentum), float(eps), _stream()),
                       "ttts_bn_train_stats")
        else:
            _lib.check(lib.ttts_bn_eval_stats(_p(running_mean), _p(running_var), _p(mean), _p(invstd), cout, float(eps),
                                              _stream()), "ttts_bn_eval_stats")
        z = torch.empty_like(y)
        _lib.check(lib.ttts_bn_apply_fwd(_p(y), _p(mean), _p(invstd), _p(gamma), _p(beta), _p(z), M, cout, act,
                                         float(drop_p), seed, _ss(), _stream()), "ttts_bn_apply_fwd")
        ctx.save_for_backward(x, conv_w, y, mean, invstd, gamma, beta)
        ctx.cfg = (training, act, float(drop_p), seed, conv_b is not None)
        ctx.ss = _ss()
        ctx.sinks = _sinks(conv_w, conv_b, gamma, beta)
        return z

    @staticmethod
    def backward(ctx, dz):
        lib = _lib.load()
        x, conv_w, y, mean, invstd, gamma, beta = ctx.saved_tensors
        training, act, drop_p, seed, has_b = ctx.cfg
        if not training:
            raise RuntimeError("conv_bn: backward is implemented for train-mode BatchNorm only")
        B, T, cin = x.shape
        cout, _, taps = conv_w.shape
        dev = x.device
        M = B * T
        dz = _chk(dz, "conv_bn.dz")
        dy = torch.empty_like(y)
        sk, acc = ctx.sinks
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
        am = _amax_slots(dev, False) if want_am else None
        _lib.check(lib.ttts_bn_bwd(_p(dz), _p(y), _p(mean), _p(invstd), _p(gamma), _p(beta), _p(dy), _p(t_g), _p(t_be),
                                   _p(ws), ws.numel() * 4, M, cout, act, drop_p, seed, ctx.ss, acc, _p(am), _stream()),
                   "ttts_bn_bwd")
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
        ws2 = _ws(lib.ttts_wgrad_workspace_bytes(M, cout, cin, taps), dev)
        _lib.check(_wgrad(lib, "ttts_conv1d_bwd_weight", dy, am, _wgrad_is_split(cout, cin), _p(x), _p(t_w), _p(t_b), _p(ws2),
                          ws2.numel() * 4, B, T, cin, cout, taps, _defer(acc, ws2) if sk is not None else acc),
                   "ttts_conv1d_bwd_weight")
        return dx, dw, db, dgamma, dbeta, None, None, None, None, None, None, None, None, None


def conv_bn(x, conv_w, conv_b, gamma, beta, running_mean, running_var, nbt, training, momentum=0.1, eps=1e-5,
            act=ACT_NONE, drop_p=0.0, seed=0):
    return ConvBNFn.apply(x, conv_w, conv_b, gamma, beta, running_mean, running_var, nbt, training, momentum, eps, act,
                          drop_p, seed)


# ----------------------------------------------------------------------------------------------- layer norm
class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps, tok_drop=None):
        lib = _lib.load()
        ctx.tok_drop = tok_drop
        x = _chk(x, "layernorm.x")
        d = x.shape[-1]
        M = x.numel() // d
        y = torch.empty_like(x)
        mean = torch.empty(M, dtype=torch.float32, device=x.device)
        rstd = torch.empty(M, dtype=torch.float32, device=x.device)
        _lib.check(lib.ttts_layernorm_fwd(_p(x), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), M, d, float(eps),
                                          _stream()), "ttts_layernorm_fwd")
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
        sk, acc = ctx.sinks
        dgamma = dbeta = None
        if sk is not None:
            t_g, t_b = sk
        else:
            t_g = dgamma = torch.empty_like(gamma)
            t_b = dbeta = torch.empty_like(gamma)
        ws = _ws(lib.ttts_layernorm_bwd_workspace_bytes(d), x.device)
        accf = _defer(acc, ws) if sk is not None else acc
        td = ctx.tok_drop
        if td is not None and d in (256, 512, 1024):
            # x is the output of a Linear with residual dropout and feeds nothing but this LayerNorm: dx is that Linear's dy
            dacc = torch.empty_like(x)
            am = _amax_slots(x.device, True)
            _lib.check(lib.ttts_layernorm_bwd_drop(_p(dy), _p(x), _p(mean), _p(rstd), _p(gamma), _p(dx), _p(t_g), _p(t_b),
                                                   _p(ws), ws.numel() * 4, M, d, accf, _p(dacc), td.p, td.seed, td.ss, _p(am),
                                                   _stream()), "ttts_layernorm_bwd_drop")
            td.dx, td.dacc, td.amax = dx, dacc, am
        else:
            _lib.check(lib.ttts_layernorm_bwd(_p(dy), _p(x), _p(mean), _p(rstd), _p(gamma), _p(dx), _p(t_g), _p(t_b),
                                              _p(ws), ws.numel() * 4, M, d, accf, _stream()), "ttts_layernorm_bwd")
        return dx, dgamma, dbeta, None, None


def layer_norm(x, gamma, beta, eps=1e-5, sole_consumer=False):
    """`sole_consumer=True` is the caller's promise that nothing but this LayerNorm reads `x`; if `x` came out of a Linear
    with a residual-dropout epilogue, that Linear's dropout backward is then written by this LayerNorm's backward kernel."""
    tok = getattr(x, "_ttts_drop_token", None) if (sole_consumer and torch.is_grad_enabled()) else None
    return LayerNormFn.apply(x, gamma, beta, eps, tok)


# ----------------------------------------------------------------------------------------------- attention
def _attn_fwd(q, k, v, ldq, ldk, ldv, B, H, Tq, Tk, lens, causal, drop_p, seed, need_weights):
    lib = _lib.load()
    dev = lens.device
    o = torch.empty(B, Tq, H * 64, dtype=torch.float32, device=dev)
    lse = torch.empty(B, H, Tq, dtype=torch.float32, device=dev)
    attn = torch.empty(B, H, Tq, Tk, dtype=torch.float32, device=dev) if need_weights else None
    fwd = {"h3": lib.ttts_attention_fwd_h3, "x6": lib.ttts_attention_fwd_x6}.get(ATTN_FWD_MODE, lib.ttts_attention_fwd)
    _lib.check(fwd(q, k, v, _p(o), _p(lse), _p(attn), _p(lens), B, H, Tq, Tk, ldq, ldk, ldv, H * 64,
                   1 if causal else 0, float(drop_p), seed, _ss(), _stream()), "ttts_attention_fwd")
    return o, lse, attn


def _off(t: torch.Tensor, col: int):
    return c_void_p(t.data_ptr() + 4 * col)


class SelfAttentionFn(torch.autograd.Function):
    """o = softmax(mask(q k^T / 8)) v over a packed in-proj output qkv (B,T,3d); heads of 64."""

    @staticmethod
    def forward(ctx, qkv, lens, n_head, causal, drop_p, seed):
        qkv = _chk(qkv, "self_attention.qkv")
        lens = _chk(lens, "self_attention.lens", torch.int64)
        B, T, d3 = qkv.shape
        d = d3 // 3
        if d != n_head * 64:
            raise ValueError(f"attention kernels need head_dim 64 (d_model {d}, heads {n_head})")
        o, lse, _ = _attn_fwd(_off(qkv, 0), _off(qkv, d), _off(qkv, 2 * d), d3, d3, d3, B, n_head, T, T, lens, causal,
                              drop_p, seed, False)
        ctx.save_for_backward(qkv, o, lse, lens)
        ctx.cfg = (n_head, causal, float(drop_p), seed)
        ctx.ss = _ss()
        return o

    @staticmethod
    def backward(ctx, do):
        lib = _lib.load()
        qkv, o, lse, lens = ctx.saved_tensors
        n_head, causal, drop_p, seed = ctx.cfg
        B, T, d3 = qkv.shape
        d = d3 // 3
        do = _chk(do, "self_attention.do")
        dqkv = torch.empty_like(qkv)
        delta = torch.empty_like(lse)
        am = _amax_slots(qkv.device, True) if ATTN_BWD_MODE == "h3" else None     # max|dqkv| for the in-projection gradients
        _lib.check(_attn_bwd(lib, do, am, am, _off(qkv, 0), _off(qkv, d), _off(qkv, 2 * d), _p(o), _p(do), _p(lse), _p(delta),
                             _off(dqkv, 0), _off(dqkv, d), _off(dqkv, 2 * d), _p(lens), B, n_head, T, T, d3,
                             d3, d3, d, d3, d3, d3, 1 if causal else 0, drop_p, seed, ctx.ss), "ttts_attention_bwd")
        if am is not None:
            dqkv._ttts_amax = am
        return dqkv, None, None, None, None, None


class CrossAttentionFn(torch.autograd.Function):
    """Encoder-decoder attention: q (B,Tq,d), packed kv (B,Tk,2d) -> o (B,Tq,d), weights (B,H,Tq,Tk) post-dropout."""

    @staticmethod
    def forward(ctx, q, kv, lens, n_head, drop_p, seed, need_weights=True):
        q = _chk(q, "cross_attention.q")
        kv = _chk(kv, "cross_attention.kv")
        lens = _chk(lens, "cross_attention.lens", torch.int64)
        B, Tq, d = q.shape
        Tk = kv.shape[1]
        if d != n_head * 64:
            raise ValueError(f"attention kernels need head_dim 64 (d_model {d}, heads {n_head})")
        o, lse, attn = _attn_fwd(_off(q, 0), _off(kv, 0), _off(kv, d), d, 2 * d, 2 * d, B, n_head, Tq, Tk, lens, False,
                                 drop_p, seed, need_weights)
        ctx.save_for_backward(q, kv, o, lse, lens)
        ctx.cfg = (n_head, float(drop_p), seed)
        ctx.ss = _ss()
        if attn is None:       # weights not requested: single-pass online softmax, nothing written
            attn = torch.empty(0, dtype=torch.float32, device=q.device)
        ctx.mark_non_differentiable(attn)
        ctx.set_materialize_grads(False)     # or the engine fills a zero "gradient" the size of the weights every backward
        return o, attn

    @staticmethod
    def backward(ctx, do, _dattn):
        if do is None:
            return None, None, None, None, None, None, None
        lib = _lib.load()
        q, kv, o, lse, lens = ctx.saved_tensors
        n_head, drop_p, seed = ctx.cfg
        B, Tq, d = q.shape
        Tk = kv.shape[1]
        do = _chk(do, "cross_attention.do")
        dq = torch.empty_like(q)
        dkv = torch.empty_like(kv)
        delta = torch.empty_like(lse)
        am_q = am_kv = None
        if ATTN_BWD_MODE == "h3":
            am_q, am_kv = _amax_slots(q.device, True), _amax_slots(q.device, True)
        _lib.check(_attn_bwd(lib, do, am_q, am_kv, _off(q, 0), _off(kv, 0), _off(kv, d), _p(o), _p(do), _p(lse), _p(delta),
                             _off(dq, 0), _off(dkv, 0), _off(dkv, d), _p(lens), B, n_head, Tq, Tk, d, 2 * d,
                             2 * d, d, d, 2 * d, 2 * d, 0, drop_p, seed, ctx.ss), "ttts_attention_bwd")
        if am_q is not None:
            dq._ttts_amax, dkv._ttts_amax = am_q, am_kv
        return dq, dkv, None, None, None, None, None


# ----------------------------------------------------------------------------------------------- small pieces
class EmbeddingFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ids, table):
        lib = _lib.load()
        ids = _chk(ids, "embedding.ids", torch.int64)
        table = _chk(table, "embedding.weight")
        vocab, d = table.shape
        out = torch.empty(*ids.shape, d, dtype=torch.float32, device=table.device)
        _lib.check(lib.ttts_embedding_fwd(_p(ids), _p(table), _p(out), ids.numel(), vocab, d, _stream()),
                   "ttts_embedding_fwd")
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
        sk, acc = ctx.sinks
        dtable = None
        if sk is not None:
            t = sk[0]
        else:
            t = dtable = torch.empty(vocab, d, dtype=torch.float32, device=dout.device)
        _lib.check(lib.ttts_embedding_bwd(_p(ids), _p(dout), _p(t), ids.numel(), vocab, d, acc, _stream()),
                   "ttts_embedding_bwd")
        return None, dtable


class PosEncFn(torch.autograd.Function):
    """y = drop(x + alpha * pe[:T])"""

    @staticmethod
    def forward(ctx, x, pe, alpha, drop_p, seed):
        lib = _lib.load()
        x = _chk(x, "posenc.x")
        B, T, d = x.shape
        if T > pe.shape[0] or d != pe.shape[1]:
            raise ValueError("posenc: sequence longer than the table or width mismatch")
        y = torch.empty_like(x)
        _lib.check(lib.ttts_posenc_fwd(_p(x), _p(pe), _p(alpha), _p(y), B, T, d, float(drop_p), seed, _ss(), _stream()),
                   "ttts_posenc_fwd")
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
        sk, acc = ctx.sinks
        dalpha = None
        if sk is not None:
            t = sk[0]
        else:
            t = dalpha = torch.empty(1, dtype=torch.float32, device=dy.device)
        ws = _ws(lib.ttts_posenc_bwd_workspace_bytes(), dy.device)
        _lib.check(lib.ttts_posenc_bwd(_p(dy), _p(pe), _p(dx), _p(t), _p(ws), ws.numel() * 4, B, T, d, drop_p, seed, ctx.ss, acc,
                                       _stream()), "ttts_posenc_bwd")
        return dx, None, dalpha, None, None


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
    _lib.check(lib.ttts_sched_sampling_mix(_p(pred), _p(mel), _p(u), _p(lens), _p(out), B, T, C, float(p_tf), l_bar, seed,
                                           _ss(), _stream()), "ttts_sched_sampling_mix")
    return out
