"""FlatAdam: torch.optim.Adam arithmetic (the reference's optimizer, lightning_module.py:160-163) as one fused HIP
kernel over flat parameter / gradient / moment buffers, with global-norm clipping (train.py:41) folded in.

All parameters become views into ONE flat fp32 buffer (layout shared with `parallel.FlatGradBucket`, so the gradient
all-reduce, the norm and the update all stream the same contiguous memory).  It is a `torch.optim.Optimizer`, so
`LambdaLR` (the Noam schedule) drives `param_groups[0]['lr']` exactly as in the reference.
"""
from __future__ import annotations

from ctypes import c_void_p
from typing import Optional

import torch

from . import _lib, ops
from .ops import _p, _stream, _ws
from .parallel import FlatGradBucket


class FlatAdam(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 1.0, betas=(0.9, 0.98), eps: float = 1e-9, max_grad_norm: float = 0.0,
                 bucket: Optional[FlatGradBucket] = None):
        params = [p for p in params if p.requires_grad]
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, max_grad_norm=max_grad_norm))
        self.bucket = bucket if bucket is not None else FlatGradBucket(params)
        if [id(p) for p in self.bucket.params] != [id(p) for p in params]:
            raise ValueError("FlatAdam: the gradient bucket must hold the same parameters in the same order")
        b = self.bucket
        if not b.flat.is_cuda:
            raise ValueError("FlatAdam: parameters must live on the HIP device (no CPU fallback)")
        # move every parameter into one flat buffer with the bucket's layout (64-float aligned slices)
        self.flat_params = torch.zeros_like(b.flat)
        for p, off in zip(b.params, b.offsets):
            n = p.numel()
            self.flat_params[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.flat_params[off:off + n].view_as(p)
        self.exp_avg = torch.zeros_like(b.flat)
        self.exp_avg_sq = torch.zeros_like(b.flat)
        self.grad_norm = torch.zeros(1, dtype=torch.float32, device=b.flat.device)
        self._ws = _ws(_lib.load().ttts_grad_norm_workspace_bytes(), b.flat.device)
        self._step = 0

    def zero_grad(self, set_to_none: bool = False):   # gradients live in the bucket: zero it, keep the views attached
        self.bucket.zero()

    def step(self, closure=None):
        """`closure` (Lightning's automatic optimisation passes one that runs training_step + backward) is evaluated
        with gradients enabled, as torch.optim.Adam does; only the update itself runs under no_grad."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        with torch.no_grad():
            self._update()
        return loss

    def _update(self) -> None:
        lib = _lib.load()
        g = self.param_groups[0]
        b = self.bucket
        b.flush_reductions()        # deferred parameter-gradient reductions of the backward pass(es) land first
        self._step += 1
        n = b.flat.numel()
        norm_ptr = None
        if g["max_grad_norm"] and g["max_grad_norm"] > 0:
            _lib.check(lib.ttts_grad_norm(_p(b.flat), _p(self.grad_norm), _p(self._ws), self._ws.numel() * 4, n, _stream()),
                       "ttts_grad_norm")
            norm_ptr = _p(self.grad_norm)
        _lib.check(lib.ttts_adam_step(_p(self.flat_params), _p(b.flat), _p(self.exp_avg), _p(self.exp_avg_sq), norm_ptr, n,
                                      float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), self._step,
                                      float(g["max_grad_norm"] or 0.0), ops._ss(), _stream()), "ttts_adam_step")
        ops.bump_param_epoch()      # parameters changed behind autograd's version counters: drop cached weight splits

    def note_external_step(self) -> None:
        """Host bookkeeping for an update that ran without `step()` being called -- a replayed HIP graph that contains
        the clip + Adam kernels (step.TrainStep): advance the step count and drop the cached weight splits."""
        self._step += 1
        ops.bump_param_epoch()

    def state_dict(self):
        d = super().state_dict()
        d["flat"] = {"step": self._step, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq}
        return d

    def load_state_dict(self, state_dict):
        flat = state_dict.pop("flat", None)
        super().load_state_dict(state_dict)
        if flat is not None:
            self._step = int(flat["step"])
            self.exp_avg.copy_(flat["exp_avg"])
            self.exp_avg_sq.copy_(flat["exp_avg_sq"])
