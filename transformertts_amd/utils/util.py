"""Step-surface helpers with the reference's names and semantics (`utils/util.py:42-120`): Noam schedule,
teacher-forcing ratio, block-wise scheduled-sampling mix, batch placement.  Pure host logic plus a handful of
element-wise torch ops on the device; experiment-directory and loguru plumbing of the reference is out of
scope."""
from __future__ import annotations

import math
from typing import Dict, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor


def get_device() -> torch.device:
    return torch.device('cuda' if torch.cuda.is_available() else 'cpu')


def get_noam_scheduler(d_model: int, warmup_steps: int):
    """lr multiplier d^-0.5 * min(s^-0.5, s * warmup^-1.5), step clamped to >= 1."""
    def lr_lambda(step):
        step = max(step, 1)
        return (d_model ** -0.5) * min(step ** -0.5, step * (warmup_steps ** -1.5))
    return lr_lambda


def get_teacher_forcing_ratio(epoch: int, total_epochs: int = 300, mode: str = "cosine", warmup_epochs: int = 10,
                              **kwargs) -> float:
    """1.0 during warm-up, then 'cosine' (floor 0.5), 'linear' (floor 0.05) or 'constant'."""
    if epoch < warmup_epochs:
        return 1.0
    e = epoch - warmup_epochs
    span = max(total_epochs - warmup_epochs, 1)
    if mode == "cosine":
        ratio = 0.5 * math.cos(math.pi * e * kwargs.get("cycles", 1) / span) + 0.5
        return max(min(ratio, 1.0), 0.5)
    if mode == "linear":
        return max(1.0 - e / span, 0.05)
    if mode == "constant":
        return kwargs.get("value", 1.0)
    raise ValueError(f"Unsupported teacher forcing mode: {mode}")


def prepare_batch(batch: Dict[str, Tensor], device) -> Tuple[Tensor, ...]:
    return tuple(batch[k].to(device, non_blocking=True) for k in ['phoneme', 'melspec', 'phoneme_lens', 'melspec_lens'])


def block_mask(mel: Tensor, p_tf: float, L_bar: int) -> Tensor:
    """(B,T,1) bool: frames replaced by the model's own prediction; a Bernoulli(1-p_tf) seed per frame dilated to
    blocks of about L_bar frames with a max-pool."""
    B, T, _ = mel.shape
    seed = (torch.rand(B, 1, T, device=mel.device) < (1 - p_tf)).float()
    dilated = F.max_pool1d(seed, kernel_size=L_bar, stride=1, padding=L_bar // 2)
    return dilated.squeeze(1).bool().unsqueeze(-1)[:, :T, :]


def apply_teacher_forcing(pred_melspec: Tensor, melspec: Tensor, melspec_lens: Tensor, p_tf: float, device=None) -> Tensor:
    if pred_melspec.is_cuda:   # fused HIP kernel; the uniform draw stays torch.rand on the device, as in the reference
        from .. import ops
        B, T, _ = pred_melspec.shape
        u = torch.rand(B, 1, T, device=pred_melspec.device)
        return ops.sched_sampling_mix(pred_melspec.detach(), melspec, u.view(B, T), melspec_lens.to(torch.int64), p_tf, 8)
    mask = block_mask(pred_melspec, p_tf, L_bar=8)
    mel_mixed = torch.where(mask, pred_melspec.detach(), melspec)
    valid = torch.arange(pred_melspec.size(1), device=pred_melspec.device).unsqueeze(0) < melspec_lens.unsqueeze(1)
    return mel_mixed * valid.unsqueeze(-1)
