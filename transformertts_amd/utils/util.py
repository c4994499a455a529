"""Step-surface helpers with the reference's names and semantics (`utils/util.py:42-120`): Noam schedule,
teacher-forcing ratio, block-wise scheduled-sampling mix (a HIP kernel), batch placement.  Experiment-directory and
loguru plumbing of the reference is out of scope.  No CPU arithmetic lives here: the
CPU restatement used by the tests is oracle/ref_model.py."""
from __future__ import annotations

import math
from typing import Dict, Tuple

import torch
from torch import Tensor


def increment_path(base_path: str) -> str:
    """Next free experiment directory `exp_<n>_<MMDD-HHMM>` under `base_path`, with the sub-directories the reference's
    plot helpers write into (utils/util.py:18-35).  Host bookkeeping only; kept so that the reference's `train.py` binds
    against this package without edits."""
    import os
    from datetime import datetime, timedelta, timezone
    os.makedirs(base_path, exist_ok=True)
    stamp = datetime.now(timezone(timedelta(hours=9))).strftime('%m%d-%H%M')      # the reference stamps in KST
    taken = os.listdir(base_path)
    n = 1
    while any(name.startswith(f"exp_{n}") for name in taken):
        n += 1
    path = os.path.join(base_path, f"exp_{n}_{stamp}")
    for sub in ('mels_batch', 'mels_single', 'align_batch', 'align_single', 'mels_scheduled'):
        os.makedirs(os.path.join(path, sub), exist_ok=True)
    return path


def setup_logger(log_path: str = None):
    """Console (and optional file) logging as the reference configures it (utils/util.py:123-132): loguru when it is
    installed, the standard `logging` module otherwise."""
    import os
    import sys
    try:
        from loguru import logger
        logger.remove()
        logger.add(sys.stdout, level="INFO")
        if log_path:
            os.makedirs(os.path.dirname(log_path) or ".", exist_ok=True)
            logger.add(log_path, level="DEBUG", rotation="10 MB")
        return logger
    except ImportError:
        import logging
        handlers = [logging.StreamHandler(sys.stdout)]
        if log_path:
            os.makedirs(os.path.dirname(log_path) or ".", exist_ok=True)
            handlers.append(logging.FileHandler(log_path))
        logging.basicConfig(level=logging.INFO, format="%(asctime)s | %(levelname)s | %(message)s", handlers=handlers, force=True)
        return logging.getLogger("transformertts_amd")


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


# Test seam: a callable (B, T, device) -> (B, T) uniform tensor that replaces the in-kernel draw, so that parity tests
# can inject the reference's own `torch.rand` values (tests/test_hip_model.py::test_training_step_surface).
_uniform_draw = None


def block_mask(mel: Tensor, p_tf: float, L_bar: int) -> Tensor:
    """(B,T,1) bool mask of the frames that take the model's own prediction (utils/util.py:103-111): frame t is set when
    any uniform draw in its `L_bar`-frame window falls below 1 - p_tf.  Same kernel as `apply_teacher_forcing`, asked
    to mix a plane of ones into a plane of zeros; the training step itself never materialises the mask."""
    from .. import ops
    B, T, _ = mel.shape
    one = torch.ones(B, T, 4, dtype=torch.float32, device=mel.device)
    u = _uniform_draw(B, T, mel.device).reshape(B, T).contiguous() if _uniform_draw is not None else None
    full = torch.full((B,), T, dtype=torch.int64, device=mel.device)
    mix = ops.sched_sampling_mix(one, torch.zeros_like(one), u, full, p_tf, L_bar, seed=ops.seeds.next())
    return mix[:, :, :1] > 0.5


def apply_teacher_forcing(pred_melspec: Tensor, melspec: Tensor, melspec_lens: Tensor, p_tf: float, device=None) -> Tensor:
    """`block_mask` + `apply_teacher_forcing` of the reference (utils/util.py:103-120) as ONE fused HIP kernel: frame t
    takes the model's own prediction when any uniform draw in its 8-frame window falls below 1 - p_tf, the ground truth
    otherwise, and zero beyond the utterance.  The draw (the reference's `torch.rand(B,1,T)`) is generated inside the
    kernel from a counter-based hash -- no separate RNG launch, and replayable from a captured HIP graph."""
    from .. import ops
    B, T, _ = pred_melspec.shape
    u = _uniform_draw(B, T, pred_melspec.device).reshape(B, T).contiguous() if _uniform_draw is not None else None
    return ops.sched_sampling_mix(pred_melspec.detach(), melspec, u, melspec_lens.to(torch.int64), p_tf, 8,
                                  seed=ops.seeds.next())
