"""TransformerTTSLoss -- masked MSE (pred + 0.5 * post) + stop-gate BCE-with-logits (pos_weight).

Same constructor, call signature, returned dict and `pos_weight` buffer as the reference's `loss.py:8-55`.
Restated without boolean-index gathers (`mel[mask]`, loss.py:34-36,44), which allocate data-dependent
shapes and force a device->host sync every step: the masked means are computed as masked sums over the
padded tensors, which is the same arithmetic up to fp32 summation order.
SURVEY.md section 8f ranks a fused HIP kernel for this as the first "next" item; until then these are
stock torch element-wise ops on the device (they sit outside the model hot path).
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch import Tensor


class TransformerTTSLoss(nn.Module):
    def __init__(self, stop_weight: float = 8.0):
        super().__init__()
        self.register_buffer("pos_weight", torch.tensor(stop_weight))

    def forward(self, outputs: Dict[str, Tensor], mel: Tensor, lengths: Tensor) -> Dict[str, Tensor]:
        pred, post, stop = outputs["pred_melspec"], outputs["post_melspec"], outputs["pred_stop"]
        B, T, C = pred.shape
        pos = torch.arange(T, device=pred.device).unsqueeze(0)
        valid = (pos < lengths.unsqueeze(1)).to(pred.dtype)                 # (B,T)
        gate = (pos == (lengths.unsqueeze(1) - 1)).to(pred.dtype)
        n_frames = valid.sum()
        vm = valid.unsqueeze(-1)
        pred_mel_loss = (((pred - mel) ** 2) * vm).sum() / (n_frames * C)
        post_mel_loss = (((post - mel) ** 2) * vm).sum() / (n_frames * C)
        bce = F.binary_cross_entropy_with_logits(stop, gate, reduction='none', pos_weight=self.pos_weight)
        stop_loss = (bce * valid).sum() / n_frames
        mel_loss = pred_mel_loss + 0.5 * post_mel_loss
        return {"total": mel_loss + stop_loss, "pred_mel": pred_mel_loss, "post_mel": post_mel_loss,
                "stop": stop_loss}
