"""TransformerTTSLoss -- masked MSE (pred + 0.5 * post) + stop-gate BCE-with-logits (pos_weight).

Same constructor, call signature, returned dict and `pos_weight` buffer as the reference's `loss.py:8-55`.
The whole loss (and its backward) runs in the fused kernels of csrc/loss.hip: one
streaming masked reduction instead of the reference's boolean-index gathers (`mel[mask]`, loss.py:34-36,44), which
allocate data-dependent shapes and force a device->host sync.  HIP tensors only.
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn as nn
from torch import Tensor


class TransformerTTSLoss(nn.Module):
    def __init__(self, stop_weight: float = 8.0):
        super().__init__()
        self.register_buffer("pos_weight", torch.tensor(stop_weight))
        self._pos_weight_host = float(stop_weight)   # host copy: reading the device buffer every step would synchronise

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)
        if prefix + "pos_weight" in state_dict:
            self._pos_weight_host = float(state_dict[prefix + "pos_weight"])

    def forward(self, outputs: Dict[str, Tensor], mel: Tensor, lengths: Tensor) -> Dict[str, Tensor]:
        from . import ops          # rejects non-HIP tensors: there is no CPU path (the CPU restatement is oracle/)
        pred, post, stop = outputs["pred_melspec"], outputs["post_melspec"], outputs["pred_stop"]
        total, pred_mel, post_mel, stop_l = ops.TTSLossFn.apply(pred, post, stop, mel, lengths.to(torch.int64),
                                                                self._pos_weight_host)
        return {"total": total, "pred_mel": pred_mel, "post_mel": post_mel, "stop": stop_l}
