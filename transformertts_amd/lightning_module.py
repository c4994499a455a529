"""LightningModule counterpart: the `training_step()` surface of the reference (`lightning_module.py:23-204`)
over the HIP-backed TransformerTTS.

Subclasses `pytorch_lightning.LightningModule` when Lightning is installed (so it drops into the reference's
`train.py`); otherwise a minimal stand-in base provides `device`, `current_epoch` and `log`, and `bench.py` /
`train_dp.py` drive `training_step` directly.  PNG plotting and loguru logging of the reference are out of scope
(SURVEY.md section 2); the per-step `.item()` of the reference (lightning_module.py:84) is kept behind
`config['training']['sync_loss_every_step']` (default True, as the reference) so benchmarks can state which.
"""
from __future__ import annotations

from typing import Any, Dict

import torch

from .loss import TransformerTTSLoss
from .model import TransformerTTS
from .utils.util import apply_teacher_forcing, get_noam_scheduler, get_teacher_forcing_ratio, prepare_batch

try:  # pragma: no cover - Lightning is absent from the build image
    import pytorch_lightning as pl
    _Base = pl.LightningModule
except Exception:  # noqa: BLE001
    class _Base(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.current_epoch = 0
            self._logged: Dict[str, float] = {}

        @property
        def device(self):
            return next(self.parameters()).device

        def log(self, name, value, **kwargs):
            self._logged[name] = float(value)


class LightningModule(_Base):
    def __init__(self, config: Dict[str, Any], exp_dir: str = None):
        super().__init__()
        self.model = TransformerTTS(**config['model'])
        self.criterion = TransformerTTSLoss(**config['loss'])
        self.config = config
        self.exp_dir = exp_dir
        self.train_losses = []
        self.valid_losses = []
        self.example_batch = None
        self.log_interval = config['training'].get('log_interval', 100)
        self.sync_loss = config['training'].get('sync_loss_every_step', True)
        # dropout / scheduled-sampling draws follow torch's process seed folded with the data-parallel rank (the reference
        # draws them from torch's global generator; replicas there differ because their generators advance differently).
        # Derived at the first training_step, not here: Lightning constructs the module before torch.distributed exists.
        # A (re-)seed of torch restarts the mask stream (ops.seeds.ensure_seeded sees it); building a module does not -- a
        # second module in the process (evaluation copy, EMA / teacher, load_from_checkpoint) must not replay the masks of step 0.

    def forward(self, phoneme, melspec, phoneme_lens, melspec_lens, **kwargs):
        return self.model(phoneme, melspec, phoneme_lens, melspec_lens, **kwargs)

    def teacher_forcing_ratio(self) -> float:
        """p_tf of the current epoch (lightning_module.py:62-67)."""
        return get_teacher_forcing_ratio(epoch=self.current_epoch + 1,
                                         total_epochs=self.config['training']['num_epochs'],
                                         mode=self.config['training']['teacher_forcing_mode'], cycles=1)

    def training_step(self, batch, batch_idx):
        from . import ops
        ops.seeds.ensure_seeded()
        phoneme, melspec, phoneme_lens, melspec_lens = prepare_batch(batch, self.device)
        # Both forwards encode the same phonemes: one pass over a batch of 2 B (model.encode_twin) hands each its encoder memory
        mem_grad = mem_nograd = None
        if self.model.twin_encode_ok(phoneme):
            mem_grad, mem_nograd = self.model.encode_twin(phoneme, phoneme_lens)
        # ... and both run the post-net, the no-grad forward for its BatchNorm running statistics alone (its post_melspec has no
        # reader): one pass over both predictions, made by the grad forward (ops.PostnetTwin)
        post_twin = ops.PostnetTwin() if self.model.twin_postnet_ok(melspec) else None
        # forward #1 (no grad, train mode: dropout on, BN statistics updated) -> the model's own prediction
        with torch.no_grad():   # only pred_melspec is used: do not materialise the attention maps
            pred_melspec = self.forward(phoneme, melspec, phoneme_lens, melspec_lens, need_alignments=False, need_stop=False,
                                        memory=mem_nograd, postnet_twin=post_twin)['pred_melspec']
        p_tf = self.teacher_forcing_ratio()
        mel_mixed = apply_teacher_forcing(pred_melspec, melspec, melspec_lens, p_tf, self.device)
        # forward #2 (with grad) on the mixed input, loss against the ground truth.  The loss reads the three prediction tensors
        # only (lightning_module.py:78-79 of the reference discards the alignments of its output dict as well), so the per-head
        # attention maps are not written here either; `validation_step` / `forward()` return them as the reference does.
        output = self.forward(phoneme, mel_mixed, phoneme_lens, melspec_lens,
                              need_alignments=self.config['training'].get('train_step_alignments', False), memory=mem_grad,
                              postnet_twin=post_twin)
        loss = self.criterion(output, melspec, melspec_lens)
        if self.sync_loss:
            self.train_losses.append(loss['total'].item())
        else:
            self.train_losses.append(loss['total'].detach())
        return loss['total']

    def on_train_epoch_end(self):
        self.train_losses.clear()

    def validation_step(self, batch, batch_idx):
        if self.example_batch is None:
            self.example_batch = batch
        phoneme, melspec, phoneme_lens, melspec_lens = prepare_batch(batch, self.device)
        output = self.forward(phoneme, melspec, phoneme_lens, melspec_lens)
        loss = self.criterion(output, melspec, melspec_lens)
        self.valid_losses.append(loss['total'].item())
        return loss['total']

    def on_validation_epoch_end(self):
        if self.valid_losses:
            self.log('val_loss', sum(self.valid_losses) / len(self.valid_losses), on_epoch=True)
        self.valid_losses.clear()

    def configure_optimizers(self):
        # same Adam arithmetic as the reference (lightning_module.py:160-163), fused over flat buffers; `fused_clip_norm`
        # (> 0) folds the Trainer's gradient_clip_val (train.py:41) into the step for loops that do not clip themselves.
        # FlatAdam raises for parameters that are not on the HIP device: there is no CPU optimizer path.
        from .optim import FlatAdam
        optimizer = FlatAdam(self.parameters(), lr=1.0, betas=(0.9, 0.98), eps=1e-9,
                             max_grad_norm=float(self.config['training'].get('fused_clip_norm', 0.0)))
        lr_lambda = get_noam_scheduler(d_model=self.config['model']['d_model'],
                                       warmup_steps=self.config['training']['warmup_steps'])
        scheduler = torch.optim.lr_scheduler.LambdaLR(optimizer, lr_lambda=lr_lambda)
        return {'optimizer': optimizer, 'lr_scheduler': {'scheduler': scheduler, 'interval': 'step', 'frequency': 1}}
