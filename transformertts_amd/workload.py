"""Named model configurations and synthetic LJSpeech-shaped batches for the benchmark and the launchers.

`model_config(name)` returns the `model:` section the reference splats into its constructor
(`TransformerTTS(**config['model'])`, lightning_module.py:30): "base" is config.yaml:25-42, "scaled" is
BASELINE.json configs[4] (d_model 512, 6+6 layers, 8 heads, d_ffn 2048), "tiny" a small stack for smoke runs.
`synth_batch` follows SURVEY.md section 8d and honours the output contract of the reference's `collate_fn`
(dataset.py:71-103): int64 ids / fp32 mels zero-padded to the batch maxima, rows sorted by phoneme length descending.
"""
from __future__ import annotations

from typing import Dict

import torch


def _cfg(d, heads, layers, dffn, prenet_layers=3, post_layers=5, n_phon=100, n_mels=80):
    return dict(encoder_prenet_n_layers=prenet_layers, encoder_prenet_in_channel=d, encoder_prenet_out_channel=d,
                encoder_prenet_kernel_size=5, encoder_prenet_dropout=0.5, encoder_n_layers=layers, encoder_n_head=heads,
                encoder_d_ffn=dffn, encoder_dropout=0.1, decoder_n_layers=layers, decoder_n_head=heads,
                decoder_d_ffn=dffn, decoder_dropout=0.1, postnet_n_layers=post_layers, postnet_kernel_size=5,
                postnet_dropout=0.5, d_model=d, n_phon=n_phon, n_mels=n_mels)


CONFIGS = {
    "base": _cfg(256, 4, 3, 1024),
    "scaled": _cfg(512, 8, 6, 2048),
    "tiny": dict(_cfg(128, 2, 1, 256, prenet_layers=2, post_layers=3, n_phon=30, n_mels=16), decoder_n_layers=2),
    "micro": _cfg(32, 2, 1, 64, prenet_layers=2, post_layers=3, n_phon=20, n_mels=16),      # head_dim 16 (SURVEY 8c's tiny golden)
    "tiny1h": dict(_cfg(128, 1, 1, 256, prenet_layers=2, post_layers=3, n_phon=30, n_mels=16), decoder_n_layers=2),   # head_dim 128
}


def model_config(name: str = "base") -> dict:
    return dict(CONFIGS[name])


def synth_batch(B: int, Tp: int = 100, Tm: int = 870, n_mels: int = 80, n_phon: int = 100, ragged: bool = False,
                seed: int = 1234) -> Dict[str, torch.Tensor]:
    """Dense: every utterance has Tp phonemes and Tm frames.  Ragged: frame counts ~ N(566, 170) scaled to Tm and clipped
    to [95, Tm] with one utterance at Tm, phoneme counts tied to them; padding is id 0 / 0.0 (dataset.py:83-84)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    if ragged:
        ml = torch.round(566.0 / 870.0 * Tm + 170.0 / 870.0 * Tm * torch.randn(B, generator=g))
        ml = torch.clamp(ml, min(95, Tm), Tm).long()
        ml[0] = Tm
        pl = torch.round(ml.float() * (Tp / Tm) * 0.87 + 4.0 * torch.randn(B, generator=g) * (Tp / 100.0))
        pl = torch.clamp(pl, min(8, Tp), Tp).long()
        pl[0] = Tp
    else:
        ml = torch.full((B,), Tm, dtype=torch.long)
        pl = torch.full((B,), Tp, dtype=torch.long)
    order = torch.argsort(pl, descending=True, stable=True)
    pl, ml = pl[order], ml[order]
    tp, tm = int(pl.max()), int(ml.max())
    ph = torch.randint(0, n_phon, (B, tp), generator=g, dtype=torch.long)
    mel = torch.randn(B, tm, n_mels, generator=g)
    ph = ph * (torch.arange(tp).unsqueeze(0) < pl.unsqueeze(1))
    mel = mel * (torch.arange(tm).unsqueeze(0) < ml.unsqueeze(1)).unsqueeze(-1)
    return {"phoneme": ph.contiguous(), "melspec": mel.contiguous(), "phoneme_lens": pl.contiguous(),
            "melspec_lens": ml.contiguous()}
