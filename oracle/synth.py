"""Synthetic LJSpeech-shaped batches (test infrastructure).

Honours the output contract of the reference `collate_fn` (dataset.py:71-103):
`phoneme` int64 (B,Tp) zero-padded, `melspec` fp32 (B,Tm,n_mels) zero-padded,
`phoneme_lens` / `melspec_lens` int64 (B,), rows sorted by phoneme length descending,
Tp / Tm = the batch maxima.
"""
from __future__ import annotations

from typing import Dict

import torch


def synth_batch(B: int, Tp: int = 100, Tm: int = 870, n_mels: int = 80, n_phon: int = 100,
                ragged: bool = False, seed: int = 1234) -> Dict[str, torch.Tensor]:
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    if ragged:
        ml = torch.clamp(torch.round(566.0 / 870.0 * Tm + 170.0 / 870.0 * Tm * torch.randn(B, generator=g)),
                         min(95, Tm), Tm).long()
        ml[0] = Tm
        pl = torch.clamp(torch.round(ml.float() * (Tp / Tm) * 0.87 + 4.0 * torch.randn(B, generator=g) * (Tp / 100.0)),
                         min(8, Tp), Tp).long()
        pl[0] = Tp
    else:
        ml = torch.full((B,), Tm, dtype=torch.long)
        pl = torch.full((B,), Tp, dtype=torch.long)
    order = torch.argsort(pl, descending=True, stable=True)
    pl, ml = pl[order], ml[order]
    Tp_, Tm_ = int(pl.max()), int(ml.max())
    ph = torch.randint(0, n_phon, (B, Tp_), generator=g, dtype=torch.long)
    mel = torch.randn(B, Tm_, n_mels, generator=g)
    ph = ph * (torch.arange(Tp_).unsqueeze(0) < pl.unsqueeze(1))
    mel = mel * (torch.arange(Tm_).unsqueeze(0) < ml.unsqueeze(1)).unsqueeze(-1)
    return {"phoneme": ph.contiguous(), "melspec": mel.contiguous(),
            "phoneme_lens": pl.contiguous(), "melspec_lens": ml.contiguous()}


def synth_samples(n: int, n_mels: int = 80, n_phon: int = 100, max_frames: int = 200, seed: int = 77):
    """`n` utterances as the reference's preprocess step stores them (preprocess.py:36-42): `melspec` fp32
    (n_mels, T), `sequence` int64 ids, `transcript` str.  Phoneme lengths repeat on purpose (ties in the sort of
    dataset.py:65), one utterance is the longest in frames but not in phonemes."""
    import numpy as np
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        T = int(rng.integers(5, max_frames + 1))
        P = int(rng.integers(3, 12)) if i % 3 else 7
        if i == 1:
            T = max_frames + 3
        out.append({"melspec": rng.standard_normal((n_mels, T)).astype(np.float32),
                    "sequence": rng.integers(1, n_phon, size=P).astype(np.int64),
                    "transcript": f"utterance {i}"})
    return out
