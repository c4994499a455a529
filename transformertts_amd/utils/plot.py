"""Validation-time figure writers with the reference's names, arguments and output paths (`utils/plot.py:11-151`), so
that `from utils.plot import ...` in the reference's `lightning_module.py:17-21` binds against this package.

Not on the training hot path (SURVEY.md section 8 marks plotting out of scope): host-side matplotlib only, imported
lazily, and a missing matplotlib makes each writer a logged no-op instead of breaking the training loop.  One grid
routine draws every figure; the five public functions only choose the panels, their captions and the file name.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence

import numpy as np
from torch import Tensor

_warned = False


def _pyplot():
    global _warned
    try:
        import matplotlib
        matplotlib.use("Agg", force=False)
        import matplotlib.pyplot as plt
        return plt
    except Exception:  # noqa: BLE001 - plotting must never take the training run down
        if not _warned:
            import logging
            logging.getLogger("transformertts_amd").warning("matplotlib is not importable: validation figures are skipped")
            _warned = True
        return None


def _host(t: Tensor) -> np.ndarray:
    return t.detach().float().cpu().numpy()


def _grid(panels: Sequence[Sequence[np.ndarray]], path: str, *, figsize, titles: Optional[Sequence[Sequence[str]]] = None,
          corner: Optional[Sequence[Sequence[tuple]]] = None, dpi: int = 300) -> Optional[str]:
    """Write a rows x cols grid of (time, channel) images, time on the horizontal axis and origin at the bottom."""
    plt = _pyplot()
    if plt is None:
        return None
    os.makedirs(os.path.dirname(path), exist_ok=True)
    rows, cols = len(panels), len(panels[0])
    fig, axes = plt.subplots(rows, cols, figsize=figsize, squeeze=False)
    for r in range(rows):
        for c in range(cols):
            ax = axes[r][c]
            ax.imshow(panels[r][c].T, aspect="auto", origin="lower")
            if titles is not None:
                ax.set_title(titles[r][c])
            if corner is not None:
                text, color = corner[r][c]
                ax.text(0.95, 0.95, text, transform=ax.transAxes, ha="right", va="top", fontsize=12, fontweight="bold",
                        color=color)
            ax.axis("off")
    fig.tight_layout()
    fig.savefig(path, dpi=dpi)
    plt.close(fig)
    return path


def _pair_grid(first: Tensor, second: Tensor, names, colors, path: str):
    """Up to 8 + 8 spectrograms (two groups, zero-padded to a common length) on a 4 x 4 sheet."""
    a, b = _host(first)[:8], _host(second)[:8]
    T = max(a.shape[1], b.shape[1])
    specs = [np.pad(x, ((0, T - x.shape[0]), (0, 0))) for x in a] + [np.pad(x, ((0, T - x.shape[0]), (0, 0))) for x in b]
    tags = [(f"{names[0]} {i + 1}", colors[0]) for i in range(len(a))] + [(f"{names[1]} {i + 1}", colors[1]) for i in range(len(b))]
    blank = np.zeros((T, a.shape[2]), dtype=a.dtype)
    while len(specs) < 16:                   # batches smaller than 8: leave the spare panels empty
        specs.append(blank)
        tags.append(("", "black"))
    return _grid([specs[r * 4:r * 4 + 4] for r in range(4)], path, figsize=(16, 10),
                 corner=[tags[r * 4:r * 4 + 4] for r in range(4)])


def plot_mels_batch(preds: Tensor, targets: Tensor, epoch: int, save_dir: str):
    """First 8 predicted vs ground-truth spectrograms -> <save_dir>/mels_batch/valid_epoch_<epoch>.png."""
    return _pair_grid(preds, targets, ("Pred", "GT"), ("blue", "red"),
                      os.path.join(save_dir, "mels_batch", f"valid_epoch_{epoch}.png"))


def plot_mels_scheduled(input_mels: Tensor, targets: Tensor, epoch: int, save_dir: str):
    """First 8 scheduled-sampling decoder inputs vs targets -> <save_dir>/mels_scheduled/scheduled_epoch_<epoch>.png."""
    return _pair_grid(input_mels, targets, ("Input", "Target"), ("green", "orange"),
                      os.path.join(save_dir, "mels_scheduled", f"scheduled_epoch_{epoch}.png"))


def plot_mels_single(pred: Tensor, target: Tensor, epoch: int, save_dir: str):
    """One inferred spectrogram above its ground truth -> <save_dir>/mels_single/infer_epoch_<epoch>.png."""
    return _grid([[_host(pred)], [_host(target)]], os.path.join(save_dir, "mels_single", f"infer_epoch_{epoch}.png"),
                 figsize=(10, 6), titles=[["Predicted"], ["Ground Truth"]])


def plot_alignments_batch(alignments: List[Tensor], epoch: int, save_dir: str, top_k: int = 4):
    """Head-averaged cross-attention of the first `top_k` utterances, one row per decoder layer
    -> <save_dir>/align_batch/valid_align_batch_epoch_<epoch>.png.  `alignments`: per layer (B, H, T_out, T_in)."""
    maps = [_host(a)[:top_k].mean(axis=1) for a in alignments]
    k = min(top_k, maps[0].shape[0])
    return _grid([[m[i] for i in range(k)] for m in maps],
                 os.path.join(save_dir, "align_batch", f"valid_align_batch_epoch_{epoch}.png"),
                 figsize=(4 * k, 3 * len(maps)),
                 titles=[[f"Layer {l + 1} - Sample {i + 1}" for i in range(k)] for l in range(len(maps))])


def plot_alignment_single(alignments: List[Tensor], idx: int, epoch: int, save_dir: str):
    """Every layer x head attention map of utterance `idx`
    -> <save_dir>/align_single/valid_align_<idx>_epoch_<epoch>.png."""
    maps = [_host(a)[idx] for a in alignments]
    heads = maps[0].shape[0]
    return _grid([[m[h] for h in range(heads)] for m in maps],
                 os.path.join(save_dir, "align_single", f"valid_align_{idx}_epoch_{epoch}.png"),
                 figsize=(4 * heads, 3 * len(maps)),
                 titles=[[f"Layer {l + 1} Head {h + 1}" for h in range(heads)] for l in range(len(maps))])
