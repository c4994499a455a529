"""oracle/collate.py -- TEST INFRASTRUCTURE ONLY.  CPU restatement of the reference's input side.

`oracle_getitem` follows dataset.py:59-69 (`melspec` transposed to (T, n_mels), `sequence` -> 'phoneme');
`oracle_collate` follows dataset.py:71-103 (sort by phoneme length descending with torch.argsort, zero padding
to the batch maxima, lengths in sorted order).  Pinned by tests/golden/collate.npz, produced by the reference's own
`collate_fn` on `oracle.synth.synth_samples` (tests/golden/make_golden.py::golden_collate).
"""
from __future__ import annotations

from typing import Any, Dict, List

import numpy as np
import torch


def oracle_getitem(stored: Dict[str, Any]) -> Dict[str, Any]:
    return {"transcript": str(stored["transcript"]),
            "melspec": torch.from_numpy(np.asarray(stored["melspec"])).T,
            "phoneme": torch.from_numpy(np.asarray(stored["sequence"]))}


def oracle_collate(items: List[Dict[str, Any]]) -> Dict[str, Any]:
    plen = torch.tensor([int(it["phoneme"].shape[0]) for it in items])
    order = torch.argsort(plen, descending=True)                       # dataset.py:65
    items = [items[int(i)] for i in order]
    B = len(items)
    Pmax = int(plen[order[0]])
    Tmax = max(int(it["melspec"].shape[0]) for it in items)
    n_mels = int(items[0]["melspec"].shape[1])
    ph = np.zeros((B, Pmax), dtype=np.int64)
    mel = np.zeros((B, Tmax, n_mels), dtype=np.float32)
    mlen = np.zeros((B,), dtype=np.int64)
    for r, it in enumerate(items):
        p, m = int(it["phoneme"].shape[0]), int(it["melspec"].shape[0])
        ph[r, :p] = it["phoneme"].numpy()
        mel[r, :m, :] = it["melspec"].numpy()
        mlen[r] = m
    return {"phoneme": torch.from_numpy(ph), "melspec": torch.from_numpy(mel), "phoneme_lens": plen[order],
            "melspec_lens": torch.from_numpy(mlen), "transcript": [it["transcript"] for it in items]}
