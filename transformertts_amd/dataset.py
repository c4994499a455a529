"""Input side of the training step -- counterpart of the reference's `dataset.py`.

Same surface: `TransformerTTSDataset(config, mode)` over the `.npz` files `preprocess.py:36-42` writes
(`melspec (n_mels, T)` fp32, `sequence` int ids, `transcript`), `collate_fn(batch)` returning the reference's dict
(`dataset.py:61-103`: sort by phoneme length descending, zero padding), `DataModule`.

MI355X-side additions, all optional and result-identical:

* `collate_ragged` + `DeviceStager`: the host only concatenates the utterances *as they lie on disk* into pinned
  memory (no per-sample transpose, no padding bytes over PCIe); the H2D copy runs on a side HIP stream and the
  `ttts_collate_*` kernels write the padded `(B, Tmax, n_mels)` / `(B, Pmax)` tensors in HBM.  The consumer stream
  only waits on an event, so staging overlaps the previous step.
* `DevicePrefetcher`: wraps any iterable of ragged / padded batches and keeps `depth` batches in flight.
* `BucketBatchSampler`: length-bucketed batches (less padding); changes batch composition, hence opt-in.
"""
from __future__ import annotations

import os
from typing import Any, Dict, Iterable, Iterator, List, Literal, Optional, Sequence

import numpy as np
import torch
from torch.utils.data import Dataset, Sampler

VALID_PREFIXES = ('LJ001', 'LJ002', 'LJ003')     # dataset.py:52


class TransformerTTSDataset(Dataset):
    """`dataset.py:44-69`: every `.npz` under config['path']['preprocessed']; LJ001-3 are the validation split."""

    def __init__(self, config: Dict, mode: Literal['train', 'valid'] = 'train'):
        super().__init__()
        self.data_dir = config['path']['preprocessed']
        self.mode = mode
        names = sorted(f for f in os.listdir(self.data_dir) if f.endswith('.npz'))
        self.data_list = [f for f in names if f.startswith(VALID_PREFIXES) == (mode == 'valid')]
        self._mel_lens: Optional[List[int]] = None

    def __len__(self):
        return len(self.data_list)

    def __getitem__(self, idx):
        data = np.load(os.path.join(self.data_dir, self.data_list[idx]), allow_pickle=True)
        return {
            'transcript': str(data['transcript']),
            'melspec': torch.from_numpy(data['melspec']).T,      # (T, n_mels) view of the stored (n_mels, T)
            'phoneme': torch.from_numpy(data['sequence']),
        }

    def mel_lengths(self) -> List[int]:
        """Frames per utterance (one pass over the files, cached; used by BucketBatchSampler)."""
        if self._mel_lens is None:
            lens = []
            for f in self.data_list:
                with np.load(os.path.join(self.data_dir, f), allow_pickle=True) as d:
                    lens.append(int(d['melspec'].shape[1]))
            self._mel_lens = lens
        return self._mel_lens


def _sorted_batch(batch: Sequence[Dict[str, Any]]):
    # dataset.py:64-66 -- same torch call, so ties resolve exactly as in the reference
    phoneme_lens = torch.tensor([len(s['phoneme']) for s in batch])
    order = torch.argsort(phoneme_lens, descending=True)
    return [batch[i] for i in order], phoneme_lens[order]


def _pinned(shape, dtype) -> torch.Tensor:
    # page-locked when collating in the main process; DataLoader workers return pageable tensors (they must not
    # touch the device) and the loader's pin_memory thread locks them afterwards
    pin = torch.utils.data.get_worker_info() is None and torch.cuda.is_available()
    return torch.zeros(shape, dtype=dtype, pin_memory=pin)


def collate_fn(batch: Sequence[Dict[str, Any]]) -> Dict[str, Any]:
    """Reference-identical padded batch on the host (`dataset.py:61-103`)."""
    batch, phoneme_lens = _sorted_batch(batch)
    B = len(batch)
    max_in = int(phoneme_lens[0]) if B else 0
    max_out = max((s['melspec'].shape[0] for s in batch), default=0)
    n_mels = batch[0]['melspec'].shape[1] if B else 0
    phoneme = torch.zeros(B, max_in, dtype=torch.long)
    melspec = torch.zeros(B, max_out, n_mels)
    melspec_lens = torch.zeros(B, dtype=torch.long)
    for i, s in enumerate(batch):
        p, m = len(s['phoneme']), s['melspec'].shape[0]
        phoneme[i, :p] = s['phoneme']
        melspec[i, :m] = s['melspec']
        melspec_lens[i] = m
    return {'phoneme': phoneme, 'melspec': melspec, 'phoneme_lens': phoneme_lens, 'melspec_lens': melspec_lens,
            'transcript': [s['transcript'] for s in batch]}


def collate_ragged(batch: Sequence[Dict[str, Any]]) -> Dict[str, Any]:
    """Same ordering as `collate_fn`, but no padding and no transpose on the host: utterances are laid back to back
    in their stored `(n_mels, T)` layout in (pinned) host memory.  `DeviceStager.stage` turns this into the padded
    device batch."""
    batch, phoneme_lens = _sorted_batch(batch)
    B = len(batch)
    n_mels = batch[0]['melspec'].shape[1] if B else 0
    mel_lens = torch.tensor([s['melspec'].shape[0] for s in batch], dtype=torch.long)
    frame_off = torch.zeros(B + 1, dtype=torch.long)
    phon_off = torch.zeros(B + 1, dtype=torch.long)
    if B:
        frame_off[1:] = torch.cumsum(mel_lens, 0)
        phon_off[1:] = torch.cumsum(phoneme_lens.to(torch.long), 0)
    mel = _pinned(int(frame_off[-1]) * n_mels, torch.float32)
    ids = _pinned(int(phon_off[-1]), torch.long)
    mel_np, ids_np = mel.numpy(), ids.numpy()
    for i, s in enumerate(batch):
        a, b = int(frame_off[i]) * n_mels, int(frame_off[i + 1]) * n_mels
        # s['melspec'] is the (T, n_mels) transposed view; its .T is the stored contiguous (n_mels, T) array
        mel_np[a:b] = s['melspec'].T.numpy().reshape(-1)
        ids_np[int(phon_off[i]):int(phon_off[i + 1])] = s['phoneme'].numpy()
    return {'ragged': True, 'mel_ragged': mel, 'phoneme_ragged': ids, 'frame_offsets': frame_off,
            'phoneme_offsets': phon_off, 'phoneme_lens': phoneme_lens, 'melspec_lens': mel_lens, 'n_mels': n_mels,
            'transcript': [s['transcript'] for s in batch]}


class DeviceStager:
    """Moves one batch to the HIP device on a side stream and (for ragged batches) pads it there.

    stage(batch) -> dict with the reference's keys on the device plus '_ready' (a recorded event);
    `wait(dev_batch)` makes the current stream wait for it.  No host synchronisation anywhere.
    """

    def __init__(self, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError("DeviceStager needs the HIP device (no CPU fallback)")
        self.device = torch.device(device if device is not None else 'cuda')
        self.stream = torch.cuda.Stream(device=self.device)

    def stage(self, batch: Dict[str, Any]) -> Dict[str, Any]:
        from . import _lib
        from .ops import _p
        lib = _lib.load()
        dev = self.device
        with torch.cuda.stream(self.stream):
            out: Dict[str, Any] = {'transcript': batch.get('transcript')}
            out['phoneme_lens'] = batch['phoneme_lens'].to(dev, non_blocking=True)
            out['melspec_lens'] = batch['melspec_lens'].to(dev, non_blocking=True)
            if batch.get('ragged'):
                B = int(batch['melspec_lens'].numel())
                n_mels = int(batch['n_mels'])
                Tmax = int(batch['melspec_lens'].max()) if B else 0      # host tensors: no device sync
                Pmax = int(batch['phoneme_lens'].max()) if B else 0
                mel_r = batch['mel_ragged'].to(dev, non_blocking=True)
                ids_r = batch['phoneme_ragged'].to(dev, non_blocking=True)
                f_off = batch['frame_offsets'].to(dev, non_blocking=True)
                p_off = batch['phoneme_offsets'].to(dev, non_blocking=True)
                mel = torch.empty(B, Tmax, n_mels, dtype=torch.float32, device=dev)
                ids = torch.empty(B, Pmax, dtype=torch.long, device=dev)
                s = self.stream.cuda_stream
                _lib.check(lib.ttts_collate_melspec(_p(mel_r), _p(f_off), _p(mel), B, Tmax, n_mels, s), "ttts_collate_melspec")
                _lib.check(lib.ttts_collate_phoneme(_p(ids_r), _p(p_off), _p(ids), B, Pmax, s), "ttts_collate_phoneme")
                out['melspec'], out['phoneme'] = mel, ids
                out['_keep'] = (mel_r, ids_r, f_off, p_off, batch)      # alive until the consumer has waited
            else:
                out['melspec'] = batch['melspec'].to(dev, non_blocking=True)
                out['phoneme'] = batch['phoneme'].to(dev, non_blocking=True)
                out['_keep'] = (batch,)
            ev = torch.cuda.Event()
            ev.record(self.stream)
            out['_ready'] = ev
        return out

    @staticmethod
    def wait(dev_batch: Dict[str, Any]) -> Dict[str, Any]:
        cur = torch.cuda.current_stream()
        cur.wait_event(dev_batch['_ready'])
        for k in ('melspec', 'phoneme', 'phoneme_lens', 'melspec_lens'):
            dev_batch[k].record_stream(cur)      # allocated on the side stream, consumed on this one
        for t in dev_batch.get('_keep', ())[:4]:
            if isinstance(t, torch.Tensor) and t.is_cuda:
                t.record_stream(cur)
        dev_batch.pop('_keep', None)
        return dev_batch


class DevicePrefetcher:
    """Iterates host batches (padded or ragged) and yields device batches, `depth` of them staged ahead."""

    def __init__(self, loader: Iterable[Dict[str, Any]], device=None, depth: int = 2):
        self.loader, self.stager, self.depth = loader, DeviceStager(device), max(int(depth), 1)

    def __iter__(self) -> Iterator[Dict[str, Any]]:
        it = iter(self.loader)
        queue: List[Dict[str, Any]] = []
        done = False
        while True:
            while not done and len(queue) < self.depth:
                try:
                    queue.append(self.stager.stage(next(it)))
                except StopIteration:
                    done = True
            if not queue:
                return
            yield DeviceStager.wait(queue.pop(0))


class BucketBatchSampler(Sampler):
    """Batches of utterances of similar length: sort a shuffled pool of `bucket_batches * batch_size` indices by
    mel length, cut it into batches, shuffle the batches.  Deterministic in (seed, epoch); with world_size > 1 each
    rank takes every world_size-th batch (same number of batches per rank, remainder dropped)."""

    def __init__(self, lengths: Sequence[int], batch_size: int, bucket_batches: int = 16, shuffle: bool = True,
                 drop_last: bool = True, seed: int = 0, rank: int = 0, world_size: int = 1):
        if batch_size < 1 or bucket_batches < 1:
            raise ValueError("batch_size and bucket_batches must be >= 1")
        if not 0 <= rank < world_size:
            raise ValueError("need 0 <= rank < world_size")
        self.lengths = np.asarray(lengths, dtype=np.int64)
        self.batch_size, self.bucket_batches = batch_size, bucket_batches
        self.shuffle, self.drop_last, self.seed = shuffle, drop_last, seed
        self.rank, self.world_size, self.epoch = rank, world_size, 0

    def set_epoch(self, epoch: int):
        self.epoch = int(epoch)

    def _batches(self) -> List[List[int]]:
        n = len(self.lengths)
        rng = np.random.default_rng([self.seed, self.epoch])
        order = rng.permutation(n) if self.shuffle else np.arange(n)
        pool = self.batch_size * self.bucket_batches
        batches: List[List[int]] = []
        for a in range(0, n, pool):
            chunk = order[a:a + pool]
            chunk = chunk[np.argsort(-self.lengths[chunk], kind='stable')]
            for b in range(0, len(chunk), self.batch_size):
                idx = chunk[b:b + self.batch_size]
                if len(idx) == self.batch_size or not self.drop_last:
                    batches.append([int(i) for i in idx])
        if self.shuffle:
            batches = [batches[i] for i in rng.permutation(len(batches))]
        per_rank = len(batches) // self.world_size
        return batches[self.rank:per_rank * self.world_size:self.world_size]

    def __iter__(self):
        return iter(self._batches())

    def __len__(self):
        return len(self._batches())

    def padding_fraction(self) -> float:
        """Padded frames / total frames over one epoch of this sampler (diagnostic)."""
        pad = tot = 0
        for b in self._batches():
            l = self.lengths[b]
            tot += int(l.max()) * len(b)
            pad += int(l.max()) * len(b) - int(l.sum())
        return pad / max(tot, 1)


try:                                    # pragma: no cover - depends on the installed stack
    import pytorch_lightning as _pl
    _DMBase = _pl.LightningDataModule
except Exception:                       # noqa: BLE001 - Lightning is absent in the build image
    _DMBase = object


class DataModule(_DMBase):
    """`dataset.py:9-41` with two extra config keys under 'training': `ragged` (device-side padding) and
    `bucket_batches` (0 = the reference's plain shuffling)."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        t = config['training']
        self.batch_size = t.get('batch_size', 1)
        self.num_workers = t.get('num_workers', 2)
        self.ragged = bool(t.get('ragged', False))
        self.bucket_batches = int(t.get('bucket_batches', 0))

    def setup(self, stage=None):
        self.train_dataset = TransformerTTSDataset(self.config, mode='train')
        self.valid_dataset = TransformerTTSDataset(self.config, mode='valid')

    def _loader(self, ds, train: bool):
        coll = collate_ragged if self.ragged else collate_fn
        kw = dict(num_workers=self.num_workers, collate_fn=coll, pin_memory=True)
        if train and self.bucket_batches > 0:
            import torch.distributed as dist
            r, w = (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)
            sampler = BucketBatchSampler(ds.mel_lengths(), self.batch_size, self.bucket_batches, rank=r, world_size=w)
            return torch.utils.data.DataLoader(ds, batch_sampler=sampler, **kw)
        return torch.utils.data.DataLoader(ds, batch_size=self.batch_size, shuffle=train, drop_last=train, **kw)

    def train_dataloader(self):
        return self._loader(self.train_dataset, True)

    def val_dataloader(self):
        return self._loader(self.valid_dataset, False)
