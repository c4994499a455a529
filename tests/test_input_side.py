"""Input side: Dataset / collate_fn contract, ragged staging + device-side padding, prefetcher,
length-bucketed sampler.  Integer / byte work: every comparison is bit-exact."""
import os

import numpy as np
import pytest
import torch

from oracle.collate import oracle_collate, oracle_getitem
from oracle.synth import synth_samples

GOLD = os.path.join(os.path.dirname(__file__), "golden", "collate.npz")
CASES = {"a": (7, 80), "b": (5, 16), "c": (1, 80)}       # as in make_golden.py::golden_collate
KEYS = ("phoneme", "melspec", "phoneme_lens", "melspec_lens")


def _write(tmp_path, samples):
    for i, s in enumerate(samples):
        np.savez(os.path.join(tmp_path, f"LJ0{10 + i}-0001.npz"), melspec=s["melspec"], transcript=s["transcript"],
                 phoneme=np.array(["x"]), sequence=s["sequence"])
    return {"path": {"preprocessed": str(tmp_path)}, "training": {"batch_size": 4, "num_workers": 0}}


def _same(got, gold, tag):
    for k in KEYS:
        g = gold[f"{tag}_{k}"]
        t = got[k].cpu().numpy()
        assert t.dtype == g.dtype and t.shape == g.shape, (k, t.dtype, g.dtype, t.shape, g.shape)
        assert np.array_equal(t, g), k
    assert list(got["transcript"]) == list(gold[f"{tag}_transcript"])


@pytest.mark.parametrize("tag", list(CASES))
def test_oracle_collate_matches_reference_fixture(tag):
    n, n_mels = CASES[tag]
    gold = np.load(GOLD)
    items = [oracle_getitem(s) for s in synth_samples(n, n_mels=n_mels, seed=77 + n)]
    _same(oracle_collate(items), gold, tag)


@pytest.mark.parametrize("tag", list(CASES))
def test_dataset_and_collate_fn_match_reference_fixture(tag, tmp_path):
    from transformertts_amd.dataset import TransformerTTSDataset, collate_fn
    n, n_mels = CASES[tag]
    cfg = _write(tmp_path, synth_samples(n, n_mels=n_mels, seed=77 + n))
    ds = TransformerTTSDataset(cfg, mode="train")
    assert len(ds) == n and len(TransformerTTSDataset(cfg, mode="valid")) == 0
    items = [ds[i] for i in range(n)]
    assert items[0]["melspec"].shape[1] == n_mels and items[0]["phoneme"].dtype == torch.int64
    _same(collate_fn(items), np.load(GOLD), tag)
    assert ds.mel_lengths() == [int(it["melspec"].shape[0]) for it in items]


def test_valid_split_prefixes(tmp_path):
    from transformertts_amd.dataset import TransformerTTSDataset
    s = synth_samples(4, n_mels=8)
    for name, smp in zip(["LJ001-0001", "LJ003-0002", "LJ004-0001", "LJ050-0001"], s):
        np.savez(os.path.join(tmp_path, name + ".npz"), melspec=smp["melspec"], transcript="t", sequence=smp["sequence"])
    cfg = {"path": {"preprocessed": str(tmp_path)}}
    assert TransformerTTSDataset(cfg, "valid").data_list == ["LJ001-0001.npz", "LJ003-0002.npz"]
    assert TransformerTTSDataset(cfg, "train").data_list == ["LJ004-0001.npz", "LJ050-0001.npz"]


@pytest.mark.parametrize("tag", list(CASES))
def test_collate_ragged_host_layout(tag):
    """The ragged batch holds exactly the padded batch's payload: stored (n_mels, T) blocks in sorted order."""
    from transformertts_amd.dataset import collate_ragged
    n, n_mels = CASES[tag]
    gold = np.load(GOLD)
    items = [oracle_getitem(s) for s in synth_samples(n, n_mels=n_mels, seed=77 + n)]
    r = collate_ragged(items)
    assert r["n_mels"] == n_mels and r["mel_ragged"].dtype == torch.float32 and r["phoneme_ragged"].dtype == torch.int64
    assert np.array_equal(r["phoneme_lens"].numpy(), gold[f"{tag}_phoneme_lens"])
    assert np.array_equal(r["melspec_lens"].numpy(), gold[f"{tag}_melspec_lens"])
    fo, po = r["frame_offsets"].numpy(), r["phoneme_offsets"].numpy()
    assert fo[0] == 0 and np.array_equal(np.diff(fo), gold[f"{tag}_melspec_lens"])
    assert po[0] == 0 and np.array_equal(np.diff(po), gold[f"{tag}_phoneme_lens"])
    mel, ids = r["mel_ragged"].numpy(), r["phoneme_ragged"].numpy()
    assert mel.size == fo[-1] * n_mels and ids.size == po[-1]
    for b in range(n):
        L = int(fo[b + 1] - fo[b])
        blk = mel[fo[b] * n_mels:fo[b + 1] * n_mels].reshape(n_mels, L)
        assert np.array_equal(blk.T, gold[f"{tag}_melspec"][b, :L])
        assert np.array_equal(ids[po[b]:po[b + 1]], gold[f"{tag}_phoneme"][b, :po[b + 1] - po[b]])
    assert list(r["transcript"]) == list(gold[f"{tag}_transcript"])


def test_bucket_sampler_partitions_and_cuts_padding():
    from transformertts_amd.dataset import BucketBatchSampler
    rng = np.random.default_rng(0)
    lens = np.clip(rng.normal(566, 170, size=1000), 95, 870).astype(np.int64)
    plain = BucketBatchSampler(lens, 16, bucket_batches=1, seed=3)
    buck = BucketBatchSampler(lens, 16, bucket_batches=16, seed=3)
    for s in (plain, buck):
        flat = [i for b in s for i in b]
        assert len(flat) == len(set(flat)) and all(len(b) == 16 for b in s) and len(s) == 1000 // 16
        assert [list(b) for b in s] == [list(b) for b in s]          # deterministic within an epoch
    assert buck.padding_fraction() < 0.5 * plain.padding_fraction()
    e0 = [list(b) for b in buck]
    buck.set_epoch(1)
    assert [list(b) for b in buck] != e0
    # ranks: disjoint, same count
    r0 = BucketBatchSampler(lens, 16, 16, seed=3, rank=0, world_size=2)
    r1 = BucketBatchSampler(lens, 16, 16, seed=3, rank=1, world_size=2)
    a, b = [i for x in r0 for i in x], [i for x in r1 for i in x]
    assert len(r0) == len(r1) == (1000 // 16) // 2 and not set(a) & set(b)
    # drop_last=False keeps every index
    keep = BucketBatchSampler(lens, 16, 4, drop_last=False, shuffle=False)
    assert sorted(i for x in keep for i in x) == list(range(1000))
    with pytest.raises(ValueError):
        BucketBatchSampler(lens, 0)
    with pytest.raises(ValueError):
        BucketBatchSampler(lens, 4, rank=2, world_size=2)


def test_data_module_loaders(tmp_path):
    from transformertts_amd.dataset import DataModule
    cfg = _write(tmp_path, synth_samples(9, n_mels=16, seed=5))
    dm = DataModule(cfg)
    dm.setup()
    batches = list(dm.train_dataloader())
    assert len(batches) == 2 and all(b["melspec"].shape[0] == 4 for b in batches)       # drop_last
    assert len(list(dm.val_dataloader())) == 0
    cfg["training"].update(ragged=True, bucket_batches=2)
    dm = DataModule(cfg)
    dm.setup()
    batches = list(dm.train_dataloader())
    assert len(batches) == 2 and all(b["ragged"] and b["melspec_lens"].numel() == 4 for b in batches)


def test_stager_requires_device():
    from transformertts_amd.dataset import DeviceStager
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            DeviceStager()


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("tag", list(CASES))
def test_device_padding_matches_reference_fixture(tag):
    from transformertts_amd.dataset import DeviceStager, collate_fn, collate_ragged
    n, n_mels = CASES[tag]
    gold = np.load(GOLD)
    items = [oracle_getitem(s) for s in synth_samples(n, n_mels=n_mels, seed=77 + n)]
    st = DeviceStager("cuda")
    dev = DeviceStager.wait(st.stage(collate_ragged(items)))
    torch.cuda.synchronize()
    assert all(dev[k].is_cuda for k in KEYS)
    _same(dev, gold, tag)
    dev = DeviceStager.wait(st.stage(collate_fn(items)))          # padded host batch: plain async copy
    torch.cuda.synchronize()
    _same(dev, gold, tag)


@pytest.mark.gpu
def test_device_padding_full_size_against_oracle():
    """B=64 LJSpeech-sized utterances (up to 870 frames, tile remainders, a 1-frame utterance)."""
    from transformertts_amd.dataset import DeviceStager, collate_ragged
    samples = synth_samples(64, n_mels=80, max_frames=867, seed=9)
    samples[5]["melspec"] = samples[5]["melspec"][:, :1].copy()
    samples[6]["melspec"] = samples[6]["melspec"][:, :64].copy()
    samples[7]["melspec"] = samples[7]["melspec"][:, :65].copy()
    items = [oracle_getitem(s) for s in samples]
    want = oracle_collate(items)
    dev = DeviceStager.wait(DeviceStager("cuda").stage(collate_ragged(items)))
    torch.cuda.synchronize()
    for k in KEYS:
        assert torch.equal(dev[k].cpu(), want[k]), k
    assert dev["melspec"].shape == (64, 870, 80)


@pytest.mark.gpu
def test_collate_kernels_argument_checks():
    from transformertts_amd import _lib
    lib = _lib.load()
    x = torch.zeros(16, device="cuda")
    assert lib.ttts_collate_melspec(x.data_ptr(), x.data_ptr(), x.data_ptr(), 1, 1, 129, None) != 0
    assert b"n_mels" in lib.ttts_last_error()
    assert lib.ttts_collate_melspec(None, None, None, 0, 0, 80, None) == 0            # empty batch is a no-op
    assert lib.ttts_collate_phoneme(None, None, None, 0, 5, None) == 0


@pytest.mark.gpu
def test_prefetcher_order_and_training_step_consumes_it():
    from transformertts_amd.dataset import DevicePrefetcher, collate_ragged
    batches, wants = [], []
    for s in range(5):
        items = [oracle_getitem(x) for x in synth_samples(4, n_mels=16, n_phon=30, max_frames=40, seed=100 + s)]
        batches.append(collate_ragged(items))
        wants.append(oracle_collate(items))
    got = list(DevicePrefetcher(batches, "cuda", depth=2))
    torch.cuda.synchronize()
    assert len(got) == 5
    for g, w in zip(got, wants):
        for k in KEYS:
            assert torch.equal(g[k].cpu(), w[k]), k
    # the module surface takes the staged batch as is (prepare_batch is a no-op for device tensors)
    from oracle.spec import model_config
    from transformertts_amd.lightning_module import LightningModule
    cfg = {"model": dict(model_config("tiny"), device="cuda"), "loss": {"stop_weight": 8.0},
           "training": {"num_epochs": 300, "teacher_forcing_mode": "linear", "warmup_steps": 4000}}
    lm = LightningModule(cfg).to("cuda")
    loss = lm.training_step(got[0], 0)
    assert torch.isfinite(loss).item()
