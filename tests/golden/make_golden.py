#!/usr/bin/env python3
"""Generate the golden fixtures that pin `oracle/` to the real reference.

Runs ONLY in the build container (needs /root/reference, which never travels to the GPU
box): imports the unmodified reference model, loads the seed-defined weights of
`oracle.spec.fill_state`, feeds the seed-defined batch of `oracle.synth.synth_batch`, and
stores inputs-by-seed + expected outputs as small .npz files next to this script.

    python tests/golden/make_golden.py

Only data is written (expected outputs, gradient samples/norms, scalar tables); no
reference source is copied.  `pytorch_lightning` and `loguru` are absent from this image, so
the step-surface golden uses two non-arithmetic stand-ins injected into `sys.modules`
(a no-op logger; `LightningModule` = `nn.Module` with `device`/`current_epoch`/`log`).
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)

from oracle.spec import model_config, state_spec, fill_state  # noqa: E402
from oracle.synth import synth_batch  # noqa: E402

MAX_GRAD_SAMPLES = 4096


def _install_stubs():
    lg = types.ModuleType("loguru")

    class _L:
        def __getattr__(self, name):
            return lambda *a, **k: self
    lg.logger = _L()
    sys.modules.setdefault("loguru", lg)
    pl = types.ModuleType("pytorch_lightning")

    class LightningModule(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.current_epoch = 0

        @property
        def device(self):
            return torch.device("cpu")

        def log(self, *a, **k):
            pass
    pl.LightningModule = LightningModule
    pl.LightningDataModule = object
    sys.modules.setdefault("pytorch_lightning", pl)
    mp = types.ModuleType("matplotlib")
    mpp = types.ModuleType("matplotlib.pyplot")
    try:
        import matplotlib  # noqa: F401
    except Exception:
        sys.modules.setdefault("matplotlib", mp)
        sys.modules.setdefault("matplotlib.pyplot", mpp)


def _ref_model(cfg, seed):
    sys.path.insert(0, REF)
    from model import TransformerTTS  # the real reference
    m = TransformerTTS(**cfg, device="cpu")
    sd = fill_state(cfg, seed)
    ref_sd = m.state_dict()
    spec = state_spec(cfg)
    assert list(ref_sd.keys()) == list(spec.keys()), "state-dict key order/contents differ from oracle.spec"
    for k, v in ref_sd.items():
        assert tuple(v.shape) == tuple(spec[k]), (k, v.shape, spec[k])
    m.load_state_dict(sd, strict=True)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
        if isinstance(mod, torch.nn.MultiheadAttention):
            mod.dropout = 0.0
    return m


def _np(t):
    return t.detach().cpu().numpy()


def _grad_record(named_params, max_samples=MAX_GRAD_SAMPLES):
    rec = {}
    for k, p in named_params:
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        flat = g.flatten()
        stride = max(1, (flat.numel() + max_samples - 1) // max_samples)
        rec[f"gradnorm/{k}"] = np.float64(g.double().norm().item())
        rec[f"gradsample/{k}"] = _np(flat[::stride])
        rec[f"gradstride/{k}"] = np.int64(stride)
    return rec


def golden_model(name, cfg_name, B, Tp, Tm, w_seed, b_seed, align_stride, max_grad_samples=MAX_GRAD_SAMPLES):
    cfg = model_config(cfg_name)
    batch = synth_batch(B, Tp, Tm, cfg["n_mels"], cfg["n_phon"], ragged=True, seed=b_seed)
    args = (batch["phoneme"], batch["melspec"], batch["phoneme_lens"], batch["melspec_lens"])
    rec = {"meta/cfg_name": cfg_name, "meta/B": B, "meta/Tp": Tp, "meta/Tm": Tm,
           "meta/w_seed": w_seed, "meta/b_seed": b_seed, "meta/align_stride": align_stride,
           "meta/phoneme_lens": _np(batch["phoneme_lens"]), "meta/melspec_lens": _np(batch["melspec_lens"])}

    # (i) eval-mode forward
    m = _ref_model(cfg, w_seed)
    m.eval()
    with torch.no_grad():
        out = m(*args)
    rec["eval/pred_melspec"] = _np(out["pred_melspec"])
    rec["eval/post_melspec"] = _np(out["post_melspec"])
    rec["eval/pred_stop"] = _np(out["pred_stop"])
    for i, a in enumerate(out["alignments"]):
        rec[f"eval/align{i}"] = _np(a[:, :, ::align_stride])

    # (ii)+(iii) train-mode forward (BN batch statistics) + loss + backward
    from loss import TransformerTTSLoss
    m = _ref_model(cfg, w_seed)
    m.train()
    out = m(*args)
    crit = TransformerTTSLoss(stop_weight=8.0)
    loss = crit(out, batch["melspec"], batch["melspec_lens"])
    loss["total"].backward()
    rec["train/pred_melspec"] = _np(out["pred_melspec"])
    rec["train/post_melspec"] = _np(out["post_melspec"])
    rec["train/pred_stop"] = _np(out["pred_stop"])
    for i, a in enumerate(out["alignments"]):
        rec[f"train/align{i}"] = _np(a[:, :, ::align_stride])
    for k in ("total", "pred_mel", "post_mel", "stop"):
        rec[f"train/loss_{k}"] = np.float64(loss[k].item())
    rec.update(_grad_record(m.named_parameters(), max_grad_samples))
    for k, v in m.state_dict().items():
        if "running_" in k or "num_batches" in k:
            rec[f"bn/{k}"] = _np(v)
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **rec)
    print(name, "written;", "loss", float(loss["total"]))


def golden_step(name, cfg_name, B, Tp, Tm, w_seed, b_seed, epoch):
    """The real LightningModule.training_step (lightning_module.py:45-86) with dropout off."""
    _install_stubs()
    sys.path.insert(0, REF)
    import lightning_module as lm
    import utils.plot as uplot  # noqa: F401
    lm.plot_mels_scheduled = lambda *a, **k: None
    cfg = model_config(cfg_name)
    config = {"model": dict(cfg, device="cpu"), "loss": {"stop_weight": 8.0},
              "training": {"num_epochs": 300, "teacher_forcing_mode": "linear", "log_interval": 10 ** 9,
                           "warmup_steps": 4000}}
    mod = lm.LightningModule(config, exp_dir=None)
    mod.model.load_state_dict(fill_state(cfg, w_seed), strict=True)
    for sub in mod.modules():
        if isinstance(sub, torch.nn.Dropout):
            sub.p = 0.0
        if isinstance(sub, torch.nn.MultiheadAttention):
            sub.dropout = 0.0
    mod.train()
    mod.current_epoch = epoch
    batch = synth_batch(B, Tp, Tm, cfg["n_mels"], cfg["n_phon"], ragged=True, seed=b_seed)
    torch.manual_seed(777)
    u = torch.rand(B, 1, batch["melspec"].size(1))     # the draw block_mask will make (util.py:108)
    torch.manual_seed(777)
    loss = mod.training_step(dict(batch), batch_idx=1)
    loss.backward()
    rec = {"meta/cfg_name": cfg_name, "meta/B": B, "meta/Tp": Tp, "meta/Tm": Tm, "meta/w_seed": w_seed,
           "meta/b_seed": b_seed, "meta/epoch": epoch, "seed_u": _np(u), "loss_total": np.float64(loss.item())}
    rec.update(_grad_record(mod.model.named_parameters()))
    for k, v in mod.model.state_dict().items():
        if "running_" in k or "num_batches" in k:
            rec[f"bn/{k}"] = _np(v)
    opt = mod.configure_optimizers()
    sched = opt["lr_scheduler"]["scheduler"]
    rec["noam_lambda"] = np.array([sched.lr_lambdas[0](s) for s in (0, 1, 100, 4000, 16000)], dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **rec)
    print(name, "written; loss", float(loss))


def golden_inference(name, cfg_name, B, Tp, w_seed, b_seed, max_len):
    """The real TransformerTTS.inference (model/model.py:323-394), stop threshold out of reach -> max_len-1 steps."""
    cfg = model_config(cfg_name)
    m = _ref_model(cfg, w_seed)
    batch = synth_batch(B, Tp, 40, cfg["n_mels"], cfg["n_phon"], ragged=True, seed=b_seed)
    out = m.inference(batch["phoneme"], batch["phoneme_lens"], max_len=max_len, stop_threshold=2.0)
    rec = {"meta/cfg_name": cfg_name, "meta/B": B, "meta/Tp": Tp, "meta/w_seed": w_seed, "meta/b_seed": b_seed,
           "meta/max_len": max_len, "pred_melspec": _np(out["pred_melspec"]), "post_melspec": _np(out["post_melspec"]),
           "pred_stop": _np(out["pred_stop"])}
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **rec)
    print(name, "written;", tuple(out["pred_melspec"].shape))


def golden_helpers(name):
    _install_stubs()
    sys.path.insert(0, REF)
    from utils.util import get_teacher_forcing_ratio, get_noam_scheduler, block_mask, apply_teacher_forcing
    from loss import TransformerTTSLoss
    rec = {}
    epochs = [1, 5, 9, 10, 11, 50, 155, 299, 300, 400]
    for mode in ("linear", "cosine", "constant"):
        rec[f"tf/{mode}"] = np.array([get_teacher_forcing_ratio(e, 300, mode, cycles=1) for e in epochs])
    rec["tf/epochs"] = np.array(epochs)
    steps = [0, 1, 2, 100, 3999, 4000, 4001, 16000, 100000]
    lam = get_noam_scheduler(256, 4000)
    rec["noam/steps"] = np.array(steps)
    rec["noam/256_4000"] = np.array([lam(s) for s in steps])
    g = torch.Generator().manual_seed(5)
    B, T, C = 3, 37, 16
    pred = torch.randn(B, T, C, generator=g)
    mel = torch.randn(B, T, C, generator=g)
    lens = torch.tensor([37, 20, 9])
    for p_tf in (1.0, 0.7, 0.05):
        torch.manual_seed(99)
        u = torch.rand(B, 1, T)
        torch.manual_seed(99)
        bm = block_mask(pred, p_tf, 8)
        torch.manual_seed(99)
        mixed = apply_teacher_forcing(pred, mel, lens, p_tf, "cpu")
        rec[f"ss/u_{p_tf}"] = _np(u)
        rec[f"ss/mask_{p_tf}"] = _np(bm)
        rec[f"ss/mixed_{p_tf}"] = _np(mixed)
    rec["ss/pred"], rec["ss/mel"], rec["ss/lens"] = _np(pred), _np(mel), _np(lens)
    crit = TransformerTTSLoss(stop_weight=8.0)
    outs = {"pred_melspec": pred, "post_melspec": torch.randn(B, T, C, generator=g),
            "pred_stop": torch.randn(B, T, generator=g)}
    ls = crit(outs, mel, lens)
    rec["loss/post"], rec["loss/stop_logits"] = _np(outs["post_melspec"]), _np(outs["pred_stop"])
    for k, v in ls.items():
        rec[f"loss/{k}"] = np.float64(v.item())
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **rec)
    print(name, "written")


def golden_collate(name):
    """The reference's own Dataset.__getitem__ + collate_fn on files written in preprocess.py's format."""
    import tempfile
    _install_stubs()
    from oracle.synth import synth_samples
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import dataset as ref_dataset                      # the real reference
    out = {}
    for tag, (n, n_mels) in {"a": (7, 80), "b": (5, 16), "c": (1, 80)}.items():
        samples = synth_samples(n, n_mels=n_mels, seed=77 + n)
        with tempfile.TemporaryDirectory() as d:
            for i, s in enumerate(samples):
                np.savez(os.path.join(d, f"LJ0{10 + i}-0001.npz"), melspec=s["melspec"], transcript=s["transcript"],
                         phoneme=np.array(["x"]), sequence=s["sequence"])
            ds = ref_dataset.TransformerTTSDataset({"path": {"preprocessed": d}}, mode="train")
            items = [ds[i] for i in range(len(ds))]
        b = ref_dataset.collate_fn(items)
        for k in ("phoneme", "melspec", "phoneme_lens", "melspec_lens"):
            out[f"{tag}_{k}"] = _np(b[k])
        out[f"{tag}_transcript"] = np.array(b["transcript"])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "written")


if __name__ == "__main__":
    torch.set_num_threads(8)
    only = set(sys.argv[1:])            # e.g. `make_golden.py scaled_model`: regenerate the named fixtures only
    jobs = [
        ("helpers", lambda: golden_helpers("helpers")),
        ("tiny_model", lambda: golden_model("tiny_model", "tiny", B=3, Tp=12, Tm=40, w_seed=11, b_seed=21, align_stride=1)),
        ("base_model", lambda: golden_model("base_model", "base", B=2, Tp=60, Tm=300, w_seed=12, b_seed=22, align_stride=8)),
        # SURVEY 8c's tiny-golden plan: d_model 32, 2 heads (head_dim 16), 1+1 layers, B = 3 ragged, Tp <= 12, Tm <= 40
        ("micro_model", lambda: golden_model("micro_model", "micro", B=3, Tp=12, Tm=40, w_seed=16, b_seed=26, align_stride=1)),
        # BASELINE configs[4]: d_model 512, 6+6 layers, 8 heads, d_ffn 2048 (fewer gradient samples per parameter: 282 tensors)
        ("scaled_model", lambda: golden_model("scaled_model", "scaled", B=2, Tp=60, Tm=300, w_seed=14, b_seed=24, align_stride=8,
                                              max_grad_samples=1024)),
        ("tiny_step", lambda: golden_step("tiny_step", "tiny", B=3, Tp=12, Tm=40, w_seed=11, b_seed=21, epoch=120)),
        ("collate", lambda: golden_collate("collate")),
        ("tiny_inference", lambda: golden_inference("tiny_inference", "tiny", B=3, Tp=12, w_seed=11, b_seed=21, max_len=14)),
    ]
    for name, job in jobs:
        if not only or name in only:
            job()
