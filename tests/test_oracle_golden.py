"""oracle/ (CPU restatement) against the fixtures generated from the real reference.

Tolerance: 1e-5 rel-L2 on outputs (measured 4e-6 on base-config post_melspec: the restatement differs
from torch's own MHA/SDPA kernels only in fp32 rounding order), 2e-5 on gradient samples.
"""
import os

import numpy as np
import pytest
import torch

from conftest import rel_l2
from oracle import (model_config, fill_state, synth_batch, oracle_forward, oracle_loss, oracle_training_step,
                    oracle_inference)
from oracle.ref_model import teacher_forcing_ratio, noam_lambda, scheduled_sampling_mix

OUT_TOL = 1e-5
GRAD_TOL = 2e-5


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, f"{name}.npz"), allow_pickle=False)


def _setup(g):
    cfg = model_config(str(g["meta/cfg_name"]))
    sd = fill_state(cfg, int(g["meta/w_seed"]))
    batch = synth_batch(int(g["meta/B"]), int(g["meta/Tp"]), int(g["meta/Tm"]), cfg["n_mels"], cfg["n_phon"],
                        ragged=True, seed=int(g["meta/b_seed"]))
    assert np.array_equal(batch["melspec_lens"].numpy(), g["meta/melspec_lens"])
    return cfg, sd, batch


def _check_grads(g, sd, tol):
    worst = 0.0
    for k in [k for k in g.files if k.startswith("gradsample/")]:
        name = k.split("/", 1)[1]
        stride = int(g[f"gradstride/{name}"])
        grad = sd[name].grad
        got = grad.flatten()[::stride] if grad is not None else torch.zeros(g[k].shape)
        ref = torch.from_numpy(g[k])
        gn = float(g[f"gradnorm/{name}"])
        if gn < 1e-6:      # e.g. conv bias in front of train-mode BN: analytically zero
            assert got.abs().max().item() < 1e-4, name
            continue
        # error relative to the whole tensor's norm (samples are a strided subset)
        err = (got.double() - ref.double()).norm().item() / (ref.double().norm().item() + 1e-30)
        worst = max(worst, err)
        assert err < tol, (name, err)
        full = grad.double().norm().item()
        assert abs(full - gn) / gn < tol * 10, (name, full, gn)
    return worst


# The scaled configuration (12 layers of d_model 512, d_ffn 2048) is held against the reference with the oracle evaluated
# in fp64: two fp32 evaluations of that depth each gate a few of their 2.5e6 ReLU units differently from exact arithmetic
# (see tests/test_hip_model.py::test_gradient_gap_is_relu_gate_flips), so fp32-vs-fp32 gradients differ by up to 8e-4,
# while exact arithmetic of the restatement reproduces the reference's fp32 gradients to 5.5e-5 and its outputs to
# 2.5e-5 (the reference's own fp32 rounding through the 512-channel post-net): measured when the fixture was generated.
@pytest.mark.parametrize("name,dtype,out_tol,grad_tol", [("tiny_model", torch.float32, OUT_TOL, GRAD_TOL),
                                                         ("micro_model", torch.float32, OUT_TOL, GRAD_TOL),
                                                         ("base_model", torch.float32, OUT_TOL, GRAD_TOL),
                                                         ("scaled_model", torch.float64, 5e-5, 2e-4)])
def test_forward_eval_and_train(golden_dir, name, dtype, out_tol, grad_tol):
    g = _load(golden_dir, name)
    cfg, sd, batch = _setup(g)
    st = int(g["meta/align_stride"])

    def state():
        sd_ = fill_state(cfg, int(g["meta/w_seed"]))
        for k in list(sd_):
            if sd_[k].is_floating_point():
                sd_[k] = sd_[k].to(dtype)
        return sd_
    args = (batch["phoneme"], batch["melspec"].to(dtype), batch["phoneme_lens"], batch["melspec_lens"])
    sd = state()
    with torch.no_grad():
        out = oracle_forward(sd, cfg, *args, training=False)
    for k in ("pred_melspec", "post_melspec", "pred_stop"):
        assert rel_l2(out[k], torch.from_numpy(g[f"eval/{k}"])) < out_tol, k
    for i, a in enumerate(out["alignments"]):
        assert rel_l2(a[:, :, ::st], torch.from_numpy(g[f"eval/align{i}"])) < out_tol

    sd = state()
    for k, v in sd.items():
        if v.is_floating_point() and "running" not in k and k != "pe.pe":
            v.requires_grad_(True)
    out = oracle_forward(sd, cfg, *args, training=True, dropout=False)
    for k in ("pred_melspec", "post_melspec", "pred_stop"):
        assert rel_l2(out[k], torch.from_numpy(g[f"train/{k}"])) < out_tol, k
    for i, a in enumerate(out["alignments"]):
        assert rel_l2(a[:, :, ::st], torch.from_numpy(g[f"train/align{i}"])) < out_tol
    loss = oracle_loss(out, args[1], batch["melspec_lens"])
    for k in ("total", "pred_mel", "post_mel", "stop"):
        assert abs(loss[k].item() - float(g[f"train/loss_{k}"])) < 1e-5 * max(1.0, abs(float(g[f"train/loss_{k}"])))
    loss["total"].backward()
    _check_grads(g, sd, grad_tol)
    for k in [k for k in g.files if k.startswith("bn/")]:
        name_ = k.split("/", 1)[1]
        ref = torch.from_numpy(np.asarray(g[k]))
        if ref.dtype == torch.int64:
            assert int(sd[name_]) == int(ref)
        else:
            assert rel_l2(sd[name_], ref) < out_tol, name_


def test_training_step(golden_dir):
    g = _load(golden_dir, "tiny_step")
    cfg, sd, batch = _setup(g) if "meta/melspec_lens" in g.files else (None, None, None)
    if cfg is None:
        cfg = model_config(str(g["meta/cfg_name"]))
        sd = fill_state(cfg, int(g["meta/w_seed"]))
        batch = synth_batch(int(g["meta/B"]), int(g["meta/Tp"]), int(g["meta/Tm"]), cfg["n_mels"], cfg["n_phon"],
                            ragged=True, seed=int(g["meta/b_seed"]))
    for k, v in sd.items():
        if v.is_floating_point() and "running" not in k and k != "pe.pe":
            v.requires_grad_(True)
    loss, _, _ = oracle_training_step(sd, cfg, batch, epoch=int(g["meta/epoch"]), num_epochs=300, tf_mode="linear",
                                      dropout=False, seed_u=torch.from_numpy(g["seed_u"]))
    assert abs(loss["total"].item() - float(g["loss_total"])) < 1e-5 * abs(float(g["loss_total"]))
    loss["total"].backward()
    _check_grads(g, sd, GRAD_TOL)
    for k in [k for k in g.files if k.startswith("bn/")]:
        name_ = k.split("/", 1)[1]
        ref = torch.from_numpy(np.asarray(g[k]))
        if ref.dtype == torch.int64:
            assert int(sd[name_]) == int(ref) == 2      # BN stats advance on BOTH forwards of a step
        else:
            assert rel_l2(sd[name_], ref) < OUT_TOL, name_
    lam = noam_lambda(cfg["d_model"], 4000)
    assert np.allclose([lam(s) for s in (0, 1, 100, 4000, 16000)], g["noam_lambda"], rtol=1e-12)


def test_helpers(golden_dir):
    g = _load(golden_dir, "helpers")
    for mode in ("linear", "cosine", "constant"):
        got = [teacher_forcing_ratio(int(e), 300, mode, cycles=1) for e in g["tf/epochs"]]
        assert np.allclose(got, g[f"tf/{mode}"], rtol=0, atol=1e-15)
    lam = noam_lambda(256, 4000)
    assert np.allclose([lam(int(s)) for s in g["noam/steps"]], g["noam/256_4000"], rtol=1e-14)
    pred, mel, lens = (torch.from_numpy(g["ss/pred"]), torch.from_numpy(g["ss/mel"]), torch.from_numpy(g["ss/lens"]))
    for p_tf in (1.0, 0.7, 0.05):
        mixed = scheduled_sampling_mix(pred, mel, lens, p_tf, torch.from_numpy(g[f"ss/u_{p_tf}"]))
        assert torch.equal(mixed, torch.from_numpy(g[f"ss/mixed_{p_tf}"]))
    outs = {"pred_melspec": pred, "post_melspec": torch.from_numpy(g["loss/post"]),
            "pred_stop": torch.from_numpy(g["loss/stop_logits"])}
    ls = oracle_loss(outs, mel, lens)
    for k in ("total", "pred_mel", "post_mel", "stop"):
        assert abs(ls[k].item() - float(g[f"loss/{k}"])) < 2e-6 * max(1.0, abs(float(g[f"loss/{k}"])))


def test_inference(golden_dir):
    """The restated autoregressive loop against the reference's own `inference()` (13 forced steps, tiny config)."""
    g = _load(golden_dir, "tiny_inference")
    cfg = model_config(str(g["meta/cfg_name"]))
    sd = fill_state(cfg, int(g["meta/w_seed"]))
    batch = synth_batch(int(g["meta/B"]), int(g["meta/Tp"]), 40, cfg["n_mels"], cfg["n_phon"], ragged=True,
                        seed=int(g["meta/b_seed"]))
    out = oracle_inference(sd, cfg, batch["phoneme"], batch["phoneme_lens"], max_len=int(g["meta/max_len"]),
                           stop_threshold=2.0)
    for k in ("pred_melspec", "post_melspec", "pred_stop"):
        assert out[k].shape == tuple(g[k].shape), k
        assert rel_l2(out[k], torch.from_numpy(g[k])) < 2e-5, (k, rel_l2(out[k], torch.from_numpy(g[k])))
