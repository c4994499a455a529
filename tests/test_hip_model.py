"""End-to-end parity of the HIP TransformerTTS against the CPU oracle (fp64 evaluation of the restated
reference) on identical seed-defined weights and batches, dropout disabled exactly as for the golden fixtures
(all nn.Dropout.p = 0, MultiheadAttention.dropout = 0).  Gate: 1e-4 rel-L2 (north_star) on pred_melspec,
post_melspec, pred_stop, every alignment map, every parameter gradient and the updated BN buffers."""
import os

import numpy as np
import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu
GATE = 1e-4        # outputs: north_star's stated fp32 tolerance (measured 1e-6 .. 5e-6)
# Gradients: every backward kernel is individually within 1e-6 of fp64 (tests/test_hip_ops.py), but a full backward vs the
# fp64 oracle contains DISCRETE events: a ReLU pre-activation within rounding distance of zero gates differently in two correct
# evaluations ("gate flip"), which perturbs that unit's gradient and everything below it (one flipped unit of 9.5e6 has moved
# pe.alpha -- one scalar, summed over every position with heavy cancellation -- by 5e-3).  The end-to-end gradient gates are
# therefore CONTINUOUS comparisons: the fp64 oracle is evaluated under the HIP path's own ReLU gates (oracle.relu_gates), and
# every parameter of every case is held to the flip-free gate.  Raw comparisons are reports (gpurun_out/parity_*.txt).
FLIP_FREE_GATE = 2e-5          # gradients vs the fp64 oracle under the HIP path's own ReLU gates (measured <= 1.4e-5)
FLIP_FREE_GATE_SCALED = 5e-5   # ... through the 12 layers of the scaled model (stock fp32 torch under the same gates: 3e-5)
GRAD_GATE = 2e-3               # the tiny training_step surface test only (flips included; no flip observed there)


def _oracle_gated_grads(cfg, w_seed, batch, gates, dtype=torch.float64):
    """parameter gradients (and the loss) of the oracle evaluated in `dtype` with its ReLUs replaced by the given 0/1 gates"""
    from oracle import fill_state, oracle_forward, oracle_loss, relu_gates
    sd = fill_state(cfg, w_seed)
    for k in list(sd):
        if sd[k].is_floating_point():
            sd[k] = sd[k].to(dtype)
            if "running" not in k and k != "pe.pe":
                sd[k].requires_grad_(True)
    with relu_gates(gates=[g.to(dtype) for g in gates]):
        ref = oracle_forward(sd, cfg, batch["phoneme"], batch["melspec"].to(dtype), batch["phoneme_lens"], batch["melspec_lens"],
                             training=True, dropout=False)
    loss = oracle_loss(ref, batch["melspec"].to(dtype), batch["melspec_lens"])
    loss["total"].backward()
    return {k: v.grad for k, v in sd.items() if v.requires_grad}, loss


def _no_dropout(m):
    from transformertts_amd.model.layers import MultiheadAttention
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
        if isinstance(mod, MultiheadAttention):
            mod.dropout = 0.0


def _build(cfg_name, w_seed):
    from oracle import model_config, fill_state
    from transformertts_amd.model import TransformerTTS
    cfg = model_config(cfg_name)
    m = TransformerTTS(**cfg, device="cuda")
    m.load_state_dict(fill_state(cfg, w_seed), strict=True)
    m = m.to("cuda")
    _no_dropout(m)
    return cfg, m


def _oracle64(cfg, w_seed):
    from oracle import fill_state
    sd = fill_state(cfg, w_seed)
    for k in list(sd):
        if sd[k].is_floating_point():
            sd[k] = sd[k].double()
            if "running" not in k and k != "pe.pe":
                sd[k].requires_grad_(True)
    return sd


@pytest.mark.parametrize("cfg_name,B,Tp,Tm,w_seed,b_seed", [("tiny", 3, 12, 40, 11, 21), ("base", 2, 60, 300, 12, 22),
                                                            ("base", 4, 100, 870, 13, 23), ("scaled", 2, 60, 300, 14, 24),
                                                            ("base", 16, 100, 870, 15, 25),
                                                            ("micro", 3, 12, 40, 16, 26),       # d_model 32, head_dim 16
                                                            ("tiny1h", 3, 12, 40, 18, 28),      # d_model 128, ONE head of 128 columns
                                                            # utterances far beyond LJSpeech's longest (870 frames): 24
                                                            # query blocks, 47 key tiles, 3000 rows of the 5000-row pe table
                                                            ("base", 2, 200, 3000, 17, 27)])
def test_forward_backward_vs_oracle(cfg_name, B, Tp, Tm, w_seed, b_seed):
    from oracle import synth_batch, oracle_forward, oracle_loss
    from transformertts_amd.loss import TransformerTTSLoss
    cfg, m = _build(cfg_name, w_seed)
    batch = synth_batch(B, Tp, Tm, cfg["n_mels"], cfg["n_phon"], ragged=True, seed=b_seed)
    dev = torch.device("cuda")
    args = [batch[k].to(dev) for k in ("phoneme", "melspec", "phoneme_lens", "melspec_lens")]

    # eval forward
    m.eval()
    with torch.no_grad():
        out = m(*args)
    sd = _oracle64(cfg, w_seed)
    with torch.no_grad():
        ref = oracle_forward(sd, cfg, batch["phoneme"], batch["melspec"].double(), batch["phoneme_lens"],
                             batch["melspec_lens"], training=False)
    for k in ("pred_melspec", "post_melspec", "pred_stop"):
        assert rel_l2(out[k], ref[k]) < GATE, ("eval", k, rel_l2(out[k], ref[k]))
    for a, r in zip(out["alignments"], ref["alignments"]):
        assert rel_l2(a, r) < GATE

    # train forward + loss + backward; the HIP path's ReLU decisions are recorded (ops._relu_observer)
    from transformertts_amd import ops
    m.train()
    hip_gates = []
    ops._relu_observer = lambda y: hip_gates.append((y.detach() > 0).cpu())
    try:
        out = m(*args)
    finally:
        ops._relu_observer = None
    crit = TransformerTTSLoss(8.0).to(dev)
    loss = crit(out, args[1], args[3])
    loss["total"].backward()
    # outputs: against the fp64 oracle as it stands (its own gates)
    big = B * Tm > 6000                      # the raw-gradient report costs a second fp64 backward: small cases only
    from oracle import relu_gates
    with relu_gates() as rec:                # the oracle's own pre-activations at every ReLU site, in call order
        ref = oracle_forward(sd, cfg, batch["phoneme"], batch["melspec"].double(), batch["phoneme_lens"],
                             batch["melspec_lens"], training=True, dropout=False)
    rloss = oracle_loss(ref, batch["melspec"].double(), batch["melspec_lens"])
    # The gate decisions themselves, ABSOLUTELY (the gradient comparison below adopts the HIP path's gates, so a wrongly applied
    # gate or mask would be adopted with them): the HIP path may gate a unit differently from exact arithmetic only where the
    # pre-activation is within rounding distance of zero, and only a handful of units are.
    assert len(rec.pre) == len(hip_gates)
    flips, units = 0, 0
    for pre, gate in zip(rec.pre, hip_gates):
        diff = (pre > 0) != gate.reshape(pre.shape)
        flips += int(diff.sum())
        units += pre.numel()
        if diff.any():
            lim = 1e-5 * max(1.0, float(pre.abs().max()) / 30.0)
            assert float(pre[diff].abs().max()) < lim, ("a ReLU unit far from zero was gated differently", float(pre[diff].abs().max()))
    assert flips <= max(20, units // 100000), (flips, units)
    if not big:
        rloss["total"].backward()
    errs = {}
    for k in ("pred_melspec", "post_melspec", "pred_stop"):
        errs[k] = rel_l2(out[k], ref[k])
    for i, (a, r) in enumerate(zip(out["alignments"], ref["alignments"])):
        errs[f"align{i}"] = rel_l2(a, r)
    errs["loss"] = abs(loss["total"].item() - rloss["total"].item()) / abs(rloss["total"].item())
    for name, buf in m.named_buffers():
        if "running_" in name:
            errs[name] = rel_l2(buf, sd[name])
        if "num_batches" in name:
            assert int(buf.item()) == int(sd[name])
    # gradients: against the fp64 oracle evaluated UNDER THE HIP PATH'S GATES.  A ReLU is the one discontinuity of the path: a
    # pre-activation within rounding distance of zero is gated differently by two correct evaluations, and a flipped unit moves
    # every gradient below it by far more than any rounding does (test_gradient_gap_is_relu_gate_flips shows that the units that
    # differ are a handful, all with |pre-activation| < 1e-5).  With the gates fixed the comparison is continuous, and it is held
    # to the flip-free gate for EVERY parameter of EVERY case -- no per-parameter allowance.
    gated, gloss = _oracle_gated_grads(cfg, w_seed, batch, hip_gates)
    gerrs, raw = {}, {}
    for name, p in m.named_parameters():
        rg = gated[name]
        if rg.norm().item() < 1e-7 * max(1.0, sd[name].detach().norm().item()):   # analytically-zero grads (conv bias before BN)
            assert p.grad.abs().max().item() < 1e-4, name
            continue
        gerrs[name] = rel_l2(p.grad, rg)
        if not big:
            raw[name] = rel_l2(p.grad, sd[name].grad)
    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/parity_{cfg_name}_B{B}_Tm{Tm}.txt", "w") as f:
        f.write("# outputs vs the fp64 oracle; grad/: vs the fp64 oracle under the HIP path's ReLU gates [raw: vs its own gates]\n")
        for k, v in sorted(errs.items(), key=lambda kv: -kv[1]):
            f.write(f"{v:.3e} {k}\n")
        for k, v in sorted(gerrs.items(), key=lambda kv: -kv[1]):
            f.write(f"{v:.3e} grad/{k}" + (f"  [raw {raw[k]:.3e}]" if k in raw else "") + "\n")
    bad = {k: v for k, v in errs.items() if not v < GATE}
    # raw gradients (the oracle under its OWN gates) keep a loose bound of their own: a flipped unit moves single tensors by up
    # to ~1e-2 (pe.alpha: one scalar summed over every position), never more; with no flip at all they meet the flip-free gate
    RAW_GATE = 5e-2
    for k, v in raw.items():
        if not v < (RAW_GATE if flips else 10 * FLIP_FREE_GATE_SCALED):
            bad["raw/" + k] = v
    gate = FLIP_FREE_GATE_SCALED if cfg_name == "scaled" else FLIP_FREE_GATE
    over = {k: v for k, v in gerrs.items() if not v < gate}
    if over:
        # a parameter above the flat gate (seen: 2.6e-5 on a BatchNorm bias at 3000 frames, 7e-5 on pe.alpha of the scaled model --
        # both sums with heavy cancellation) is still held to what stock fp32 torch achieves UNDER THE SAME GATES: twice its error,
        # and never more than the contract's 1e-4 (north_star).  (Round 5 allowed a scalar parameter three times stock fp32's
        # error while pe.alpha of the scaled model read 1.1e-4 mid-round; the final build reads 4.6e-5 and no case needs it.)
        stock, _ = _oracle_gated_grads(cfg, w_seed, batch, hip_gates, dtype=torch.float32)
        for k, v in over.items():
            e32 = rel_l2(stock[k], gated[k])
            if not (v < 2.0 * e32 and v < GATE):
                bad[k] = (v, e32)
    assert not bad, bad


def test_train_forward_across_the_routing_thresholds():
    """Train-mode forward at B = 48 x 870 dense frames (M = 41 760 rows: 327 tiles of 128 x 256 >= ops.DMA_MIN_TILES, so the square
    256 -> 256 projections AND the gated data gradients take the LDS-DMA kernel, gemm_h3i -- routes no smaller end-to-end case
    reaches; the B = 64-only routes therefore meet the reference's arithmetic once) against the oracle in fp32 on the four
    outputs (batch statistics of BatchNorm over all 41 760 rows included).  Outputs only: the fp32 oracle's backward at this
    size is minutes of CPU; gradients at these routes are held per kernel by test_dma_gemm_streams_across_tiles."""
    from oracle import synth_batch, oracle_forward, fill_state
    from transformertts_amd import ops
    B, Tp, Tm = 48, 100, 870
    assert ops._dma_shape_ok(B * Tm, 256, 256, False) and ops._dma_shape_ok(B * Tm, 256, 1024, True)
    cfg, m = _build("base", 18)
    batch = synth_batch(B, Tp, Tm, cfg["n_mels"], cfg["n_phon"], ragged=False, seed=28)
    args = [batch[k].to("cuda") for k in ("phoneme", "melspec", "phoneme_lens", "melspec_lens")]
    m.train()
    with torch.no_grad():
        out = m(*args)
    sd = fill_state(cfg, 18)
    with torch.no_grad():
        ref = oracle_forward(sd, cfg, batch["phoneme"], batch["melspec"], batch["phoneme_lens"], batch["melspec_lens"],
                             training=True, dropout=False)
    errs = {k: rel_l2(out[k], ref[k]) for k in ("pred_melspec", "post_melspec", "pred_stop")}
    for i, (a, r) in enumerate(zip(out["alignments"], ref["alignments"])):
        errs[f"align{i}"] = rel_l2(a, r)
    for name, buf in m.named_buffers():
        if "running_" in name:
            errs[name] = rel_l2(buf, sd[name])
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/parity_base_B48_train_forward_fp32_oracle.txt", "w") as f:
        for k, v in sorted(errs.items(), key=lambda kv: -kv[1]):
            f.write(f"{v:.3e} {k}\n")
    assert all(v < GATE for v in errs.values()), {k: v for k, v in errs.items() if not v < GATE}


@pytest.mark.parametrize("fixture", ["base_model", "scaled_model", "micro_model"])
def test_golden_outputs_direct(golden_dir, fixture):
    """HIP path against the committed reference outputs themselves (fp32 reference run): base config and the scaled
    configuration of BASELINE configs[4] (d_model 512, 6+6 layers, 8 heads, d_ffn 2048)."""
    from oracle import synth_batch
    g = np.load(os.path.join(golden_dir, f"{fixture}.npz"))
    cfg, m = _build(str(g["meta/cfg_name"]), int(g["meta/w_seed"]))
    batch = synth_batch(int(g["meta/B"]), int(g["meta/Tp"]), int(g["meta/Tm"]), cfg["n_mels"], cfg["n_phon"], ragged=True,
                        seed=int(g["meta/b_seed"]))
    args = [batch[k].to("cuda") for k in ("phoneme", "melspec", "phoneme_lens", "melspec_lens")]
    st = int(g["meta/align_stride"])
    m.train()
    out = m(*args)
    for k in ("pred_melspec", "post_melspec", "pred_stop"):
        assert rel_l2(out[k], torch.from_numpy(g[f"train/{k}"])) < GATE, k
    for i, a in enumerate(out["alignments"]):
        assert rel_l2(a[:, :, ::st], torch.from_numpy(g[f"train/align{i}"])) < GATE


@pytest.mark.parametrize("fixture", ["base_model", "scaled_model"])
def test_golden_gradients_direct(golden_dir, fixture):
    """HIP gradients against the reference's OWN fp32 backward (tests/golden/{base,scaled}_model.npz: per-parameter norms and
    strided samples written by make_golden.py from the imported reference).  The reference's gradients are an fp32 evaluation
    with its own ReLU gate decisions, so a raw comparison measures which units two evaluations happened to gate differently
    (it is written to the report).  What is ASSERTED is continuous: (1) the HIP gradients equal exact arithmetic under the HIP
    path's gates to the flip-free gate, and (2) they are as close to the reference's numbers as that exact evaluation is --
    whatever separates them from the fixture is the fixture's own distance from exact arithmetic, not this path's."""
    from oracle import synth_batch
    from transformertts_amd import ops
    from transformertts_amd.loss import TransformerTTSLoss
    g = np.load(os.path.join(golden_dir, f"{fixture}.npz"))
    cfg_name, w_seed = str(g["meta/cfg_name"]), int(g["meta/w_seed"])
    cfg, m = _build(cfg_name, w_seed)
    batch = synth_batch(int(g["meta/B"]), int(g["meta/Tp"]), int(g["meta/Tm"]), cfg["n_mels"], cfg["n_phon"], ragged=True,
                        seed=int(g["meta/b_seed"]))
    args = [batch[k].to("cuda") for k in ("phoneme", "melspec", "phoneme_lens", "melspec_lens")]
    m.train()
    hip_gates = []
    ops._relu_observer = lambda y: hip_gates.append((y.detach() > 0).cpu())
    try:
        out = m(*args)
    finally:
        ops._relu_observer = None
    loss = TransformerTTSLoss(8.0).to("cuda")(out, args[1], args[3])
    assert abs(loss["total"].item() - float(g["train/loss_total"])) < 1e-5 * abs(float(g["train/loss_total"]))
    loss["total"].backward()
    exact, _ = _oracle_gated_grads(cfg, w_seed, batch, hip_gates)
    gate = FLIP_FREE_GATE_SCALED if cfg_name == "scaled" else FLIP_FREE_GATE
    rows, bad, over = [], {}, {}
    for name, p in m.named_parameters():
        ref_norm = float(g[f"gradnorm/{name}"])
        stride = int(g[f"gradstride/{name}"])
        ref = torch.from_numpy(g[f"gradsample/{name}"])
        if ref_norm < 1e-6:                       # analytically zero (conv bias in front of BN, key bias of a softmax):
            assert float(p.grad.norm()) < 1e-4, name      # rounding noise on both sides (2e-7 in the 512-channel post-net)
            continue
        e_exact_hip = rel_l2(p.grad, exact[name])
        e_hip = rel_l2(p.grad.flatten()[::stride].cpu(), ref)
        e_exact = rel_l2(exact[name].flatten()[::stride], ref)
        rows.append((e_hip, e_exact, e_exact_hip, name))
        if not e_exact_hip < gate:
            over[name] = e_exact_hip
        elif not e_hip <= e_exact + 2.0 * gate:
            bad[name] = ("farther from the reference's fp32 gradients than exact arithmetic is", e_hip, e_exact)
    if over:
        # above the flat gate: held to what stock fp32 torch achieves under the same gates -- twice its error, and never above the
        # contract's 1e-4 (see test_forward_backward_vs_oracle)
        stock, _ = _oracle_gated_grads(cfg, w_seed, batch, hip_gates, dtype=torch.float32)
        for k, v in over.items():
            e32 = rel_l2(stock[k], exact[k])
            if not (v < 2.0 * e32 and v < 1e-4):
                bad[k] = ("vs exact arithmetic under the same gates", v, "stock fp32", e32)
    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/parity_golden_gradients_{fixture}.txt", "w") as f:
        f.write("# hip_vs_reference_fp32  exact(fp64, hip gates)_vs_reference_fp32  hip_vs_exact  parameter\n")
        for e_hip, e_exact, e_eh, name in sorted(rows, reverse=True):
            f.write(f"{e_hip:.3e} {e_exact:.3e} {e_eh:.3e} {name}\n")
    assert not bad, bad


@pytest.mark.parametrize("cfg_name,B,Tp,Tm,w_seed,b_seed", [("base", 4, 100, 870, 13, 23), ("scaled", 2, 60, 300, 14, 24)])
def test_gradient_gap_is_relu_gate_flips(cfg_name, B, Tp, Tm, w_seed, b_seed):
    """Why the end-to-end gradient gate is looser than the 1e-6 every backward kernel meets on its own: a ReLU is the one
    discontinuous operation on the path.  A pre-activation within rounding distance of zero is gated differently by two
    correct evaluations in different precision, and a flipped unit perturbs its own weight gradient and everything below
    it.  Made an assertion here, on the full-length base case:
      (1) the HIP path and the fp64 oracle disagree on the gate of only a handful of units out of ~1.4e7, and every one of
          them has |pre-activation| < 1e-5 in fp64 (measured: 1 unit of 1.37e7);
      (2) evaluating the fp64 oracle UNDER THE HIP PATH'S GATES makes every parameter gradient agree to FLIP_FREE_GATE --
          the flips are the entire gap;
      (3) flips aside, the HIP path's typical parameter error is of the order of stock fp32 torch's own (vs fp64).
    (2) is what test_forward_backward_vs_oracle asserts for every one of its cases; here it is shown together with (1) and (3)
    on the full-length base case and the scaled one."""
    from oracle import synth_batch, oracle_forward, oracle_loss, relu_gates
    from transformertts_amd import ops
    from transformertts_amd.loss import TransformerTTSLoss
    first = (cfg_name, B) == ("base", 4)
    cfg, m = _build(cfg_name, w_seed)
    batch = synth_batch(B, Tp, Tm, cfg["n_mels"], cfg["n_phon"], ragged=True, seed=b_seed)
    args = [batch[k].to("cuda") for k in ("phoneme", "melspec", "phoneme_lens", "melspec_lens")]
    m.train()
    hip_out = []
    ops._relu_observer = lambda y: hip_out.append((y.detach() > 0).cpu())
    try:
        out = m(*args)
    finally:
        ops._relu_observer = None
    TransformerTTSLoss(8.0).to("cuda")(out, args[1], args[3])["total"].backward()

    def oracle_grads(dtype, gates=None, record=False):
        from oracle import fill_state
        sd = fill_state(cfg, w_seed)
        for k in list(sd):
            if sd[k].is_floating_point():
                sd[k] = sd[k].to(dtype)
                if "running" not in k and k != "pe.pe":
                    sd[k].requires_grad_(True)
        with relu_gates(gates=gates) as rec:
            ref = oracle_forward(sd, cfg, batch["phoneme"], batch["melspec"].to(dtype), batch["phoneme_lens"],
                                 batch["melspec_lens"], training=True, dropout=False)
        oracle_loss(ref, batch["melspec"].to(dtype), batch["melspec_lens"])["total"].backward()
        return {k: v.grad for k, v in sd.items() if v.requires_grad}, rec.pre

    g64, pre64 = oracle_grads(torch.float64)
    assert len(pre64) == len(hip_out) == cfg["encoder_n_layers"] + 2 + cfg["decoder_n_layers"]
    flips, units = 0, 0
    for pre, gate in zip(pre64, hip_out):
        diff = (pre > 0) != gate.reshape(pre.shape)
        flips += int(diff.sum())
        units += pre.numel()
        if diff.any():
            assert float(pre[diff].abs().max()) < 1e-5, "a unit far from zero was gated differently"
    assert flips < 200 and units > 2e6, (flips, units)
    ggate, _ = oracle_grads(torch.float64, gates=[g.to(torch.float64) for g in hip_out])
    g32 = oracle_grads(torch.float32)[0] if first else g64          # claim (3) is checked on the first case only
    # Flip-free gate: 2e-5 on the base configuration, 5e-5 through the 12 layers of the scaled one -- the level stock fp32
    # torch itself reaches UNDER THE SAME GATES there (the oracle evaluated in fp32 with the HIP path's gates, against its
    # fp64 evaluation with the same gates: 0.4 - 5e-5 on pe.alpha, a single scalar summed over every position, and
    # 2.7e-5 on the post-net BatchNorm biases, depending on the run; written into the report below).
    FLIP_FREE_GATE = 2e-5
    g32gate = None
    if cfg_name != "base":
        g32gate, _ = oracle_grads(torch.float32, gates=[g.to(torch.float32) for g in hip_out])
    rows = []
    for name, p in m.named_parameters():
        if g64[name].norm().item() < 1e-7 * max(1.0, p.detach().norm().item()):
            continue
        rows.append((name, rel_l2(p.grad, ggate[name]), rel_l2(p.grad, g64[name]), rel_l2(g32[name], g64[name])))
    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/parity_gate_flips_{cfg_name}_B{B}.txt", "w") as f:
        f.write(f"# {flips} of {units} ReLU units gated differently by the HIP path and the fp64 oracle\n")
        f.write("# hip_vs_fp64_under_hip_gates  hip_vs_fp64  oracle_fp32_vs_fp64  parameter\n")
        for name, a, b, c in sorted(rows, key=lambda r: -r[2]):
            f.write(f"{a:.3e} {b:.3e} {c:.3e} {name}\n")
    if g32gate is None:
        bad = {n: a for n, a, _, _ in rows if not a < FLIP_FREE_GATE}
    else:
        stock = {n: rel_l2(g32gate[n], ggate[n]) for n, _, _, _ in rows}
        with open(f"gpurun_out/parity_gate_flips_{cfg_name}_B{B}.txt", "a") as f:
            f.write("# stock fp32 torch under the same gates vs fp64 under the same gates (worst five):\n")
            for n, v in sorted(stock.items(), key=lambda kv: -kv[1])[:5]:
                f.write(f"# {v:.3e} {n}\n")
        # (above the flat gate: twice what stock fp32 makes under the same gates, three times for a scalar -- see
        # test_forward_backward_vs_oracle)
        bad = {n: (a, stock[n]) for n, a, _, _ in rows
               if not a < 5e-5 and not a < (3.0 if ggate[n].numel() == 1 else 2.0) * stock[n]}
    assert not bad, bad
    # (3) with the flips taken out, the HIP path is as close to exact arithmetic as stock fp32 torch is to its own fp64
    # run (which of the two flips a unit in a given run is chance: their summation orders differ)
    med = lambda xs: sorted(xs)[len(xs) // 2]
    if first:
        assert med([a for _, a, _, _ in rows]) <= max(5.0 * med([c for _, _, _, c in rows]), 2e-6)


@pytest.mark.parametrize("mel_scale", [1e3, 1e-3])
def test_parity_holds_for_inputs_of_any_magnitude(mel_scale):
    """The reference runs in fp32 and has no operand-magnitude window; neither has this path: with the mel inputs (and
    targets) 1000 x larger or smaller than the normalised features the model expects, outputs and gradients still meet
    the gates of test_forward_backward_vs_oracle (the fp16x3 kernels take their pre-scales from measured maxima)."""
    from oracle import synth_batch, oracle_forward, oracle_loss
    from transformertts_amd.loss import TransformerTTSLoss
    cfg_name, B, Tp, Tm, w_seed, b_seed = "base", 2, 60, 300, 12, 22
    cfg, m = _build(cfg_name, w_seed)
    batch = synth_batch(B, Tp, Tm, cfg["n_mels"], cfg["n_phon"], ragged=True, seed=b_seed)
    batch["melspec"] = batch["melspec"] * mel_scale
    args = [batch[k].to("cuda") for k in ("phoneme", "melspec", "phoneme_lens", "melspec_lens")]
    from transformertts_amd import ops
    m.train()
    hip_gates = []
    ops._relu_observer = lambda y: hip_gates.append((y.detach() > 0).cpu())
    try:
        out = m(*args)
    finally:
        ops._relu_observer = None
    loss = TransformerTTSLoss(8.0).to("cuda")(out, args[1], args[3])
    loss["total"].backward()
    sd = _oracle64(cfg, w_seed)
    with torch.no_grad():
        ref = oracle_forward(sd, cfg, batch["phoneme"], batch["melspec"].double(), batch["phoneme_lens"], batch["melspec_lens"],
                             training=True, dropout=False)
        rloss = oracle_loss(ref, batch["melspec"].double(), batch["melspec_lens"])
    for k in ("pred_melspec", "post_melspec", "pred_stop"):
        assert torch.isfinite(out[k]).all() and rel_l2(out[k], ref[k]) < GATE, (k, rel_l2(out[k], ref[k]))
    for a, r in zip(out["alignments"], ref["alignments"]):
        assert rel_l2(a, r) < GATE
    assert abs(loss["total"].item() - rloss["total"].item()) < 1e-5 * abs(rloss["total"].item())
    # Gradients, gate-controlled (see test_forward_backward_vs_oracle): fp64 under the HIP path's ReLU gates is the reference.
    # Inputs 1000 x off the normalised features make the step itself ill-conditioned in fp32 (the decoder's first residual sums
    # add O(1) attention outputs to O(1000) pre-net activations), so next to the flip-free gate the yardstick is what stock
    # fp32 torch achieves on the same inputs under the same gates.  No parameter is exempt: at x 1000 decoder layer 0's
    # self-attention softmax saturates to exact one-hot rows, whose backward this path now takes as the exact zero torch gets
    # (csrc/attention.hip, `saturated`); round 3 waived 15 % on that layer's in-projection and on the pre-net.
    exact, _ = _oracle_gated_grads(cfg, w_seed, batch, hip_gates)
    stock, _ = _oracle_gated_grads(cfg, w_seed, batch, hip_gates, dtype=torch.float32)
    rows, bad = [], {}
    for name, p in m.named_parameters():
        rg = exact[name]
        if rg.norm().item() < 1e-7 * max(1.0, p.detach().norm().item()) * max(1.0, mel_scale ** 2):
            continue
        e, e32 = rel_l2(p.grad, rg), rel_l2(stock[name], rg)
        rows.append((e, e32, name))
        if not e < max(FLIP_FREE_GATE, 3.0 * e32):
            bad[name] = (e, e32)
    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/parity_mel_x{mel_scale:g}.txt", "w") as f:
        f.write("# gradients under the HIP path's ReLU gates: hip_vs_fp64  stock_fp32_vs_fp64  parameter\n")
        for e, e32, name in sorted(rows, reverse=True):
            f.write(f"{e:.3e} {e32:.3e} {name}\n")
    assert not bad, bad


def test_training_step_surface():
    """LightningModule.training_step: loss/gradients vs the oracle's restated step (dropout off, p_tf < 1 so the
    scheduled-sampling mix is exercised with an injected uniform draw)."""
    import transformertts_amd.utils.util as U
    from oracle import model_config, fill_state, synth_batch, oracle_training_step
    from transformertts_amd.lightning_module import LightningModule
    cfg = model_config("tiny")
    config = {"model": dict(cfg, device="cuda"), "loss": {"stop_weight": 8.0},
              "training": {"num_epochs": 300, "teacher_forcing_mode": "linear", "warmup_steps": 4000}}
    lm = LightningModule(config).to("cuda")
    lm.model.load_state_dict(fill_state(cfg, 11), strict=True)
    _no_dropout(lm)
    lm.train()
    lm.current_epoch = 120
    batch = synth_batch(3, 12, 40, cfg["n_mels"], cfg["n_phon"], ragged=True, seed=21)
    u = torch.rand(3, 1, batch["melspec"].size(1), generator=torch.Generator().manual_seed(5))
    U._uniform_draw = lambda B, T, device: u.to(device)       # inject the draw the oracle uses (test seam)
    try:
        loss = lm.training_step(dict(batch), 1)
    finally:
        U._uniform_draw = None
    loss.backward()
    sd = _oracle64(cfg, 11)
    b64 = dict(batch, melspec=batch["melspec"].double())
    rloss, _, _ = oracle_training_step(sd, cfg, b64, epoch=120, seed_u=u.double())
    rloss["total"].backward()
    assert abs(loss.item() - rloss["total"].item()) < 1e-4 * abs(rloss["total"].item())
    for name, p in lm.model.named_parameters():
        rg = sd[name].grad
        if rg.norm().item() < 1e-7:
            continue
        assert rel_l2(p.grad, rg) < GRAD_GATE, (name, rel_l2(p.grad, rg))
    for name, buf in lm.model.named_buffers():
        if "num_batches" in name:
            assert int(buf.item()) == 2


def test_twin_keywords_refuse_misuse():
    """`forward(memory=, postnet_twin=)` are for training_step's pair of forwards over ONE batch: a memory of another shape, a
    post-net box left by another batch, a box handed to a forward that wants the stop logits or to a grad forward first are
    refused with a ValueError naming what is wrong (never a silently wrong post-net pass)."""
    from oracle import model_config, synth_batch
    from transformertts_amd import ops
    from transformertts_amd.model import TransformerTTS
    cfg = model_config("base")
    m = TransformerTTS(**cfg, device="cuda").to("cuda").train()
    b = {k: v.to("cuda") for k, v in synth_batch(2, 16, 40, cfg["n_mels"], cfg["n_phon"], ragged=False, seed=3).items()}
    args = (b["phoneme"], b["melspec"], b["phoneme_lens"], b["melspec_lens"])
    with torch.no_grad():
        mem = m.encode(b["phoneme"], b["phoneme_lens"])
        with pytest.raises(ValueError, match="memory"):
            m(*args, memory=mem[:, :-1])
        with pytest.raises(ValueError, match="no-grad forward"):
            m(*args, postnet_twin=ops.PostnetTwin())                      # (need_stop defaults to True)
        box = ops.PostnetTwin()
        out = m(*args, need_alignments=False, need_stop=False, postnet_twin=box)
        assert out["post_melspec"] is None and out["pred_stop"] is None and box.full.shape == (4, 40, cfg["n_mels"])
        assert out["pred_melspec"].data_ptr() == box.full[2:].data_ptr()
    with pytest.raises(ValueError, match="no-grad forward"):
        m(*args, postnet_twin=ops.PostnetTwin())                          # an EMPTY box reaches a grad forward
    b3 = {k: v.to("cuda") for k, v in synth_batch(3, 16, 40, cfg["n_mels"], cfg["n_phon"], ragged=False, seed=4).items()}
    with pytest.raises(ValueError, match="another batch"):
        m(b3["phoneme"], b3["melspec"], b3["phoneme_lens"], b3["melspec_lens"], postnet_twin=box)
    out = m(*args, postnet_twin=box)                                      # the pair it was made for
    assert out["post_melspec"].shape == (2, 40, cfg["n_mels"]) and out["pred_melspec"].data_ptr() == box.full.data_ptr()
    out["post_melspec"].sum().backward()
    with torch.no_grad():                                                 # a whole pair without grad (a dry run of training_step)
        box2 = ops.PostnetTwin()
        m(*args, need_alignments=False, need_stop=False, postnet_twin=box2)
        out2 = m(*args, postnet_twin=box2)
    assert out2["post_melspec"] is not None and torch.isfinite(out2["post_melspec"]).all() and out2["pred_stop"] is not None


@pytest.mark.parametrize("flag,B,Tp,Tm", [("TWIN_ENCODER", 4, 64, 200), ("TWIN_ENCODER", 3, 37, 150), ("TWIN_POSTNET", 4, 64, 200),
                                          ("TWIN_POSTNET", 3, 37, 150), ("TWIN_POSTNET", 32, 50, 870)])
def test_twin_batches_are_the_two_forwards(flag, B, Tp, Tm):
    """model.encode_twin: the encoder of BOTH forwards of training_step as one pass over a batch of 2 B (the reference encodes the
    same phonemes twice, lightning_module.py:53-59,77).  With dropout off the step must be the step of two separate encodes:
    loss, every parameter gradient, the pre-net's BatchNorm running statistics (updated TWICE, by each forward's own batch
    statistics -- identical here, the inputs being identical) and num_batches_tracked.  (4, 64): the halves are whole row chunks
    of the convolution's BatchNorm partials; (3, 37): they are not, and each half takes its own statistics pass.  p_tf < 1: the
    no-grad forward's prediction -- built on the no-grad half of the twin batch -- enters the grad forward's input.
    TWIN_POSTNET (ops.PostnetTwin): the same for the post-net, which the reference runs in the no-grad forward for nothing but its
    BatchNorm running statistics (model/model.py:310, lightning_module.py:53-59) -- here the grad forward runs it once over both
    predictions; (32, 50, 870): its convolutions on the 224-row tile, whose row chunk the halves share."""
    import transformertts_amd.utils.util as U
    from oracle import model_config, fill_state, synth_batch
    from transformertts_amd import ops
    from transformertts_amd.lightning_module import LightningModule
    cfg = model_config("base")
    config = {"model": dict(cfg, device="cuda"), "loss": {"stop_weight": 8.0},
              "training": {"num_epochs": 300, "teacher_forcing_mode": "linear", "warmup_steps": 4000}}
    batch = synth_batch(B, Tp, Tm, cfg["n_mels"], cfg["n_phon"], ragged=True, seed=31)
    u = torch.rand(B, 1, batch["melspec"].size(1), generator=torch.Generator().manual_seed(6))
    res = []
    for twin in (True, False):
        setattr(ops, flag, twin)
        try:
            lm = LightningModule(config).to("cuda")
            lm.model.load_state_dict(fill_state(cfg, 12), strict=True)
            _no_dropout(lm)
            lm.train()
            lm.current_epoch = 120
            assert (lm.model.twin_encode_ok(batch["phoneme"].to("cuda")) if flag == "TWIN_ENCODER" else
                    lm.model.twin_postnet_ok(batch["melspec"].to("cuda"))) == twin
            U._uniform_draw = lambda B_, T_, device: u.to(device)
            try:
                loss = lm.training_step(dict(batch), 1)
            finally:
                U._uniform_draw = None
            loss.backward()
            torch.cuda.synchronize()
            res.append((loss.item(), {n: p.grad.clone() for n, p in lm.model.named_parameters()},
                        {n: b.clone() for n, b in lm.model.named_buffers() if "running" in n or "num_batches" in n}))
        finally:
            setattr(ops, flag, True)
    (l1, g1, b1), (l0, g0, b0) = res
    assert abs(l1 - l0) < 2e-6 * abs(l0), (l1, l0)
    for n in g0:
        if g0[n].norm().item() < 1e-7:
            continue
        assert rel_l2(g1[n], g0[n]) < 2e-5, (n, rel_l2(g1[n], g0[n]))
    for n in b0:
        if "num_batches" in n:
            assert int(b1[n]) == int(b0[n]) == 2, n
        else:
            assert rel_l2(b1[n], b0[n]) < 2e-6, (n, rel_l2(b1[n], b0[n]))


def test_grad_sinks_match_autograd():
    """Gradients written straight into the flat data-parallel bucket (accumulate=1 sinks) are bit-identical to the
    ones autograd returns without a bucket, and survive a second accumulation step as 2x."""
    from oracle import synth_batch
    from transformertts_amd.loss import TransformerTTSLoss
    from transformertts_amd.parallel import FlatGradBucket
    cfg, m1 = _build("tiny", 5)
    _, m2 = _build("tiny", 5)
    batch = synth_batch(3, 12, 40, cfg["n_mels"], cfg["n_phon"], ragged=True, seed=9)
    args = [batch[k].to("cuda") for k in ("phoneme", "melspec", "phoneme_lens", "melspec_lens")]
    crit = TransformerTTSLoss(8.0).to("cuda")
    m1.train(); m2.train()
    crit(m1(*args), args[1], args[3])["total"].backward()
    bucket = FlatGradBucket(m2.parameters())
    bucket.zero()
    crit(m2(*args), args[1], args[3])["total"].backward()
    for (n1, p1), (n2, p2) in zip(m1.named_parameters(), m2.named_parameters()):
        assert p2.grad.data_ptr() == p2._ttts_grad_sink.data_ptr(), n2
        assert torch.equal(p1.grad, p2.grad), n1
    # BN buffers advanced by one step in both replicas; run the bucketed one again without zeroing: grads add up
    for b1, b2 in zip(m1.buffers(), m2.buffers()):
        b2.copy_(b1) if b1.dtype.is_floating_point else None
    g1 = bucket.flat.clone()
    m3_state = {k: v.clone() for k, v in m1.state_dict().items()}
    m2.load_state_dict(m3_state)
    _, m4 = _build("tiny", 5)
    m4.load_state_dict(m3_state); m4.train()
    crit(m4(*args), args[1], args[3])["total"].backward()
    crit(m2(*args), args[1], args[3])["total"].backward()
    expect = g1 + torch.cat([torch.nn.functional.pad(p.grad.flatten(), (0, (-p.numel()) % 64)) for p in m4.parameters()])
    assert rel_l2(bucket.flat, expect) < 1e-6


def test_inference_kv_cache_vs_recompute_vs_oracle(golden_dir):
    """Incremental decoding with K/V caches == the reference-style full recomputation == the oracle loop (fp64) ==
    the reference's own inference output (golden)."""
    from oracle import synth_batch, oracle_inference
    g = np.load(os.path.join(golden_dir, "tiny_inference.npz"))
    cfg, m = _build(str(g["meta/cfg_name"]), int(g["meta/w_seed"]))
    batch = synth_batch(int(g["meta/B"]), int(g["meta/Tp"]), 40, cfg["n_mels"], cfg["n_phon"], ragged=True,
                        seed=int(g["meta/b_seed"]))
    ph, pl = batch["phoneme"].to("cuda"), batch["phoneme_lens"].to("cuda")
    L = int(g["meta/max_len"])
    fast = m.inference(ph, pl, max_len=L, stop_threshold=2.0, use_kv_cache=True)
    slow = m.inference(ph, pl, max_len=L, stop_threshold=2.0, use_kv_cache=False)
    ref = oracle_inference(_oracle64(cfg, int(g["meta/w_seed"])), cfg, batch["phoneme"], batch["phoneme_lens"], max_len=L,
                           stop_threshold=2.0)
    for k in ("pred_melspec", "post_melspec", "pred_stop"):
        assert fast[k].shape == slow[k].shape == tuple(g[k].shape), k
        assert rel_l2(fast[k], slow[k]) < 1e-5, (k, rel_l2(fast[k], slow[k]))
        assert rel_l2(fast[k], ref[k]) < GATE, (k, rel_l2(fast[k], ref[k]))
        assert rel_l2(fast[k], torch.from_numpy(g[k])) < GATE, k
    # early stop: a threshold every item passes at once ends the loop after the first frame
    one = m.inference(ph, pl, max_len=L, stop_threshold=0.0)
    assert one["pred_melspec"].shape[1] == 1 and one["pred_stop"].shape[1] == 1


def test_inference_with_heads_wider_than_64():
    """`inference()` of a model whose heads are 128 columns wide (tensor-algebra attention, the recompute loop: the incremental
    kernels read 64-wide heads) against the oracle loop in fp64."""
    from oracle import synth_batch, oracle_inference
    cfg, m = _build("tiny1h", 31)
    batch = synth_batch(2, 12, 40, cfg["n_mels"], cfg["n_phon"], ragged=True, seed=32)
    ph, pl = batch["phoneme"].to("cuda"), batch["phoneme_lens"].to("cuda")
    out = m.inference(ph, pl, max_len=10, stop_threshold=2.0)
    ref = oracle_inference(_oracle64(cfg, 31), cfg, batch["phoneme"], batch["phoneme_lens"], max_len=10, stop_threshold=2.0)
    for k in ("pred_melspec", "post_melspec", "pred_stop"):
        assert out[k].shape == tuple(ref[k].shape) and rel_l2(out[k], ref[k]) < GATE, (k, rel_l2(out[k], ref[k]))


@pytest.mark.parametrize("which", ["decoder", "encoder"])
def test_pre_norm_layers_match_torch(which):
    """norm_first=True (the other branch of the reference's decoder layer, model/layers.py:41-45, and of torch's encoder
    layer): output and gradients against torch's own layers evaluated in fp64 on the CPU with the same weights and masks."""
    from transformertts_amd.model.layers import TransformerDecoderLayer, TransformerEncoderLayer
    torch.manual_seed(3)
    d, H, dff, B, Tm, Tp = 128, 2, 256, 3, 50, 17
    x = torch.randn(B, Tm, d)
    mem = torch.randn(B, Tp, d)
    tl, ml = torch.tensor([50, 31, 8]), torch.tensor([17, 9, 3])
    gout = torch.randn(B, Tm, d)
    if which == "decoder":
        ours = TransformerDecoderLayer(d, H, dff, dropout=0.0, norm_first=True).to("cuda")
        ref = torch.nn.TransformerDecoderLayer(d, H, dff, dropout=0.0, norm_first=True, batch_first=True).double()
    else:
        ours = TransformerEncoderLayer(d, H, dff, dropout=0.0, norm_first=True).to("cuda")
        ref = torch.nn.TransformerEncoderLayer(d, H, dff, dropout=0.0, norm_first=True, batch_first=True).double()
    with torch.no_grad():
        for p in ours.parameters():
            p.copy_(torch.randn(p.shape) * (0.3 if p.dim() > 1 else 0.1) + (1.0 if "norm" in "" else 0.0))
    ref.load_state_dict({k: v.detach().cpu().double() for k, v in ours.state_dict().items()}, strict=True)
    ours.train(); ref.train()
    xg = x.to("cuda").requires_grad_()
    xr = x.double().requires_grad_()
    kpm = torch.arange(Tm)[None, :] >= tl[:, None]
    if which == "decoder":
        out, _ = ours(xg, mem.to("cuda"), tgt_lens=tl.to("cuda"), memory_lens=ml.to("cuda"), tgt_is_causal=True)
        causal = torch.triu(torch.ones(Tm, Tm), 1).bool()
        mkpm = torch.arange(Tp)[None, :] >= ml[:, None]
        want = ref(xr, mem.double(), tgt_mask=causal, tgt_key_padding_mask=kpm, memory_key_padding_mask=mkpm)
    else:
        out = ours(xg, tl.to("cuda"))
        want = ref(xr, src_key_padding_mask=kpm)
    valid = (~kpm).unsqueeze(-1)                      # rows past an utterance's length are padding on both sides
    out.backward(gout.to("cuda") * valid.to("cuda"))
    want.backward(gout.double() * valid)
    assert rel_l2(out * valid.to("cuda"), want * valid) < 1e-5, rel_l2(out * valid.to("cuda"), want * valid)
    assert rel_l2(xg.grad, xr.grad) < 1e-5, rel_l2(xg.grad, xr.grad)
    for (n, p), (_, q) in zip(ours.named_parameters(), ref.named_parameters()):
        if q.grad is None or q.grad.norm() < 1e-9:
            continue
        assert rel_l2(p.grad, q.grad) < 1e-4, (n, rel_l2(p.grad, q.grad))


def test_eval_mode_backward_through_batchnorm():
    """ConvNormBN in eval mode (running statistics are constants of the graph) is differentiable like the reference's: the
    gradients of the input, the convolution and the affine parameters against torch in fp64."""
    from transformertts_amd.model.module import ConvNormBN
    torch.manual_seed(5)
    B, T, cin, cout = 3, 41, 64, 96
    m = ConvNormBN(cin, cout, 5, activation="tanh").to("cuda")
    with torch.no_grad():
        m.bn.running_mean.copy_(torch.randn(cout) * 0.3)
        m.bn.running_var.copy_(torch.rand(cout) + 0.5)
        m.bn.weight.copy_(1 + 0.2 * torch.randn(cout)); m.bn.bias.copy_(0.1 * torch.randn(cout))
    m.eval()
    x = torch.randn(B, T, cin)
    xg = x.to("cuda").requires_grad_()
    g = torch.randn(B, T, cout)
    z = m.fused(xg, 2, 0.0)            # tanh epilogue
    z.backward(g.to("cuda"))
    conv = torch.nn.Conv1d(cin, cout, 5, padding=2).double()
    bn = torch.nn.BatchNorm1d(cout).double()
    conv.load_state_dict({k: v.detach().cpu().double() for k, v in m.conv.state_dict().items()})
    bn.load_state_dict({k: (v.detach().cpu().double() if v.is_floating_point() else v.detach().cpu()) for k, v in m.bn.state_dict().items()})
    conv.eval(); bn.eval()
    xr = x.double().requires_grad_()
    zr = torch.tanh(bn(conv(xr.transpose(1, 2)))).transpose(1, 2)
    zr.backward(g.double())
    assert rel_l2(z, zr) < 1e-5 and rel_l2(xg.grad, xr.grad) < 1e-5
    assert rel_l2(m.conv.weight.grad, conv.weight.grad) < 1e-5 and rel_l2(m.conv.bias.grad, conv.bias.grad) < 1e-5
    assert rel_l2(m.bn.weight.grad, bn.weight.grad) < 1e-5 and rel_l2(m.bn.bias.grad, bn.bias.grad) < 1e-5
    assert int(m.bn.num_batches_tracked) == 0


@pytest.mark.parametrize("d,H", [(96, 3), (64, 4), (48, 1)])
def test_attention_heads_narrower_than_64(d, H):
    """head_dim 32 / 16 / 48: the kernels work on 64-column heads, narrower ones are zero-padded copies; q is scaled by
    sqrt(1 / head_dim) as torch does.  Self- (causal) and cross-attention with weights against fp64."""
    import math
    from transformertts_amd import ops
    torch.manual_seed(1)
    B, T, Tk, hd = 2, 70, 23, d // H
    qkv = torch.randn(B, T, 3 * d)
    lens = torch.tensor([70, 33])
    do = torch.randn(B, T, d)

    def ref(q, k, v, kl, causal):
        q = q.double().view(B, -1, H, hd).transpose(1, 2); k = k.double().view(B, -1, H, hd).transpose(1, 2)
        v = v.double().view(B, -1, H, hd).transpose(1, 2)
        s = (q * math.sqrt(1.0 / hd)) @ k.transpose(-1, -2)
        mask = torch.arange(k.shape[2])[None, None, None, :] >= kl[:, None, None, None]
        if causal:
            mask = mask | (torch.arange(k.shape[2])[None, :] > torch.arange(q.shape[2])[:, None])
        p = torch.softmax(s.masked_fill(mask, float("-inf")), -1)
        return (p @ v).transpose(1, 2).reshape(B, -1, d), p
    qg = qkv.to("cuda").requires_grad_()
    o = ops.self_attention(qg, lens.to("cuda"), H, True, 0.0, 0)
    o.backward(do.to("cuda"))
    qr = qkv.double().requires_grad_()
    orf, _ = ref(qr[..., :d], qr[..., d:2 * d], qr[..., 2 * d:], lens, True)
    orf.backward(do.double())
    assert rel_l2(o, orf) < 5e-6 and rel_l2(qg.grad, qr.grad) < 2e-5, (rel_l2(o, orf), rel_l2(qg.grad, qr.grad))
    q, kv = torch.randn(B, T, d), torch.randn(B, Tk, 2 * d)
    kl = torch.tensor([23, 7])
    q1, kv1 = q.to("cuda").requires_grad_(), kv.to("cuda").requires_grad_()
    o2, attn = ops.cross_attention(q1, kv1, kl.to("cuda"), H, 0.0, 0, True)
    o2.backward(do.to("cuda"))
    q2, kv2 = q.double().requires_grad_(), kv.double().requires_grad_()
    o2r, pr = ref(q2, kv2[..., :d], kv2[..., d:], kl, False)
    o2r.backward(do.double())
    assert rel_l2(o2, o2r) < 5e-6 and rel_l2(attn, pr) < 5e-6
    assert rel_l2(q1.grad, q2.grad) < 2e-5 and rel_l2(kv1.grad, kv2.grad) < 2e-5


@pytest.mark.parametrize("cfg_name,B,Tp,Tm,w_seed,b_seed", [("base", 2, 60, 300, 12, 22), ("scaled", 2, 60, 300, 14, 24)])
def test_image_operand_path_on_small_shapes(monkeypatch, cfg_name, B, Tp, Tm, w_seed, b_seed):
    """The image-operand GEMMs (csrc/gemm_h3i.hip: LayerNorm leaves its output / its gradient already split into f16 hi / lo with
    a per-row scale, the in-projections, FFN1 and the data gradients behind LayerNorm stage both operands by LDS-DMA) take over
    from 8 192 rows on; here the threshold is lowered so that the whole model runs on them at a size the fp64 oracle finishes
    in seconds, and held to the same gates as the default path.  (At full size the same kernels run inside
    test_forward_backward_vs_oracle[base-16-100-870] and the batch-64 property tests.)"""
    from transformertts_amd import ops
    monkeypatch.setattr(ops, "IMAGE_MIN_ROWS", 1)
    monkeypatch.setattr(ops, "LAYERNORM_IMAGES", True)
    calls = {"fwd": 0, "bwd": 0}
    from transformertts_amd import _lib
    lib = _lib.load()
    f0, b0 = lib.ttts_linear_fwd_h3i, lib.ttts_linear_bwd_data_h3i
    monkeypatch.setattr(lib, "ttts_linear_fwd_h3i", lambda *a: (calls.__setitem__("fwd", calls["fwd"] + 1), f0(*a))[1])
    monkeypatch.setattr(lib, "ttts_linear_bwd_data_h3i", lambda *a: (calls.__setitem__("bwd", calls["bwd"] + 1), b0(*a))[1])
    test_forward_backward_vs_oracle(cfg_name, B, Tp, Tm, w_seed, b_seed)
    assert calls["fwd"] > 0 and calls["bwd"] > 0, calls


def test_dma_gemm_on_plain_fp32_operands_on_small_shapes(monkeypatch):
    """ttts_linear_fwd_h3d / ttts_linear_bwd_data_h3d -- the LDS-DMA kernel on a plain fp32 activation (raw k-tile staged by DMA,
    split in place in LDS) -- are what the step uses for the gated data gradients from 8 192 rows on and for the square projections
    whose tiles fill the chip (ops._dma_shape_ok); with the thresholds lowered the whole base model runs through them against
    the fp64 oracle."""
    from transformertts_amd import _lib, ops
    monkeypatch.setattr(ops, "IMAGE_MIN_ROWS", 1)
    monkeypatch.setattr(ops, "DMA_MIN_TILES", 0)
    calls = {"fwd": 0, "bwd": 0}
    lib = _lib.load()
    f0, b0 = lib.ttts_linear_fwd_h3d, lib.ttts_linear_bwd_data_h3d
    monkeypatch.setattr(lib, "ttts_linear_fwd_h3d", lambda *a: (calls.__setitem__("fwd", calls["fwd"] + 1), f0(*a))[1])
    monkeypatch.setattr(lib, "ttts_linear_bwd_data_h3d", lambda *a: (calls.__setitem__("bwd", calls["bwd"] + 1), b0(*a))[1])
    test_forward_backward_vs_oracle("base", 2, 60, 300, 12, 22)
    assert calls["fwd"] > 0 and calls["bwd"] > 0, calls
