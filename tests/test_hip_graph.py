"""The training step as one captured HIP graph (transformertts_amd/step.py): a replayed step must be the SAME arithmetic
as the eager step -- bit for bit, dropout on -- including everything that changes per step and therefore has to be read
from device memory by the kernels (dropout seed word, learning rate, Adam step count, teacher-forcing ratio)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(cfg_name, w_seed, epoch):
    from oracle import fill_state
    from transformertts_amd.lightning_module import LightningModule
    from transformertts_amd.workload import model_config
    cfg = model_config(cfg_name)
    config = {"model": dict(cfg, device="cuda"), "loss": {"stop_weight": 8.0},
              "training": {"num_epochs": 300, "teacher_forcing_mode": "linear", "warmup_steps": 50,
                           "sync_loss_every_step": False, "fused_clip_norm": 1.0}}
    lm = LightningModule(config).to("cuda")
    lm.model.load_state_dict(fill_state(cfg, w_seed), strict=True)
    lm.train()
    lm.current_epoch = epoch
    oc = lm.configure_optimizers()
    return cfg, lm, oc["optimizer"], oc["lr_scheduler"]["scheduler"]


@pytest.mark.parametrize("cfg_name,B,Tp,Tm,epoch", [("tiny", 3, 12, 40, 150), ("base", 4, 60, 300, 0)])
def test_graph_replay_equals_eager_bitwise(cfg_name, B, Tp, Tm, epoch):
    """Six optimizer steps (dropout ON, scheduled sampling active at epoch 150: p_tf = 0.52) eagerly and as
    2 eager + 4 replayed steps from the same initial state: identical parameters, Adam moments, BatchNorm buffers
    and per-step losses, bit for bit."""
    from transformertts_amd.step import TrainStep
    from transformertts_amd.workload import synth_batch
    runs = []
    for graph in (False, True):
        cfg, lm, opt, sch = _setup(cfg_name, 3, epoch)
        batch = {k: v.to("cuda") for k, v in synth_batch(B, Tp, Tm, cfg["n_mels"], cfg["n_phon"], ragged=True, seed=8).items()}
        ts = TrainStep(lm, opt, sch, batch, graph=graph, seed=77)
        losses = [ts().detach().clone() for _ in range(6)]
        torch.cuda.synchronize()
        assert ts.graphed == graph
        runs.append((losses, opt.flat_params.clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone(),
                     {k: v.clone() for k, v in lm.model.state_dict().items() if "running" in k or "num_batches" in k},
                     opt._step, opt.param_groups[0]["lr"]))
    (l0, p0, m0, v0, bn0, s0, lr0), (l1, p1, m1, v1, bn1, s1, lr1) = runs
    assert s0 == s1 == 6 and lr0 == lr1
    for a, b in zip(l0, l1):
        assert torch.isfinite(a) and torch.equal(a, b)
    assert len({float(x) for x in l0}) == 6                       # every step drew fresh masks / saw new parameters
    assert torch.equal(p0, p1) and torch.equal(m0, m1) and torch.equal(v0, v1)
    for k in bn0:
        assert torch.equal(bn0[k], bn1[k]), k
    assert int(bn0[[k for k in bn0 if "num_batches" in k][0]]) == 12     # two train-mode forwards per step


def test_graph_step_accepts_new_batches_of_the_same_shape():
    """The graph reads the batch from static buffers: `ts(batch)` copies the next batch in; a different shape is refused."""
    from transformertts_amd.step import TrainStep
    from transformertts_amd.workload import synth_batch
    cfg, lm, opt, sch = _setup("tiny", 4, 0)
    mk = lambda seed: {k: v.to("cuda") for k, v in synth_batch(3, 12, 40, cfg["n_mels"], cfg["n_phon"], ragged=False, seed=seed).items()}
    ts = TrainStep(lm, opt, sch, mk(1), graph=True, seed=5)
    for _ in range(3):
        ts()
    a = ts(mk(2)).item()
    cfg, lm2, opt2, sch2 = _setup("tiny", 4, 0)
    ts2 = TrainStep(lm2, opt2, sch2, mk(1), graph=False, seed=5)
    for _ in range(3):
        ts2()
    b = ts2(mk(2)).item()
    assert a == b
    with pytest.raises(ValueError, match="shape"):
        ts({k: v[:2] for k, v in mk(3).items()})


def test_deferred_reductions_equal_immediate_bitwise(monkeypatch):
    """The parameter-gradient reductions queued during backward and run as one batched launch at its end
    (ops._defer / ttts_reduce_defer_*) sum in the same order as the per-parameter launches: identical state after three
    steps, eager and replayed, and nothing is left queued."""
    from transformertts_amd import _lib, ops
    from transformertts_amd.step import TrainStep
    from transformertts_amd.workload import synth_batch
    lib = _lib.load()
    runs = []
    for defer, graph in ((False, False), (True, False), (True, True)):
        monkeypatch.setattr(ops, "DEFER_REDUCE", defer)
        cfg, lm, opt, sch = _setup("base", 3, 0)
        batch = {k: v.to("cuda") for k, v in synth_batch(3, 40, 200, cfg["n_mels"], cfg["n_phon"], ragged=True, seed=8).items()}
        ts = TrainStep(lm, opt, sch, batch, graph=graph, seed=9)
        losses = [ts().detach().clone() for _ in range(4)]
        torch.cuda.synchronize()
        assert lib.ttts_reduce_defer_pending() == 0 and not ops._defer_keep and not ops._defer_armed
        runs.append((losses, opt.flat_params.clone(), opt.exp_avg_sq.clone()))
    for losses, p, v in runs[1:]:
        assert all(torch.equal(a, b) for a, b in zip(losses, runs[0][0]))
        assert torch.equal(p, runs[0][1]) and torch.equal(v, runs[0][2])


def test_deferred_reduction_queue_semantics():
    """Entry points called with accumulate bit 1 queue only while deferral is open; a second reduction into the same
    destination is ordered behind the first; abort drops the queue."""
    from transformertts_amd import _lib, ops
    lib = _lib.load()
    g = torch.Generator(device="cuda").manual_seed(1)
    M, N, K = 4096, 256, 256
    dy = torch.randn(M, N, device="cuda", generator=g) * 1e-3
    x = torch.randn(M, K, device="cuda", generator=g)
    nb = lib.ttts_wgrad_workspace_bytes(M, N, K, 1)
    st = ops._stream()

    def call(dw, db, ws, acc):
        _lib.check(lib.ttts_linear_bwd_weight(ops._p(dy), ops._p(x), ops._p(dw), ops._p(db), ops._p(ws), ws.numel() * 4, M, N, K,
                                              0, 0, acc, st), "ttts_linear_bwd_weight")
    ref_w, ref_b = torch.zeros(N, K, device="cuda"), torch.zeros(N, device="cuda")
    call(ref_w, ref_b, ops._ws(nb, "cuda"), 0)
    # bit 1 without an open deferral: immediate
    w1, b1 = torch.zeros(N, K, device="cuda"), torch.zeros(N, device="cuda")
    call(w1, b1, ops._ws(nb, "cuda"), 2)
    assert lib.ttts_reduce_defer_pending() == 0 and torch.equal(w1, ref_w) and torch.equal(b1, ref_b)
    # open: queued (twice into the same destination, accumulating), nothing written before the flush
    w2, b2 = torch.zeros(N, K, device="cuda"), torch.zeros(N, device="cuda")
    wsa, wsb = ops._ws(nb, "cuda"), ops._ws(nb, "cuda")
    _lib.check(lib.ttts_reduce_defer_begin(), "begin")
    try:
        call(w2, b2, wsa, 3)
        call(w2, b2, wsb, 3)
        assert lib.ttts_reduce_defer_pending() == 4
        torch.cuda.synchronize()
        assert float(w2.abs().max()) == 0.0
        _lib.check(lib.ttts_reduce_defer_flush(0, st), "flush")
    finally:
        lib.ttts_reduce_defer_abort()
    assert lib.ttts_reduce_defer_pending() == 0
    assert torch.equal(w2, ref_w + ref_w) and torch.equal(b2, ref_b + ref_b)
    # abort drops what is queued
    w3, b3 = torch.zeros(N, K, device="cuda"), torch.zeros(N, device="cuda")
    _lib.check(lib.ttts_reduce_defer_begin(), "begin")
    call(w3, b3, wsa, 3)
    assert lib.ttts_reduce_defer_pending() == 2
    lib.ttts_reduce_defer_abort()
    assert lib.ttts_reduce_defer_pending() == 0
    call(w3, b3, wsa, 1)
    assert torch.equal(w3, ref_w)
