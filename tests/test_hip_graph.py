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
    """The graph reads the batch from static buffers: `ts(batch)` copies the next batch of the same shape in."""
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
    assert a == b and ts.n_graphs == 1


def _state(lm, opt):
    return (opt.flat_params.clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone(),
            {k: v.clone() for k, v in lm.model.state_dict().items() if "running" in k or "num_batches" in k})


def _same(a, b):
    return (torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
            and all(torch.equal(a[3][k], b[3][k]) for k in a[3]))


def test_shape_keyed_graph_cache_equals_eager_bitwise():
    """A stream of differently shaped ragged batches (what the reference's collate_fn produces, dataset.py:71-103): every
    shape gets its own captured graph (static buffers + graph, all graphs in one memory pool) and a cached-graph step on
    batch A after batch B is the same arithmetic as the eager step -- parameters, moments, BatchNorm buffers and losses
    identical bit for bit over 12 steps cycling through four shapes."""
    from transformertts_amd.step import TrainStep
    from transformertts_amd.workload import synth_batch
    shapes = [(3, 12, 40), (3, 10, 33), (2, 12, 40), (3, 9, 25)]
    runs = []
    for graph in (False, True):
        cfg, lm, opt, sch = _setup("tiny", 3, 150)
        batches = [{k: v.to("cuda") for k, v in synth_batch(B, Tp, Tm, cfg["n_mels"], cfg["n_phon"], ragged=True, seed=20 + i).items()}
                   for i, (B, Tp, Tm) in enumerate(shapes)]
        ts = TrainStep(lm, opt, sch, graph=graph, seed=11)
        losses = [ts(batches[i % 4]).detach().clone() for i in range(12)]
        torch.cuda.synchronize()
        if graph:
            assert ts.n_graphs == 4           # steps 0-1 ran eagerly; every shape was captured at its first use after them
        runs.append((losses, _state(lm, opt)))
    assert all(torch.equal(a, b) for a, b in zip(runs[0][0], runs[1][0]))
    assert _same(runs[0][1], runs[1][1])


def test_lattice_rounds_shapes_and_padding_is_inert_for_masked_outputs():
    """`lattice=(p, m)` pads a batch to multiples of p phonemes / m frames: three ragged batches of different maxima share
    ONE graph, and the loss -- masked by the true lengths -- is finite and sees the same valid frames."""
    from transformertts_amd.step import TrainStep
    from transformertts_amd.workload import synth_batch
    cfg, lm, opt, sch = _setup("tiny", 3, 0)
    ts = TrainStep(lm, opt, sch, graph=True, seed=2, lattice=(8, 16))
    for i, (Tp, Tm) in enumerate([(12, 40), (10, 35), (14, 44), (9, 33), (11, 47), (13, 41)]):
        b = {k: v.to("cuda") for k, v in synth_batch(3, Tp, Tm, cfg["n_mels"], cfg["n_phon"], ragged=True, seed=i).items()}
        loss = ts(b)
        assert ts.batch["melspec"].shape[1] == 48 and ts.batch["phoneme"].shape[1] == 16
        assert torch.equal(ts.batch["melspec"][:, :Tm], b["melspec"]) and float(ts.batch["melspec"][:, Tm:].abs().max()) == 0.0
        assert torch.isfinite(loss)
    assert ts.n_graphs == 1 and len(ts._slots) == 1


def test_gradient_accumulation_graphs_equal_eager_bitwise():
    """accumulate=3 (Lightning's accumulate_grad_batches, train.py:42): gradients of three micro-batches, each scaled by
    1/3, are summed in the bucket and the optimizer steps on every third call -- eagerly and as first / mid / last graphs."""
    from transformertts_amd.step import TrainStep
    from transformertts_amd.workload import synth_batch
    runs = []
    for graph in (False, True):
        cfg, lm, opt, sch = _setup("tiny", 5, 0)
        batch = {k: v.to("cuda") for k, v in synth_batch(3, 12, 40, cfg["n_mels"], cfg["n_phon"], ragged=True, seed=8).items()}
        ts = TrainStep(lm, opt, sch, batch, graph=graph, seed=3, accumulate=3)
        losses = [ts().detach().clone() for _ in range(9)]
        torch.cuda.synchronize()
        assert opt._step == 3
        if graph:
            assert ts.n_graphs == 3
        runs.append((losses, _state(lm, opt)))
    assert all(torch.equal(a, b) for a, b in zip(runs[0][0], runs[1][0]))
    assert _same(runs[0][1], runs[1][1])
    # the accumulated gradient is the MEAN of the micro-batch gradients: with dropout off and the same batch three times it
    # is the gradient of one pass (BatchNorm buffers aside, the three passes are the same arithmetic)
    grads = []
    for k in (1, 3):
        cfg, lm, opt, sch = _setup("tiny", 5, 0)
        for mod in lm.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
            if isinstance(getattr(mod, "dropout", None), float):
                mod.dropout = 0.0
        opt.param_groups[0]["lr"] = 0.0
        batch = {kk: v.to("cuda") for kk, v in synth_batch(3, 12, 40, cfg["n_mels"], cfg["n_phon"], ragged=True, seed=8).items()}
        ts = TrainStep(lm, opt, sch, batch, graph=False, seed=3, accumulate=k)
        for _ in range(k):
            ts()
        torch.cuda.synchronize()
        grads.append(opt.bucket.flat.clone())
    assert float((grads[0] - grads[1]).norm() / grads[0].norm()) < 1e-5


def test_capture_after_an_eval_forward_still_refreshes_the_weight_planes():
    """Regression (round-2 advisor): a forward pass between the last eager optimizer step and the capture used to refresh
    the cached weight splits, so the capture recorded no split kernel and every replay multiplied with planes frozen at
    capture time.  The step now refreshes its planes explicitly inside the graph: 2 eager steps, an eval forward, capture,
    4 replays == 6 eager steps."""
    from transformertts_amd.step import TrainStep
    from transformertts_amd.workload import synth_batch
    runs = []
    for graph in (False, True):
        cfg, lm, opt, sch = _setup("tiny", 3, 0)
        batch = {k: v.to("cuda") for k, v in synth_batch(3, 12, 40, cfg["n_mels"], cfg["n_phon"], ragged=True, seed=8).items()}
        ts = TrainStep(lm, opt, sch, batch, graph=graph, seed=77)
        ts(); ts()
        with torch.no_grad():                    # e.g. a validation batch: touches every forward weight plane
            lm.model.eval()
            lm.model(batch["phoneme"], batch["melspec"], batch["phoneme_lens"], batch["melspec_lens"])
            lm.model.train()
        if graph:
            ts.ensure_captured()
        for _ in range(4):
            ts()
        with torch.no_grad():                    # and an eager forward right after replays sees current planes as well
            lm.model.eval()
            out = lm.model(batch["phoneme"], batch["melspec"], batch["phoneme_lens"], batch["melspec_lens"])["pred_melspec"].clone()
            lm.model.train()
        torch.cuda.synchronize()
        runs.append((_state(lm, opt), out))
    assert _same(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])


def test_two_models_in_one_process_keep_their_own_plane_tables():
    """Regression (round-2 advisor): the captured graph must not read a process-wide descriptor table that another model
    can replace.  Two TrainSteps interleaved, the first model deleted half-way: the survivor still equals its eager run."""
    import gc
    from transformertts_amd.step import TrainStep
    from transformertts_amd.workload import synth_batch
    cfg, lmA, optA, schA = _setup("tiny", 3, 0)
    _, lmB, optB, schB = _setup("tiny", 4, 0)
    _, lmR, optR, schR = _setup("tiny", 4, 0)
    batch = {k: v.to("cuda") for k, v in synth_batch(3, 12, 40, cfg["n_mels"], cfg["n_phon"], ragged=True, seed=8).items()}
    tsA = TrainStep(lmA, optA, schA, batch, graph=True, seed=1)
    tsB = TrainStep(lmB, optB, schB, batch, graph=True, seed=2)
    tsR = TrainStep(lmR, optR, schR, batch, graph=False, seed=2)
    for _ in range(3):
        tsA(); tsB(); tsR()
    del tsA, lmA, optA, schA
    gc.collect()
    torch.cuda.empty_cache()
    for _ in range(3):
        tsB(); tsR()
    torch.cuda.synchronize()
    assert _same(_state(lmB, optB), _state(lmR, optR))


def test_deferred_reductions_equal_immediate_bitwise(monkeypatch):
    """The parameter-gradient reductions queued during backward and run as one batched launch when the bucket is flushed
    (ops.ReduceQueue / ttts_reduce_queue_*) sum in the same order as the per-parameter launches: identical state after
    four steps, eager and replayed, and nothing is left queued."""
    from transformertts_amd import ops
    from transformertts_amd.step import TrainStep
    from transformertts_amd.workload import synth_batch
    runs = []
    for defer, graph in ((False, False), (True, False), (True, True)):
        monkeypatch.setattr(ops, "DEFER_REDUCE", defer)
        cfg, lm, opt, sch = _setup("base", 3, 0)
        batch = {k: v.to("cuda") for k, v in synth_batch(3, 40, 200, cfg["n_mels"], cfg["n_phon"], ragged=True, seed=8).items()}
        ts = TrainStep(lm, opt, sch, batch, graph=graph, seed=9)
        losses = [ts().detach().clone() for _ in range(4)]
        torch.cuda.synchronize()
        assert opt.bucket.queue.pending() == 0 and not opt.bucket.queue.keep
        runs.append((losses, opt.flat_params.clone(), opt.exp_avg_sq.clone()))
    for losses, p, v in runs[1:]:
        assert all(torch.equal(a, b) for a, b in zip(losses, runs[0][0]))
        assert torch.equal(p, runs[0][1]) and torch.equal(v, runs[0][2])


def test_failed_backward_does_not_poison_the_next_step():
    """Regression (round-2 advisor): a backward pass that raises leaves reductions queued; the next zero_grad drops them
    (the queue belongs to the bucket) and training goes on -- under plain `optimizer.zero_grad(); loss.backward();
    optimizer.step()` as Lightning's automatic optimisation drives it, without TrainStep."""
    from transformertts_amd.workload import synth_batch
    cfg, lm, opt, sch = _setup("tiny", 3, 0)
    batch = {k: v.to("cuda") for k, v in synth_batch(3, 12, 40, cfg["n_mels"], cfg["n_phon"], ragged=True, seed=8).items()}

    def boom(grad):
        raise RuntimeError("boom")
    opt.zero_grad()
    out = lm.model(batch["phoneme"], batch["melspec"], batch["phoneme_lens"], batch["melspec_lens"])
    out["pred_melspec"].register_hook(boom)       # fires once the post-net gradients are queued, before the decoder's
    with pytest.raises(RuntimeError, match="boom"):
        lm.criterion(out, batch["melspec"], batch["melspec_lens"])["total"].backward()
    assert opt.bucket.queue.pending() > 0
    opt.zero_grad()
    assert opt.bucket.queue.pending() == 0
    loss2 = lm.training_step(batch, 0)
    loss2.backward()
    opt.step()
    torch.cuda.synchronize()
    assert opt.bucket.queue.pending() == 0
    assert float(opt.bucket.flat.abs().max()) > 0.0 and torch.isfinite(opt.flat_params).all()


def test_reduce_queue_semantics_through_the_c_abi():
    """A NULL queue launches the reduction at once; with a caller-owned queue nothing is written before the flush, a second
    reduction into the same destination is ordered behind the first, and clear drops what is queued."""
    from transformertts_amd import _lib, ops
    lib = _lib.load()
    g = torch.Generator(device="cuda").manual_seed(1)
    M, N, K = 4096, 256, 256
    dy = torch.randn(M, N, device="cuda", generator=g) * 1e-3
    x = torch.randn(M, K, device="cuda", generator=g)
    nb = lib.ttts_wgrad_workspace_bytes(M, N, K, 1)
    st = ops._stream()

    def call(dw, db, ws, acc, q):
        _lib.check(lib.ttts_linear_bwd_weight(ops._p(dy), ops._p(x), ops._p(dw), ops._p(db), ops._p(ws), ws.numel() * 4, M, N, K,
                                              0, 0, acc, q, st), "ttts_linear_bwd_weight")
    ref_w, ref_b = torch.zeros(N, K, device="cuda"), torch.zeros(N, device="cuda")
    call(ref_w, ref_b, ops._ws(nb, "cuda"), 0, None)
    q = ops.ReduceQueue()       # (q.arg() also registers an engine callback and is for backward nodes: use the handle here)
    w2, b2 = torch.zeros(N, K, device="cuda"), torch.zeros(N, device="cuda")
    wsa, wsb = ops._ws(nb, "cuda"), ops._ws(nb, "cuda")
    call(w2, b2, wsa, 1, q.handle)
    call(w2, b2, wsb, 1, q.handle)
    assert q.pending() == 4
    torch.cuda.synchronize()
    assert float(w2.abs().max()) == 0.0
    q.flush()
    assert q.pending() == 0 and not q.keep
    assert torch.equal(w2, ref_w + ref_w) and torch.equal(b2, ref_b + ref_b)
    w3, b3 = torch.zeros(N, K, device="cuda"), torch.zeros(N, device="cuda")
    call(w3, b3, wsa, 1, q.handle)
    assert q.pending() == 2
    q.clear()
    assert q.pending() == 0
    call(w3, b3, wsa, 1, None)
    assert torch.equal(w3, ref_w)


def test_two_host_threads_on_two_streams_run_backward_concurrently():
    """The C ABI keeps no state between calls (include/ttts_hip.h): two host threads, each with its own model, gradient
    bucket (and therefore reduction queue) and stream, run forward + backward + optimizer concurrently and end in exactly
    the state their serial runs end in."""
    import threading
    from transformertts_amd.workload import synth_batch

    def build(seed):
        cfg, lm, opt, sch = _setup("tiny", seed, 0)
        for mod in lm.modules():                  # dropout off: the site-seed counter is process-wide host state
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
            if hasattr(mod, "dropout") and isinstance(getattr(mod, "dropout"), float):
                mod.dropout = 0.0
        batch = {k: v.to("cuda") for k, v in synth_batch(3, 12, 40, cfg["n_mels"], cfg["n_phon"], ragged=True, seed=seed).items()}
        return lm, opt, batch

    def run(lm, opt, batch, stream, steps=3):
        with torch.cuda.stream(stream):
            for _ in range(steps):
                opt.zero_grad()
                lm.training_step(batch, 0).backward()
                opt.step()
        stream.synchronize()
    serial = []
    for seed in (3, 4):
        lm, opt, batch = build(seed)
        run(lm, opt, batch, torch.cuda.current_stream())
        serial.append(_state(lm, opt))
    models = [build(3), build(4)]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    errs = []

    def worker(i):
        try:
            run(*models[i], streams[i])
        except Exception as e:  # noqa: BLE001
            errs.append(e)
    threads = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errs, errs
    for i in range(2):
        assert _same(_state(models[i][0], models[i][1]), serial[i])


def test_mid_and_last_graphs_record_no_weight_splits(monkeypatch):
    """accumulate > 1: only the window's FIRST micro-batch refreshes the weight planes (one batched launch).  The graphs of
    the later micro-batches must not carry per-weight `ttts_weight_split` launches (round 3 baked ~200 of them into every
    mid / last graph: after any capture the host-side plane tags read 'stale', and inside a capture the batched refresh is
    refused).  Counted at the C ABI while the graphs are being captured."""
    from transformertts_amd import _lib
    from transformertts_amd.step import TrainStep
    from transformertts_amd.workload import synth_batch
    cfg, lm, opt, sch = _setup("tiny", 5, 0)
    batch = {k: v.to("cuda") for k, v in synth_batch(3, 12, 40, cfg["n_mels"], cfg["n_phon"], ragged=True, seed=8).items()}
    ts = TrainStep(lm, opt, sch, batch, graph=True, seed=3, accumulate=3)
    lib = _lib.load()
    calls = {"single": 0, "batched": 0}
    single, batched = lib.ttts_weight_split, lib.ttts_weight_split_batched

    def count_single(*a):
        if torch.cuda.is_current_stream_capturing():
            calls["single"] += 1
        return single(*a)

    def count_batched(*a):
        if torch.cuda.is_current_stream_capturing():
            calls["batched"] += 1
        return batched(*a)

    monkeypatch.setattr(lib, "ttts_weight_split", count_single)
    monkeypatch.setattr(lib, "ttts_weight_split_batched", count_batched)
    for _ in range(9):
        ts()
    torch.cuda.synchronize()
    assert ts.n_graphs == 3
    assert calls == {"single": 0, "batched": 1}, calls


def test_load_state_dict_on_a_live_trainstep_drops_its_graphs():
    """A parameter edited behind the plane table (load_state_dict bumps the version counters) invalidates the table whose
    device descriptors the captured graphs recorded: the graphs must be dropped and re-captured against the new table, and the
    run must continue exactly as an eager run that loads the same state at the same step."""
    from oracle import fill_state
    from transformertts_amd.step import TrainStep
    from transformertts_amd.workload import synth_batch
    outs = []
    for graph in (False, True):
        cfg, lm, opt, sch = _setup("tiny", 5, 0)
        batch = {k: v.to("cuda") for k, v in synth_batch(3, 12, 40, cfg["n_mels"], cfg["n_phon"], ragged=True, seed=8).items()}
        ts = TrainStep(lm, opt, sch, batch, graph=graph, seed=3)
        for _ in range(4):
            ts()
        lm.model.load_state_dict(fill_state(cfg, 11), strict=True)
        losses = [ts().detach().clone() for _ in range(3)]
        torch.cuda.synchronize()
        if graph:
            assert ts.recaptures == 1 and ts.n_graphs == 1
        outs.append((losses, _state(lm, opt)))
    assert all(torch.equal(a, b) for a, b in zip(outs[0][0], outs[1][0]))
    assert _same(outs[0][1], outs[1][1])


def test_graph_cache_evicts_the_least_recently_used_shape():
    """max_shapes = 2 and three shapes round-robin: every new shape evicts the least recently used one (graphs destroyed,
    static buffers freed) instead of aborting training; the arithmetic stays that of the eager run."""
    from transformertts_amd.step import TrainStep
    from transformertts_amd.workload import synth_batch
    outs = []
    for graph in (False, True):
        cfg, lm, opt, sch = _setup("tiny", 5, 0)
        mk = lambda tm, seed: {k: v.to("cuda") for k, v in synth_batch(3, 12, tm, cfg["n_mels"], cfg["n_phon"], ragged=True, seed=seed).items()}
        batches = [mk(40, 1), mk(36, 2), mk(44, 3)]
        ts = TrainStep(lm, opt, sch, batches[0], graph=graph, seed=3, max_shapes=2)
        losses = [ts(batches[i % 3]).detach().clone() for i in range(9)]
        torch.cuda.synchronize()
        if graph:
            assert ts.evictions >= 6 and len(ts._slots) == 2
        outs.append((losses, _state(lm, opt)))
    assert all(torch.equal(a, b) for a, b in zip(outs[0][0], outs[1][0]))
    assert _same(outs[0][1], outs[1][1])


def test_graphs_of_shapes_on_either_side_of_a_kernel_routing_threshold(monkeypatch):
    """Which GEMM kernel a Linear takes depends on the batch's row count (ops._dma_shape_ok: the LDS-DMA kernel when its tiles fill
    the chip), and the two kernels read different weight images.  A shape met for the first time AFTER the eager warm-up is
    captured at once, so the image its routing needs must already exist: the eager steps make both images of every weight whose
    dimensions are eligible (ops._both_images).  Thresholds lowered so that base-model shapes of a few hundred rows straddle
    them; a graph step on either side is bit for bit the eager step."""
    from transformertts_amd import _lib, ops
    from transformertts_amd.step import TrainStep
    from transformertts_amd.workload import synth_batch
    monkeypatch.setattr(ops, "IMAGE_MIN_ROWS", 700)        # gated data gradients: DMA kernel from 700 rows on
    monkeypatch.setattr(ops, "DMA_MIN_TILES", 8)           # 256 -> 256 projections: from 8 tiles (897 rows) on
    shapes = [(3, 20, 320), (3, 20, 200), (3, 20, 260), (3, 20, 320), (3, 20, 200)]    # 960, 600, 780 rows
    lib = _lib.load()
    runs = []
    for graph in (False, True):
        cfg, lm, opt, sch = _setup("base", 5, 0)
        batches = [{k: v.to("cuda") for k, v in synth_batch(B, Tp, Tm, cfg["n_mels"], cfg["n_phon"], ragged=False, seed=40 + i).items()}
                   for i, (B, Tp, Tm) in enumerate(shapes)]
        calls = {"h3d": 0, "h3": 0}
        f_d, f_3 = lib.ttts_linear_fwd_h3d, lib.ttts_linear_fwd_h3
        monkeypatch.setattr(lib, "ttts_linear_fwd_h3d", lambda *a: (calls.__setitem__("h3d", calls["h3d"] + 1), f_d(*a))[1])
        monkeypatch.setattr(lib, "ttts_linear_fwd_h3", lambda *a: (calls.__setitem__("h3", calls["h3"] + 1), f_3(*a))[1])
        ts = TrainStep(lm, opt, sch, graph=graph, seed=13)
        losses = [ts(batches[i % len(batches)]).detach().clone() for i in range(8)]
        torch.cuda.synchronize()
        monkeypatch.setattr(lib, "ttts_linear_fwd_h3d", f_d)
        monkeypatch.setattr(lib, "ttts_linear_fwd_h3", f_3)
        assert calls["h3d"] > 0 and calls["h3"] > 0, calls           # both kernels were in use
        if graph:
            assert ts.n_graphs == 3 and ts.capture_fallback is None
        runs.append((losses, _state(lm, opt)))
    assert all(torch.equal(a, b) for a, b in zip(runs[0][0], runs[1][0]))
    assert _same(runs[0][1], runs[1][1])
