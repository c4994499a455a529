"""Both arithmetic forms of the MFMA kernels (fp32 MFMA and split-precision bf16x6), called directly through the C ABI
and held to the same fp64 reference; plus the size-independent properties at the BASELINE batch size."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = 5e-6


def _dev():
    return torch.device("cuda:0")


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(_dev())


def _rel(a, b):
    from conftest import rel_l2
    return rel_l2(a, b)


@pytest.mark.parametrize("M,N,K", [(300, 256, 256), (1000, 1024, 256), (777, 256, 1024), (129, 80, 256)])
def test_linear_forms_agree_with_fp64(M, N, K):
    from transformertts_amd import _lib, ops
    from transformertts_amd.ops import _p, _stream
    lib = _lib.load()
    x, w, b = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=K ** -0.5), _rand(N, seed=3)
    dy = _rand(M, N, seed=4)
    ref = x.double() @ w.double().t() + b.double()
    dw_ref, dx_ref, db_ref = dy.double().t() @ x.double(), dy.double() @ w.double(), dy.double().sum(0)
    for x6 in (False, True):
        y = torch.empty(M, N, device=_dev())
        if x6:
            pl = ops._planes(w, 0, N, K)
            rc = lib.ttts_linear_fwd_x6(_p(x), _p(pl), _p(b), None, _p(y), M, N, K, 0, 0.0, 0, None, 0, 0, _stream())
        else:
            rc = lib.ttts_linear_fwd(_p(x), _p(w), _p(b), None, _p(y), M, N, K, 0, 0.0, 0, None, 0, 0, _stream())
        assert rc == 0 and _rel(y, ref) < TOL
        dx = torch.empty(M, K, device=_dev())
        if x6:
            plt = ops._planes(w, 1, K, N)
            rc = lib.ttts_linear_bwd_data_x6(_p(dy), _p(plt), None, _p(dx), M, N, K, None, 1.0, _stream())
        else:
            rc = lib.ttts_linear_bwd_data(_p(dy), _p(w), None, _p(dx), M, N, K, None, 1.0, _stream())
        assert rc == 0 and _rel(dx, dx_ref) < TOL
        # fused relu / dropout backward mask + residual in the data-gradient epilogue
        hfwd = torch.relu(_rand(M, K, seed=5)) * (torch.rand(M, K, generator=torch.Generator().manual_seed(6)) > 0.3).to(_dev())
        res = _rand(M, K, seed=7)
        if x6:
            rc = lib.ttts_linear_bwd_data_x6(_p(dy), _p(plt), _p(res), _p(dx), M, N, K, _p(hfwd), 1.25, _stream())
        else:
            rc = lib.ttts_linear_bwd_data(_p(dy), _p(w), _p(res), _p(dx), M, N, K, _p(hfwd), 1.25, _stream())
        assert rc == 0 and _rel(dx, dx_ref * (hfwd > 0).double() * 1.25 + res.double()) < TOL
        dw, db = torch.empty(N, K, device=_dev()), torch.empty(N, device=_dev())
        ws = torch.empty(lib.ttts_wgrad_workspace_bytes(M, N, K, 1) // 4, device=_dev())
        f = lib.ttts_linear_bwd_weight_x6 if x6 else lib.ttts_linear_bwd_weight
        rc = f(_p(dy), _p(x), _p(dw), _p(db), _p(ws), ws.numel() * 4, M, N, K, 0, 0, 0, None, _stream())
        assert rc == 0 and _rel(dw, dw_ref) < TOL and _rel(db, db_ref) < TOL
        # accumulate = 1 adds to what is there
        rc = f(_p(dy), _p(x), _p(dw), _p(db), _p(ws), ws.numel() * 4, M, N, K, 0, 0, 1, None, _stream())
        assert rc == 0 and _rel(dw, 2 * dw_ref) < TOL and _rel(db, 2 * db_ref) < TOL


def test_fp16x3_gemms_on_random_shapes():
    """Seeded random shapes through every fp16x3 forward-type entry point (nn.Linear forward / data gradient, Conv1d forward
    / data gradient): row counts with ragged last tiles, widths and depths that are multiples of 4 but not of 32 (padded
    weight images), one to seven taps, utterance lengths down to 1 -- each against fp64, whichever tile the dispatch picks."""
    import random
    from transformertts_amd import _lib, ops
    from transformertts_amd.ops import _p, _stream
    lib = _lib.load()
    rng = random.Random(20260904)
    for case in range(24):
        M = rng.choice([1, 7, 33, 255, 257, 1000, 4099, 20011])
        N = 4 * rng.randint(1, 130)
        K = 4 * rng.randint(1, 130)
        x, w, b = _rand(M, K, seed=case), _rand(N, K, seed=100 + case, scale=K ** -0.5), _rand(N, seed=200 + case)
        y = torch.full((M, N), float("nan"), device=_dev())
        assert lib.ttts_linear_fwd_h3(_p(x), _p(ops._planes(w, 4, N, K)), _p(b), None, _p(y), M, N, K, 0, 0.0, 0, None, 0, 0,
                                      _p(ops._amax(x)), None, _stream()) == 0, (M, N, K)
        assert _rel(y, x.double() @ w.double().t() + b.double()) < TOL, ("linear fwd", M, N, K)
        dy = _rand(M, N, seed=300 + case) * 1e-4
        dx = torch.full((M, K), float("nan"), device=_dev())
        assert lib.ttts_linear_bwd_data_h3(_p(dy), _p(ops._planes(w, 5, K, N)), None, _p(dx), M, N, K, None, 1.0, _p(ops._amax(dy)),
                                           None, _stream()) == 0, (M, N, K)
        assert _rel(dx, dy.double() @ w.double()) < TOL, ("linear dgrad", M, N, K)
    for case in range(16):
        B, T = rng.choice([(1, 1), (2, 5), (3, 64), (5, 131), (2, 870)])
        cin, cout, taps = 4 * rng.randint(1, 70), 4 * rng.randint(1, 70), rng.choice([1, 3, 5, 7])
        x, w, b = _rand(B, T, cin, seed=case), _rand(cout, cin, taps, seed=400 + case, scale=(taps * cin) ** -0.5), _rand(cout, seed=500 + case)
        ref = torch.nn.functional.conv1d(x.double().transpose(1, 2), w.double(), b.double(), padding=taps // 2).transpose(1, 2)
        y = torch.full((B, T, cout), float("nan"), device=_dev())
        assert lib.ttts_conv1d_fwd_h3(_p(x), _p(ops._planes(w, 6, cout, taps * cin, cin, taps)), _p(b), _p(y), B, T, cin, cout, taps,
                                      _p(ops._amax(x)), None, _stream()) == 0, (B, T, cin, cout, taps)
        assert _rel(y, ref) < TOL, ("conv fwd", B, T, cin, cout, taps)
        dy = _rand(B, T, cout, seed=600 + case) * 1e-4
        xd = torch.zeros(B, T, cin, dtype=torch.float64, device=_dev(), requires_grad=True)
        torch.nn.functional.conv1d(xd.transpose(1, 2), w.double(), None, padding=taps // 2).transpose(1, 2).backward(dy.double())
        dx = torch.full((B, T, cin), float("nan"), device=_dev())
        assert lib.ttts_conv1d_bwd_data_h3(_p(dy), _p(ops._planes(w, 7, cin, taps * cout, cout, taps)), _p(dx), B, T, cin, cout, taps,
                                           _p(ops._amax(dy)), _stream()) == 0, (B, T, cin, cout, taps)
        assert _rel(dx, xd.grad) < TOL, ("conv dgrad", B, T, cin, cout, taps)


@pytest.mark.parametrize("M,N,K", [(70001, 768, 96), (66000, 768, 256), (41000, 1024, 128)])
def test_fp16x3_one_wave_per_simd_tile_streams_across_tiles(M, N, K):
    """Shapes the dispatch gives to gemm_h3_wide_kernel (256 x 256 tile, 4 waves; csrc/gemm_h3.hip) with MORE tiles than
    workgroups, so every workgroup runs several tiles back to back -- the operand stream continues across the tile boundary,
    the last tile of a workgroup loads through empty descriptors -- with a ragged last row block, the minimum of three
    k-tiles, and the epilogues of the step (bias + relu + maxima, residual, relu gate of a data gradient): against fp64 on
    sampled rows and bit for bit against a second run (no dependence on what the staging buffers held before)."""
    from transformertts_amd import _lib, ops
    from transformertts_amd.ops import _p, _stream
    lib = _lib.load()
    assert lib.ttts_gemm_tile_choice(M, N, K, 2) == 6               # H3_TILE_256
    x, w, b = _rand(M, K, seed=11), _rand(N, K, seed=12, scale=K ** -0.5), _rand(N, seed=13)
    res = _rand(M, N, seed=14)
    pl = ops._planes(w, 4, N, K)
    xa = ops._amax(x)
    rows = torch.cat([torch.arange(0, 300, device=_dev()), torch.randint(0, M, (700,), device=_dev()),
                      torch.arange(M - 300, M, device=_dev())])
    ref = x[rows].double() @ w.double().t() + b.double()
    y = torch.full((M, N), float("nan"), device=_dev())
    slots = torch.zeros(ops.AMAX_SLOTS, device=_dev())
    assert lib.ttts_linear_fwd_h3(_p(x), _p(pl), _p(b), None, _p(y), M, N, K, 1, 0.0, 0, None, 0, 0, _p(xa), _p(slots), _stream()) == 0
    assert torch.isfinite(y).all() and _rel(y[rows], torch.relu(ref)) < TOL
    assert slots.max().item() == y.abs().max().item()
    y2 = torch.full((M, N), 3.0, device=_dev())
    torch.randn(1 << 22, device=_dev())                              # something else through the caches in between
    assert lib.ttts_linear_fwd_h3(_p(x), _p(pl), _p(b), None, _p(y2), M, N, K, 1, 0.0, 0, None, 0, 0, _p(xa), None, _stream()) == 0
    assert torch.equal(y, y2)
    assert lib.ttts_linear_fwd_h3(_p(x), _p(pl), _p(b), _p(res), _p(y), M, N, K, 0, 0.0, 0, None, 0, 0, _p(xa), None, _stream()) == 0
    assert _rel(y[rows], ref + res[rows].double()) < TOL
    # dropout: the counter-based mask of the bf16x6 kernel, element for element
    y6 = torch.empty(M, N, device=_dev())
    assert lib.ttts_linear_fwd_h3(_p(x), _p(pl), _p(b), None, _p(y), M, N, K, 0, 0.3, 4321, None, 0, 0, _p(xa), None, _stream()) == 0
    assert lib.ttts_linear_fwd_x6(_p(x), _p(ops._planes(w, 0, N, K)), _p(b), None, _p(y6), M, N, K, 0, 0.3, 4321, None, 0, 0, _stream()) == 0
    differ = (y == 0) != (y6 == 0)              # the same mask; a sum that cancels to exactly 0 in one form only is no mask bit
    assert int(differ.sum()) <= 8 and float(torch.maximum(y.abs(), y6.abs())[differ].max() if differ.any() else 0.0) < 1e-5
    assert abs(float((y == 0).float().mean()) - 0.3) < 1e-3 and _rel(y[rows], y6[rows]) < TOL
    del y6, differ
    # data gradient of the same layer: dy (M x N) . W -> dx (M x K) is a K-wide output; the gated form through W^T instead:
    # dh (M x K2) = dy2 (M x K) . Wt with the relu gate of h -- an N-wide output with the gate operand read in the epilogue
    dy2 = _rand(M, K, seed=15) * 1e-5
    hgate = torch.relu(_rand(M, N, seed=16))
    plt = ops._planes(w.t().contiguous(), 5, N, K)                   # weight (K, N) of a Linear N -> K, data-gradient planes
    dh = torch.empty(M, N, device=_dev())
    dslots = torch.zeros(ops.AMAX_SLOTS, device=_dev())
    assert lib.ttts_linear_bwd_data_h3(_p(dy2), _p(plt), None, _p(dh), M, K, N, _p(hgate), 1.25, _p(ops._amax(dy2)), _p(dslots),
                                       _stream()) == 0
    dref = (dy2[rows].double() @ w.double().t()) * (hgate[rows] > 0).double() * 1.25
    assert _rel(dh[rows], dref) < TOL and dslots.max().item() == dh.abs().max().item()


@pytest.mark.parametrize("M,N,K", [(300, 256, 256), (1000, 1024, 256), (777, 256, 1024), (129, 80, 256), (64, 96, 32), (530, 256, 80),
                                   (70, 64, 48)])
def test_fp16x3_forward_form_agrees_with_fp64(M, N, K):
    """The fp16x3 forward kernel (three f16 MFMA terms, both operands pre-scaled from their measured maxima,
    csrc/gemm_h3.hip) against fp64: same gate as the bf16x6 and fp32-MFMA forms, with every epilogue (bias, relu, residual,
    dropout mask identical to the bf16x6 kernel's, go-frame row shift) -- and for operands of ANY magnitude: activations and
    weights from 1e-6 to 1e6, an outlier 5000 x the rest (which the fixed pre-scales of round 2 turned into inf)."""
    from transformertts_amd import _lib, ops
    from transformertts_amd.ops import _p, _stream
    lib = _lib.load()

    def fwd(x, pl, b, res, y, act=0, p=0.0, seed=0, shift=0, T=0, y_am=None):
        return lib.ttts_linear_fwd_h3(_p(x), _p(pl), _p(b), _p(res), _p(y), M, N, K, act, p, seed, None, shift, T,
                                      _p(ops._amax(x)), _p(y_am), _stream())
    b0 = _rand(N, seed=3)
    for a_scale, w_scale in ((1.0, 1.0), (30.0, 0.2), (0.02, 5.0), (1e-6, 1.0), (1e6, 1.0), (1.0, 1e-6), (1.0, 1e6),
                             (1e6, 1e-6), (1e-6, 1e6), (1e-6, 1e-6), (1e6, 1e6), (3e-20, 7e12)):
        x, w = _rand(M, K, seed=1) * a_scale, _rand(N, K, seed=2, scale=K ** -0.5) * w_scale
        b = b0 * (a_scale * w_scale)                                   # a bias of the products' magnitude
        ref = x.double() @ w.double().t() + b.double()
        y = torch.empty(M, N, device=_dev())
        pl = ops._planes(w, 4, N, K)
        assert fwd(x, pl, b, None, y) == 0
        assert _rel(y, ref) < TOL, (a_scale, w_scale, _rel(y, ref))
    b = b0
    x, w = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=K ** -0.5)
    pl, pl6 = ops._planes(w, 4, N, K), ops._planes(w, 0, N, K)
    res = _rand(M, N, seed=8)
    ref = x.double() @ w.double().t() + b.double()
    y, y6 = torch.empty(M, N, device=_dev()), torch.empty(M, N, device=_dev())
    assert fwd(x, pl, b, res, y) == 0
    assert _rel(y, ref + res.double()) < TOL
    slots = torch.zeros(1024, device=_dev())
    assert fwd(x, pl, b, None, y, act=1, y_am=slots) == 0
    assert _rel(y, torch.relu(ref)) < TOL
    assert slots.max().item() == y.abs().max().item()           # the epilogue's own maxima of y (slot-wise atomic max)
    assert fwd(x, pl, b, None, y, p=0.3, seed=1234) == 0
    assert lib.ttts_linear_fwd_x6(_p(x), _p(pl6), _p(b), None, _p(y6), M, N, K, 0, 0.3, 1234, None, 0, 0, _stream()) == 0
    assert torch.equal(y == 0, y6 == 0) and _rel(y, y6) < TOL           # same counter-based mask in both forms
    if M % 10 == 0:                                                     # go-frame shift inside utterances of T rows
        T = M // 10
        assert fwd(x, pl, b, None, y, shift=-1, T=T) == 0
        xs = torch.roll(x.view(10, T, K), 1, dims=1).clone()
        xs[:, 0] = 0
        assert _rel(y, xs.view(M, K).double() @ w.double().t() + b.double()) < TOL
    # an outlier activation: finite, and both its row and the ordinary rows beside it are fp32-grade
    x2 = x.clone()
    x2[3, 5] = 5000.0
    assert fwd(x2, pl, b, None, y) == 0
    ref2 = x2.double() @ w.double().t() + b.double()
    assert torch.isfinite(y).all() and _rel(y[3], ref2[3]) < TOL and _rel(y[4:], ref2[4:]) < TOL and _rel(y, ref2) < TOL
    # an outlier weight (|w| >= 16 overflowed the fixed 2^12 weight scale of round 2)
    w2 = w.clone()
    w2[1, 2] = 300.0
    pl2 = ops._planes(w2, 4, N, K)
    assert fwd(x, pl2, b, None, y) == 0
    ref3 = x.double() @ w2.double().t() + b.double()
    assert torch.isfinite(y).all() and _rel(y, ref3) < TOL and _rel(y[:, 2:], ref3[:, 2:]) < TOL
    # non-finite operands travel on visibly, as in fp32 arithmetic
    x3 = x.clone()
    x3[7, 1] = float("inf")
    assert fwd(x3, pl, b, None, y) == 0
    assert not torch.isfinite(y[7]).any()
    # an all-zero activation
    assert fwd(torch.zeros_like(x), pl, b, None, y) == 0
    assert _rel(y, b.double().expand(M, N)) < TOL


@pytest.mark.parametrize("M,N,K", [(300, 256, 256), (1000, 1024, 256), (777, 256, 1024), (129, 96, 80), (530, 80, 256), (70, 20, 64)])
@pytest.mark.parametrize("mag", [1.0, 3e-7, 1e-12, 1e6])
def test_fp16x3_data_gradient_with_dynamic_scale(M, N, K, mag):
    """dx = dy . w in the fp16x3 form for gradients of ANY magnitude (3e-7 is what a mean-reduced loss produces): the
    pre-scale comes from ttts_amax_partials.  Includes a wide in-tensor range (one row 1e4 x larger than the rest, padded
    rows exactly zero), the fused relu-gate / residual epilogue, and an all-zero dy."""
    from transformertts_amd import _lib, ops
    from transformertts_amd.ops import _p, _stream
    lib = _lib.load()
    w = _rand(N, K, seed=2, scale=K ** -0.5)
    dy = _rand(M, N, seed=4) * mag
    dy[5] *= 1e4
    dy[M - 7:] = 0.0
    plt = ops._planes(w, 5, K, N)
    dx = torch.empty(M, K, device=_dev())
    am = ops._amax(dy)
    assert abs(am.max().item() - dy.abs().max().item()) == 0.0
    slots = torch.zeros(1024, device=_dev())
    assert lib.ttts_linear_bwd_data_h3(_p(dy), _p(plt), None, _p(dx), M, N, K, None, 1.0, _p(am), _p(slots), _stream()) == 0
    assert slots.max().item() == dx.abs().max().item()          # the epilogue's own maxima of dx (slot-wise atomic max)
    ref = dy.double() @ w.double()
    assert _rel(dx, ref) < TOL, _rel(dx, ref)
    small = torch.ones(M, dtype=torch.bool); small[5] = False                  # the small rows on their own, not drowned by row 5
    assert _rel(dx[small.to(_dev())], ref[small.to(_dev())]) < TOL
    hfwd = torch.relu(_rand(M, K, seed=5))
    res = _rand(M, K, seed=7) * mag
    assert lib.ttts_linear_bwd_data_h3(_p(dy), _p(plt), _p(res), _p(dx), M, N, K, _p(hfwd), 1.25, _p(am), None, _stream()) == 0
    assert _rel(dx, ref * (hfwd > 0).double() * 1.25 + res.double()) < TOL
    zero = torch.zeros_like(dy)
    assert lib.ttts_linear_bwd_data_h3(_p(zero), _p(plt), None, _p(dx), M, N, K, None, 1.0, _p(ops._amax(zero)), None, _stream()) == 0
    assert float(dx.abs().max()) == 0.0


@pytest.mark.parametrize("M,N,K", [(300, 256, 256), (1000, 1024, 256), (777, 256, 1024), (129, 80, 256), (4000, 256, 80),
                                   (25630, 768, 256), (25610, 300, 1024),      # long row ranges: the 8-wave 256-wide tile
                                   (25990, 512, 512), (26001, 256, 768)])      # ... whole 256 x 256 tiles: rows by LDS-DMA (wgrad_dma.hip)
@pytest.mark.parametrize("mag", [1.0, 3e-7])
def test_fp16x3_weight_gradient(M, N, K, mag):
    """dW = dy^T x and db = column sums of dy in the fp16x3 form (dynamic pre-scale of dy, 32-row k-steps), stored and
    accumulated, with the go-frame row shift of the decoder pre-net."""
    from transformertts_amd import _lib, ops
    from transformertts_amd.ops import _p, _stream
    lib = _lib.load()
    x, dy = _rand(M, K, seed=1), _rand(M, N, seed=4) * mag
    dy[M // 2] *= 300.0
    am = ops._amax(dy)
    dw, db = torch.empty(N, K, device=_dev()), torch.empty(N, device=_dev())
    ws = torch.empty(lib.ttts_wgrad_workspace_bytes(M, N, K, 1) // 4, device=_dev())
    f = lib.ttts_linear_bwd_weight_h3
    xm = ops._amax(x)
    # LDS is not cleared between kernels: leave NaN operand rows in every CU's LDS first (the same launch on NaN inputs), so that a
    # step the kernel never requested but still converts -- the step past a short last row split, (25630, 768, 256) has one --
    # would show (round 6: `stale * 0` = NaN reached one bias gradient; csrc/wgrad_dma.hip, fetch)
    nan_dy, nan_x = torch.full_like(dy, float("nan")), torch.full_like(x, float("nan"))
    assert f(_p(nan_dy), _p(nan_x), _p(dw), _p(db), _p(ws), ws.numel() * 4, M, N, K, 0, 0, 0, _p(am), _p(xm), None, _stream()) == 0
    assert f(_p(dy), _p(x), _p(dw), _p(db), _p(ws), ws.numel() * 4, M, N, K, 0, 0, 0, _p(am), _p(xm), None, _stream()) == 0
    dw_ref, db_ref = dy.double().t() @ x.double(), dy.double().sum(0)
    assert _rel(dw, dw_ref) < TOL and _rel(db, db_ref) < TOL, (_rel(dw, dw_ref), _rel(db, db_ref))
    assert f(_p(dy), _p(x), _p(dw), _p(db), _p(ws), ws.numel() * 4, M, N, K, 0, 0, 1, _p(am), _p(xm), None, _stream()) == 0
    assert _rel(dw, 2 * dw_ref) < TOL and _rel(db, 2 * db_ref) < TOL
    # activations of any magnitude (the x operand had a fixed 2^4 pre-scale in round 2)
    for xs_ in (1e-6, 1e6):
        x2 = x * xs_
        x2[M // 3, 1] *= 3000.0
        assert f(_p(dy), _p(x2), _p(dw), _p(db), _p(ws), ws.numel() * 4, M, N, K, 0, 0, 0, _p(am), _p(ops._amax(x2)), None,
                 _stream()) == 0
        assert torch.isfinite(dw).all() and _rel(dw, dy.double().t() @ x2.double()) < TOL
    if M % 10 == 0:
        T = M // 10
        assert f(_p(dy), _p(x), _p(dw), None, _p(ws), ws.numel() * 4, M, N, K, -1, T, 0, _p(am), _p(xm), None, _stream()) == 0
        xs = torch.roll(x.view(10, T, K), 1, dims=1).clone()
        xs[:, 0] = 0
        assert _rel(dw, dy.double().t() @ xs.view(M, K).double()) < TOL


@pytest.mark.parametrize("M,shapes", [(13920, [(256, 256), (256, 256), (256, 256), (256, 256)]),       # a decoder layer's small linears at B = 16
                                      (6401, [(768, 256), (256, 256), (1024, 256), (256, 1024)]),        # an encoder layer (ragged last k-step)
                                      (300, [(256, 256), (128, 384)]), (55680, [(256, 256)]),
                                      (25630, [(256, 1024), (1024, 256), (768, 256)]),                   # a decoder layer's big weights: the LDS-DMA tile
                                      (27001, [(512, 512)])])
def test_grouped_weight_gradients(M, shapes):
    """ttts_wgrad_group: up to four independent dW_i = dy_i^T x_i (+ bias column sums) as ONE launch of the
    128 x 128 fp16x3 tile (class 1) or of the 256 x 256 LDS-DMA tile (class 2) with the row splits planned for the group -- every member against fp64, stored into zeroed sinks and
    accumulated on a second call (the sinks' contract), same bits on a third run into fresh sinks (fixed-order reductions);
    members of very different magnitude (each has its own dynamic pre-scales)."""
    import ctypes
    from transformertts_amd import _lib, ops
    from transformertts_amd.ops import _stream
    lib, dev = _lib.load(), _dev()
    n = len(shapes)
    xs = [_rand(M, K, seed=10 + i) * (1e-3 if i == 1 else 1.0) for i, (N, K) in enumerate(shapes)]
    dys = [_rand(M, N, seed=20 + i) * (3e-7 if i == 0 else 1.0 if i == 2 else 40.0) for i, (N, K) in enumerate(shapes)]
    cls = lib.ttts_wgrad_group_ok(M, *shapes[0], 1)
    assert cls == (2 if M > 25600 and shapes[0] != (256, 256) else 1)
    for (N, K) in shapes:
        assert lib.ttts_wgrad_group_ok(M, N, K, 1) == cls                       # the members of a launch share a class
    assert lib.ttts_wgrad_group_ok(55680, 1024, 256, 1) == 2 and lib.ttts_wgrad_group_ok(55680, 256, 256, 1) == 1
    assert lib.ttts_wgrad_group_ok(55680, 256, 80, 1) == 3 and lib.ttts_wgrad_group_ok(55680, 80, 256, 5) == 4      # the mel side's tiles
    assert lib.ttts_wgrad_group_ok(55680, 300, 1024, 1) == 0 and lib.ttts_wgrad_group_ok(55680, 32, 32, 1) == 0     # launches of their own
    ams, xms = [ops._amax(t) for t in dys], [ops._amax(t) for t in xs]
    wss = [torch.empty(lib.ttts_wgrad_workspace_bytes(M, N, K, 1) // 4, device=dev) for (N, K) in shapes]
    PA, ZA, LA, IA = ctypes.c_void_p * n, ctypes.c_size_t * n, ctypes.c_int64 * n, ctypes.c_int * n
    ptr = lambda ts: PA(*[(t.data_ptr() if t is not None else None) for t in ts])      # noqa: E731

    def run(dws, dbs):
        rc = lib.ttts_wgrad_group(n, ptr(dys), ptr(xs), ptr(dws), ptr(dbs), ptr(wss), ZA(*[w.numel() * 4 for w in wss]),
                                  LA(*[M] * n), IA(*[N for N, K in shapes]), IA(*[K for N, K in shapes]), IA(*[1] * n), IA(*[0] * n),
                                  IA(*[0] * n), 1, ptr(ams), ptr(xms), None, _stream())
        assert rc == 0, _lib.last_error()
    dws = [torch.zeros(N, K, device=dev) for (N, K) in shapes]
    dbs = [torch.zeros(N, device=dev) if i != 1 else None for i, (N, K) in enumerate(shapes)]      # (a member without a bias)
    run(dws, dbs)
    refs = [(dy.double().t() @ x.double(), dy.double().sum(0)) for dy, x in zip(dys, xs)]
    for i, (dw, db) in enumerate(zip(dws, dbs)):
        assert _rel(dw, refs[i][0]) < TOL, (i, _rel(dw, refs[i][0]))
        assert db is None or _rel(db, refs[i][1]) < TOL
    first = [dw.clone() for dw in dws]
    run(dws, dbs)                                           # accumulate: the sinks now hold twice the gradient
    for i, (dw, db) in enumerate(zip(dws, dbs)):
        assert _rel(dw, 2 * refs[i][0]) < TOL and (db is None or _rel(db, 2 * refs[i][1]) < TOL)
    dws2 = [torch.zeros(N, K, device=dev) for (N, K) in shapes]
    torch.randn(1 << 23, device=dev)
    run(dws2, [torch.zeros(N, device=dev) if i != 1 else None for i, (N, K) in enumerate(shapes)])
    assert all(torch.equal(a, b) for a, b in zip(first, dws2))


@pytest.mark.parametrize("B,T,chans", [(64, 100, [(256, 256), (256, 256), (256, 256)]),       # the encoder pre-net's convolutions
                                       (30, 870, [(256, 256), (256, 256), (256, 256)]),        # the post-net's: the LDS-DMA tile, 5 taps each
                                       (7, 33, [(256, 128), (128, 256)])])
def test_grouped_conv_weight_gradients(B, T, chans):
    """ttts_wgrad_group with convolution members (five taps, same padding, utterance clipping): one grid for the weight gradients of
    a stack of convolutions against torch's conv1d autograd in fp64; with a linear member beside them in the small class."""
    import ctypes
    from transformertts_amd import _lib, ops
    from transformertts_amd.ops import _stream
    lib, dev = _lib.load(), _dev()
    M, taps = B * T, 5
    members = [(cout, cin, taps) for cout, cin in chans]
    cls = lib.ttts_wgrad_group_ok(M, chans[0][0], chans[0][1], taps)
    assert cls in (1, 2)
    if cls == 1:
        members.append((256, 256, 1))                    # a linear weight in the same launch
    n = len(members)
    xs = [_rand(B, T, cin, seed=30 + i) for i, (cout, cin, tp) in enumerate(members)]
    dys = [_rand(B, T, cout, seed=40 + i) * (2e-7 if i == 0 else 1.0) for i, (cout, cin, tp) in enumerate(members)]
    refs = []
    for (cout, cin, tp), x, dy in zip(members, xs, dys):
        if tp == 1:
            refs.append((dy.double().view(M, cout).t() @ x.double().view(M, cin), dy.double().view(M, cout).sum(0)))
            continue
        wd = torch.zeros(cout, cin, tp, dtype=torch.float64, device=dev, requires_grad=True)
        bd = torch.zeros(cout, dtype=torch.float64, device=dev, requires_grad=True)
        torch.nn.functional.conv1d(x.double().transpose(1, 2), wd, bd, padding=tp // 2).transpose(1, 2).backward(dy.double())
        refs.append((wd.grad, bd.grad))
    dws = [torch.zeros(cout, cin, tp, device=dev) if tp > 1 else torch.zeros(cout, cin, device=dev) for cout, cin, tp in members]
    dbs = [torch.zeros(cout, device=dev) for cout, cin, tp in members]
    wss = [torch.empty(lib.ttts_wgrad_workspace_bytes(M, cout, cin, tp) // 4, device=dev) for cout, cin, tp in members]
    ams, xms = [ops._amax(t) for t in dys], [ops._amax(t) for t in xs]
    PA, ZA, LA, IA = ctypes.c_void_p * n, ctypes.c_size_t * n, ctypes.c_int64 * n, ctypes.c_int * n
    ptr = lambda ts: PA(*[t.data_ptr() for t in ts])      # noqa: E731
    rc = lib.ttts_wgrad_group(n, ptr(dys), ptr(xs), ptr(dws), ptr(dbs), ptr(wss), ZA(*[w.numel() * 4 for w in wss]), LA(*[M] * n),
                              IA(*[m[0] for m in members]), IA(*[m[1] for m in members]), IA(*[m[2] for m in members]),
                              IA(*[T if m[2] > 1 else 0 for m in members]), IA(*[0] * n), 1, ptr(ams), ptr(xms), None, _stream())
    assert rc == 0, _lib.last_error()
    for i in range(n):
        assert _rel(dws[i], refs[i][0]) < TOL and _rel(dbs[i], refs[i][1]) < TOL, (i, _rel(dws[i], refs[i][0]), _rel(dbs[i], refs[i][1]))


@pytest.mark.parametrize("cls,N,K", [(3, 256, 80), (4, 80, 256)])
def test_grouped_weight_gradients_of_the_mel_side(cls, N, K):
    """classes 3 / 4 of ttts_wgrad_group (the 96-wide tiles of weights with 80 input / output channels): a five-tap convolution and
    a linear weight in one grid -- the linear one with the decoder pre-net's go-frame row shift (class 3) -- against fp64."""
    import ctypes
    from transformertts_amd import _lib, ops
    from transformertts_amd.ops import _stream
    lib, dev = _lib.load(), _dev()
    B, T = 16, 870
    M = B * T
    assert lib.ttts_wgrad_group_ok(M, N, K, 5) == cls and lib.ttts_wgrad_group_ok(M, N, K, 1) == cls
    shift = -1 if cls == 3 else 0
    xs = [_rand(B, T, K, seed=50), _rand(B, T, K, seed=51)]
    dys = [_rand(B, T, N, seed=52) * 3e-6, _rand(B, T, N, seed=53)]
    wd = torch.zeros(N, K, 5, dtype=torch.float64, device=dev, requires_grad=True)
    bd = torch.zeros(N, dtype=torch.float64, device=dev, requires_grad=True)
    torch.nn.functional.conv1d(xs[0].double().transpose(1, 2), wd, bd, padding=2).transpose(1, 2).backward(dys[0].double())
    xl = xs[1].double()
    if shift:
        xl = torch.roll(xl, 1, dims=1).clone()
        xl[:, 0] = 0
    refs = [(wd.grad, bd.grad), (dys[1].double().view(M, N).t() @ xl.view(M, K), dys[1].double().view(M, N).sum(0))]
    taps = [5, 1]
    dws = [torch.zeros(N, K, 5, device=dev), torch.zeros(N, K, device=dev)]
    dbs = [torch.zeros(N, device=dev), torch.zeros(N, device=dev)]
    wss = [torch.empty(lib.ttts_wgrad_workspace_bytes(M, N, K, tp) // 4, device=dev) for tp in taps]
    PA, ZA, LA, IA = ctypes.c_void_p * 2, ctypes.c_size_t * 2, ctypes.c_int64 * 2, ctypes.c_int * 2
    ptr = lambda ts: PA(*[t.data_ptr() for t in ts])      # noqa: E731
    rc = lib.ttts_wgrad_group(2, ptr(dys), ptr(xs), ptr(dws), ptr(dbs), ptr(wss), ZA(*[w.numel() * 4 for w in wss]), LA(M, M), IA(N, N),
                              IA(K, K), IA(*taps), IA(T, T if shift else 0), IA(0, shift), 1, ptr([ops._amax(t) for t in dys]),
                              ptr([ops._amax(t) for t in xs]), None, _stream())
    assert rc == 0, _lib.last_error()
    for i in range(2):
        assert _rel(dws[i], refs[i][0]) < TOL and _rel(dbs[i], refs[i][1]) < TOL, (i, _rel(dws[i], refs[i][0]), _rel(dbs[i], refs[i][1]))


def test_weight_gradient_into_several_destinations():
    """ttts_linear_bwd_weight_h3_parts: ONE weight-gradient GEMM whose row blocks belong to different weights (the stacked K/V
    projection of all decoder layers): block i accumulated into destination i, bias column sums likewise."""
    import ctypes
    from transformertts_amd import _lib, ops
    from transformertts_amd.ops import _p, _stream
    lib, dev = _lib.load(), _dev()
    M, L, Np, K = 6400, 3, 512, 256
    x, dy = _rand(M, K, seed=1), _rand(M, L * Np, seed=2) * 1e-3
    dws = [torch.full((Np, K), 0.5, device=dev) for _ in range(L)]
    dbs = [torch.full((Np,), -0.25, device=dev) for _ in range(L)]
    ws = torch.empty(lib.ttts_wgrad_workspace_bytes(M, L * Np, K, 1) // 4, device=dev)
    PA = ctypes.c_void_p * L
    rc = lib.ttts_linear_bwd_weight_h3_parts(_p(dy), _p(x), PA(*[t.data_ptr() for t in dws]), PA(*[t.data_ptr() for t in dbs]), L, _p(ws),
                                             ws.numel() * 4, M, L * Np, K, 1, _p(ops._amax(dy)), _p(ops._amax(x)), None, _stream())
    assert rc == 0, _lib.last_error()
    ref = dy.double().t() @ x.double()
    for i in range(L):
        assert _rel(dws[i].double() - 0.5, ref[i * Np:(i + 1) * Np]) < TOL
        assert _rel(dbs[i].double() + 0.25, dy.double()[:, i * Np:(i + 1) * Np].sum(0)) < 1e-5


@pytest.mark.parametrize("B,T,cin,cout", [(3, 50, 128, 256), (2, 7, 256, 128), (5, 1, 128, 128), (6, 45, 80, 256), (6, 45, 256, 80),
                                          (30, 870, 256, 256),                 # five taps x one 256-wide tile (rows by LDS-DMA)
                                          (31, 833, 512, 256), (1600, 16, 256, 256)])   # ... utterances of exactly one step
def test_fp16x3_conv_weight_gradient(B, T, cin, cout):
    from transformertts_amd import _lib, ops
    from transformertts_amd.ops import _p, _stream
    lib = _lib.load()
    x, dy = _rand(B, T, cin, seed=1), _rand(B, T, cout, seed=2) * 2e-7
    wd = torch.zeros(cout, cin, 5, dtype=torch.float64, device=_dev(), requires_grad=True)
    bd = torch.zeros(cout, dtype=torch.float64, device=_dev(), requires_grad=True)
    y = torch.nn.functional.conv1d(x.double().transpose(1, 2), wd, bd, padding=2).transpose(1, 2)
    y.backward(dy.double())
    dw, db = torch.empty(cout, cin, 5, device=_dev()), torch.empty(cout, device=_dev())
    ws = torch.empty(lib.ttts_wgrad_workspace_bytes(B * T, cout, cin, 5) // 4, device=_dev())
    assert lib.ttts_conv1d_bwd_weight_h3(_p(dy), _p(x), _p(dw), _p(db), _p(ws), ws.numel() * 4, B, T, cin, cout, 5, 0,
                                         _p(ops._amax(dy)), _p(ops._amax(x)), None, _stream()) == 0
    assert _rel(dw, wd.grad) < TOL and _rel(db, bd.grad) < TOL, (_rel(dw, wd.grad), _rel(db, bd.grad))


@pytest.mark.parametrize("B,T,cin,cout", [(3, 50, 128, 256), (2, 7, 256, 128), (5, 1, 128, 128), (4, 33, 80, 64), (6, 41, 512, 80),
                                          (3, 870, 256, 80), (2, 9, 64, 20),
                                          (48, 870, 256, 256)])         # the 224-row one-wave-per-SIMD tile, clipping loader
def test_fp16x3_conv_data_gradient(B, T, cin, cout):
    from transformertts_amd import _lib, ops
    from transformertts_amd.ops import _p, _stream
    lib = _lib.load()
    w = _rand(cout, cin, 5, seed=2, scale=(5 * cin) ** -0.5)
    dy = _rand(B, T, cout, seed=3) * 2e-7
    xd = torch.zeros(B, T, cin, dtype=torch.float64, device=_dev(), requires_grad=True)
    y = torch.nn.functional.conv1d(xd.transpose(1, 2), w.double(), None, padding=2).transpose(1, 2)
    y.backward(dy.double())
    dx = torch.empty(B, T, cin, device=_dev())
    pl = ops._planes(w, 7, cin, 5 * cout, cout, 5)
    assert lib.ttts_conv1d_bwd_data_h3(_p(dy), _p(pl), _p(dx), B, T, cin, cout, 5, _p(ops._amax(dy)), _stream()) == 0
    assert _rel(dx, xd.grad) < TOL, _rel(dx, xd.grad)


@pytest.mark.parametrize("B,T,cin,cout", [(3, 50, 128, 256), (2, 7, 256, 128), (5, 1, 128, 128), (4, 33, 64, 80), (6, 41, 80, 512),
                                          (3, 870, 80, 256), (2, 9, 20, 64), (48, 870, 256, 256)])
def test_fp16x3_conv_forward_agrees_with_fp64(B, T, cin, cout):
    from transformertts_amd import _lib, ops
    from transformertts_amd.ops import _p, _stream
    lib = _lib.load()
    x, w, b = _rand(B, T, cin, seed=1), _rand(cout, cin, 5, seed=2, scale=(5 * cin) ** -0.5), _rand(cout, seed=3)
    ref = torch.nn.functional.conv1d(x.double().transpose(1, 2), w.double(), b.double(), padding=2).transpose(1, 2)
    y = torch.empty(B, T, cout, device=_dev())
    pl = ops._planes(w, 6, cout, 5 * cin, cin, 5)
    assert lib.ttts_conv1d_fwd_h3(_p(x), _p(pl), _p(b), _p(y), B, T, cin, cout, 5, _p(ops._amax(x)), None, _stream()) == 0
    assert _rel(y, ref) < TOL, _rel(y, ref)
    x2 = x * 1e5                                                    # any magnitude
    assert lib.ttts_conv1d_fwd_h3(_p(x2), _p(pl), _p(b), _p(y), B, T, cin, cout, 5, _p(ops._amax(x2)), None, _stream()) == 0
    ref2 = torch.nn.functional.conv1d(x2.double().transpose(1, 2), w.double(), b.double(), padding=2).transpose(1, 2)
    assert torch.isfinite(y).all() and _rel(y, ref2) < TOL


@pytest.mark.parametrize("B,T,cin,cout,offset", [(3, 50, 128, 256, 0.0), (7, 333, 256, 256, 0.0), (2, 7, 256, 128, 40.0),
                                                 (30, 870, 256, 256, 3.0), (5, 97, 64, 64, 0.0),
                                                 (48, 870, 256, 256, 3.0),      # 41 760 rows: the 224-row one-wave-per-SIMD tile with the
                                                 (61, 401, 256, 256, 0.0)])     # clipping loader + this epilogue (ragged last tile too)
def test_batchnorm_statistics_from_the_conv_epilogue(B, T, cin, cout, offset):
    """The fp16x3 convolution leaves BatchNorm's row-chunk partials (count, mean, M2 per channel) behind in its epilogue and
    ttts_bn_train_stats_from_partials merges them: same mean / invstd / running statistics as the separate two-pass
    statistics kernel and as fp64, also for ragged row counts (rows past M do not count) and for a channel offset 40 x
    the spread (sums are taken about each slab's first row, so nothing cancels)."""
    from transformertts_amd import _lib, ops
    from transformertts_amd.ops import _p, _stream
    lib = _lib.load()
    M = B * T
    x, w = _rand(B, T, cin, seed=1), _rand(cout, cin, 5, seed=2, scale=(5 * cin) ** -0.5)
    b = _rand(cout, seed=3) + offset
    nblk = lib.ttts_conv1d_fwd_h3_bn_blocks(B, T, cin, cout, 5)
    assert 0 < nblk <= 512
    y = torch.empty(B, T, cout, device=_dev())
    ws = ops._ws(lib.ttts_bn_workspace_bytes(M, cout), _dev())
    pl = ops._planes(w, 6, cout, 5 * cin, cin, 5)
    assert lib.ttts_conv1d_fwd_h3(_p(x), _p(pl), _p(b), _p(y), B, T, cin, cout, 5, _p(ops._amax(x)), _p(ws), _stream()) == 0
    res = []
    for fused in (True, False):
        mean, invstd = torch.empty(cout, device=_dev()), torch.empty(cout, device=_dev())
        rm, rv, nbt = torch.zeros(cout, device=_dev()), torch.ones(cout, device=_dev()), torch.zeros((), dtype=torch.int64, device=_dev())
        if fused:
            assert lib.ttts_bn_train_stats_from_partials(_p(ws), nblk, _p(mean), _p(invstd), _p(rm), _p(rv), _p(nbt), cout, 0.1, 1e-5,
                                                         _stream()) == 0
        else:
            ws2 = ops._ws(lib.ttts_bn_workspace_bytes(M, cout), _dev())
            assert lib.ttts_bn_train_stats(_p(y), _p(mean), _p(invstd), _p(rm), _p(rv), _p(nbt), _p(ws2), ws2.numel() * 4, M, cout,
                                           0.1, 1e-5, _stream()) == 0
        res.append((mean, invstd, rm, rv, int(nbt)))
    yd = y.double().view(M, cout)
    mean64, var64 = yd.mean(0), yd.var(0, unbiased=False)
    inv64 = 1.0 / torch.sqrt(var64 + 1e-5)
    for mean, invstd, rm, rv, n in res:
        assert n == 1
        assert float((mean.double() - mean64).abs().max()) < 2e-6 * (1.0 + abs(offset)) and _rel(invstd, inv64) < 2e-6, \
            (float((mean.double() - mean64).abs().max()), _rel(invstd, inv64))
        assert _rel(rv, 0.9 + 0.1 * yd.var(0, unbiased=True)) < 2e-6
    assert _rel(res[0][1], res[1][1]) < 1e-6 and _rel(res[0][0], res[1][0]) < 1e-6
    # the plain forward (no partials) writes the same y
    y2 = torch.empty_like(y)
    assert lib.ttts_conv1d_fwd_h3(_p(x), _p(pl), _p(b), _p(y2), B, T, cin, cout, 5, _p(ops._amax(x)), None, _stream()) == 0
    assert torch.equal(y, y2)


@pytest.mark.parametrize("B,T,cin,cout,offset", [(6, 50, 128, 256, 0.0), (14, 333, 256, 256, 2.0), (4, 7, 256, 128, 40.0),
                                                 (64, 870, 256, 256, 3.0),      # the 224-row tile: 55 680 = 248 chunks + 128 rows
                                                 (32, 870, 256, 256, 0.0)])     # 27 840 rows: 64-row chunks, the last tile's last chunk empty
def test_batchnorm_statistics_of_each_half_of_a_twin_batch(B, T, cin, cout, offset):
    """ttts_bn_train_stats_from_partials_rows: the statistics of EACH HALF of the rows of one convolution (a twin batch)
    from the epilogue's row-chunk partials -- a half's whole chunks -- plus the rows of the chunk the halves share, read from y
    itself: mean / invstd / running statistics of each half against fp64 over its own rows, the no-grad half (the second) first."""
    from transformertts_amd import _lib, ops
    from transformertts_amd.ops import _p, _off, _stream
    lib = _lib.load()
    M2, M = B * T, B * T // 2
    x, w = _rand(B, T, cin, seed=1), _rand(cout, cin, 5, seed=2, scale=(5 * cin) ** -0.5)
    x[B // 2:] *= 1.7                                               # (the halves differ)
    b = _rand(cout, seed=3) + offset
    nblk = lib.ttts_conv1d_fwd_h3_bn_blocks(B, T, cin, cout, 5)
    chunk = lib.ttts_conv1d_fwd_h3_bn_chunk_rows(B, T, cin, cout, 5)
    assert 0 < nblk <= 512 and 0 < chunk <= 256 and nblk >= -(-M2 // chunk)
    y = torch.empty(B, T, cout, device=_dev())
    ws = ops._ws(lib.ttts_bn_workspace_bytes(M2, cout), _dev())
    pl = ops._planes(w, 6, cout, 5 * cin, cin, 5)
    assert lib.ttts_conv1d_fwd_h3(_p(x), _p(pl), _p(b), _p(y), B, T, cin, cout, 5, _p(ops._amax(x)), _p(ws), _stream()) == 0
    s_, cut = divmod(M, chunk)
    runs = {0: (0, s_, s_ * chunk, cut), 1: (s_ + 1, nblk - s_ - 1, M, min((s_ + 1) * chunk, M2) - M)} if cut else \
           {0: (0, s_, 0, 0), 1: (s_, nblk - s_, 0, 0)}
    rm, rv, nbt = torch.zeros(cout, device=_dev()), torch.ones(cout, device=_dev()), torch.zeros((), dtype=torch.int64, device=_dev())
    yd = y.double().view(M2, cout)
    rm64, rv64 = torch.zeros(cout, dtype=torch.float64, device=_dev()), torch.ones(cout, dtype=torch.float64, device=_dev())
    for h in (1, 0):
        b0, nb, r0, nr = runs[h]
        mean, invstd = torch.empty(cout, device=_dev()), torch.empty(cout, device=_dev())
        assert lib.ttts_bn_train_stats_from_partials_rows(_off(ws, b0 * 3 * cout) if nb else None, nb, _off(y, r0 * cout) if nr else None, nr,
                                                          _p(mean), _p(invstd), _p(rm), _p(rv), _p(nbt), cout, 0.1, 1e-5, _stream()) == 0
        half = yd[h * M:(h + 1) * M]
        mean64, var64 = half.mean(0), half.var(0, unbiased=False)
        rm64 = 0.9 * rm64 + 0.1 * mean64
        rv64 = 0.9 * rv64 + 0.1 * half.var(0, unbiased=True)
        assert float((mean.double() - mean64).abs().max()) < 2e-6 * (1.0 + abs(offset)), (h, float((mean.double() - mean64).abs().max()))
        assert _rel(invstd, 1.0 / torch.sqrt(var64 + 1e-5)) < 2e-6, (h, _rel(invstd, 1.0 / torch.sqrt(var64 + 1e-5)))
    assert int(nbt) == 2 and _rel(rv, rv64) < 2e-6 and float((rm.double() - rm64).abs().max()) < 2e-6 * (1.0 + abs(offset))
    # ttts_bn_train_stats_twin: both halves in ONE launch, the same results bit for bit (running statistics: first set first)
    rm2, rv2, nbt2 = torch.zeros(cout, device=_dev()), torch.ones(cout, device=_dev()), torch.zeros((), dtype=torch.int64, device=_dev())
    mi = torch.empty(2, 2, cout, device=_dev())
    sets = []
    for h in (1, 0):
        b0, nb, r0, nr = runs[h]
        sets += [_off(ws, b0 * 3 * cout) if nb else None, nb, _off(y, r0 * cout) if nr else None, nr, _p(mi[h, 0]), _p(mi[h, 1])]
    assert lib.ttts_bn_train_stats_twin(*sets, _p(rm2), _p(rv2), _p(nbt2), cout, 0.1, 1e-5, _stream()) == 0
    assert int(nbt2) == 2 and torch.equal(rm2, rm) and torch.equal(rv2, rv)
    assert torch.equal(mi[0, 0], mean) and torch.equal(mi[0, 1], invstd)          # (the loop's last half was h = 0)
    # refusals: nothing to merge, more rows than the kernel folds
    assert lib.ttts_bn_train_stats_from_partials_rows(None, 0, None, 0, _p(rm), _p(rv), _p(rm), _p(rv), _p(nbt), cout, 0.1, 1e-5, _stream()) != 0
    assert lib.ttts_bn_train_stats_from_partials_rows(_p(ws), 1, _p(y), 257, _p(rm), _p(rv), _p(rm), _p(rv), _p(nbt), cout, 0.1, 1e-5, _stream()) != 0


@pytest.mark.parametrize("B,T,cin,cout", [(3, 50, 128, 256), (2, 7, 256, 128), (5, 1, 128, 128)])
def test_conv_weight_gradient_forms(B, T, cin, cout):
    """Both weight-gradient kernels on the conv form (shifted rows, utterance clipping, utterances shorter than a k-step)."""
    from transformertts_amd import _lib
    from transformertts_amd.ops import _p, _stream
    lib = _lib.load()
    x, dy = _rand(B, T, cin, seed=1), _rand(B, T, cout, seed=2)
    xd = x.double().transpose(1, 2)
    wd = torch.zeros(cout, cin, 5, dtype=torch.float64, device=_dev(), requires_grad=True)
    torch.nn.functional.conv1d(xd, wd, None, padding=2).backward(dy.double().transpose(1, 2))
    for f in (lib.ttts_conv1d_bwd_weight, lib.ttts_conv1d_bwd_weight_x6):
        dw, db = torch.empty(cout, cin, 5, device=_dev()), torch.empty(cout, device=_dev())
        ws = torch.empty(lib.ttts_wgrad_workspace_bytes(B * T, cout, cin, 5) // 4, device=_dev())
        assert f(_p(dy), _p(x), _p(dw), _p(db), _p(ws), ws.numel() * 4, B, T, cin, cout, 5, 0, None, _stream()) == 0
        assert _rel(dw, wd.grad) < TOL
        assert _rel(db, dy.double().sum((0, 1))) < TOL


def test_weight_split_batched_equals_single():
    from transformertts_amd import _lib
    from transformertts_amd.ops import _p, _stream
    lib = _lib.load()
    specs = [(_rand(256, 128, seed=1), 256, 128, 0, 0, 0), (_rand(256, 128, seed=2), 128, 256, 1, 0, 0),
             (_rand(64, 32, 5, seed=3), 64, 160, 2, 32, 5), (_rand(64, 32, 5, seed=4), 32, 320, 3, 64, 5),
             # fp16x3 images: two f16 planes + a tail holding max|w| (the pre-scale follows from it)
             (_rand(256, 128, seed=5) * 40.0, 256, 128, 4, 0, 0), (_rand(256, 128, seed=6) * 1e-5, 128, 256, 5, 0, 0),
             (_rand(64, 32, 5, seed=7), 64, 160, 6, 32, 5), (_rand(64, 64, 5, seed=8), 64, 320, 7, 64, 5),
             # channel counts that are no multiple of 32: the image pads every tap (every row of a linear weight) with zeros
             (_rand(256, 80, seed=9), 256, 80, 4, 0, 0), (_rand(80, 256, seed=10), 256, 80, 5, 0, 0),
             (_rand(64, 80, 5, seed=11), 64, 400, 6, 80, 5), (_rand(80, 64, 5, seed=12), 64, 400, 7, 80, 5)]
    single, batched, rows, blk = [], [], [], 0
    for w, R, C, mode, c2, taps in specs:
        nbytes = lib.ttts_split_image_bytes(R, C, mode, c2, taps)
        assert nbytes == (6 * R * C if mode < 4 else 4 * R * (taps * ((c2 + 31) // 32 * 32) if mode >= 6 else (C + 31) // 32 * 32) + 16)
        a = torch.full(((nbytes + 1) // 2,), 0x7e00, dtype=torch.int16, device=_dev())      # (f16 NaN: the padding must be WRITTEN)
        b = torch.zeros_like(a)
        assert lib.ttts_weight_split(_p(w), _p(a), R, C, mode, c2, taps, _stream()) == 0
        single.append(a); batched.append(b)
        rows.append([w.data_ptr(), b.data_ptr(), R, C, mode, c2, taps, blk])
        blk += lib.ttts_weight_split_units(R, C, mode, c2)
    table = torch.tensor(rows, dtype=torch.int64).to(_dev())
    assert lib.ttts_weight_split_batched(_p(table), len(rows), blk, _stream()) == 0
    torch.cuda.synchronize()
    for a, b in zip(single, batched):
        assert torch.equal(a, b)
    # the three planes add back to the weight (exact to 2^-25): plane image is [C/16][plane][R][16]
    w, R, C = specs[0][0], 256, 128
    pl = single[0].view(torch.bfloat16).view(C // 16, 3, R, 16).float().sum(1)          # (C/16, R, 16)
    back = pl.permute(1, 0, 2).reshape(R, C)
    assert (back - w).abs().max().item() <= 2.0 ** -24 * w.abs().max().item()
    assert lib.ttts_weight_split(_p(w), _p(single[0]), 256, 120, 0, 0, 0, _stream()) != 0    # cols must be a multiple of 16
    # fp16x3 image of specs[4]: tail = max|w|, planes = w * 2^k with max|w| * 2^k in [2^11, 2^12), hi + lo exact to 2^-22
    w, R, C = specs[4][0], 256, 128
    img = single[4]
    tail = img.view(torch.uint8)[R * C * 4: R * C * 4 + 4].view(torch.float32)
    assert tail.item() == w.abs().max().item()
    sc = 2.0 ** (11 - int(np.floor(np.log2(tail.item()))))
    pl = img[: 2 * R * C].view(torch.float16).view(C // 32, 2, R, 32).double().sum(1).permute(1, 0, 2).reshape(R, C)
    assert 2048.0 <= tail.item() * sc < 4096.0
    assert (pl / sc - w.double()).abs().max().item() <= 2.0 ** -21 * w.abs().max().item()
    # padded images: conv (cout 64, cin 80, 5 taps) -> [5 * 96 / 32][2][64][32]; channels 80..95 of every tap are exact zeros
    w, R = specs[10][0], 64
    img = single[10]
    Cp = 5 * 96
    tail = img.view(torch.uint8)[R * Cp * 4: R * Cp * 4 + 4].view(torch.float32)
    assert tail.item() == w.abs().max().item()
    sc = 2.0 ** (11 - int(np.floor(np.log2(tail.item()))))
    pl = img[: 2 * R * Cp].view(torch.float16).view(Cp // 32, 2, R, 32).double().sum(1).permute(1, 0, 2).reshape(R, 5, 96)
    assert torch.equal(pl[:, :, 80:], torch.zeros_like(pl[:, :, 80:]))
    assert (pl[:, :, :80] / sc - w.double().permute(0, 2, 1)).abs().max().item() <= 2.0 ** -21 * w.abs().max().item()


def _h3_image_ref(w, mode, scale_exp):
    """The fp16x3 image of ttts_weight_split modes 4-7 in torch: B[r][c'] (c' in the tap-padded image) as hi / lo f16 planes
    of w * 2^k laid [c'/32][plane][r][32].  Bit-exact restatement (float32 arithmetic, round-to-nearest f16 casts)."""
    if mode == 4: B = w[:, None, :]                          # (N, K) -> rows N, one tap, channels K
    elif mode == 5: B = w.t()[:, None, :]                    # rows K, channels N
    elif mode == 6: B = w.permute(0, 2, 1)                   # (co, ci, tap) -> rows co, taps, channels ci
    else: B = w.permute(1, 2, 0)                             # rows ci, taps, channels co
    R, taps, ch = B.shape
    pad = (ch + 31) // 32 * 32
    img = torch.zeros(R, taps, pad, device=w.device)
    img[:, :, :ch] = B * (2.0 ** scale_exp)
    img = img.reshape(R, taps * pad)
    hi = img.half()
    lo = (img - hi.float()).half()
    planes = torch.stack([hi, lo], 0).view(2, R, taps * pad // 32, 32).permute(2, 0, 1, 3).contiguous()
    return planes.view(torch.int16).flatten(), R, taps * pad


def test_fp16x3_weight_images_bit_exact():
    """Every mode of the tiled split (32-row x 32-channel tiles through LDS) against the torch restatement: rows and channels
    that are no multiples of 32, a 9-tap kernel (two passes of 8 taps), a source that is only 4-byte aligned, tiny and
    large magnitudes; single and batched entry points."""
    from transformertts_amd import _lib
    from transformertts_amd.ops import _p, _stream
    lib = _lib.load()
    big = _rand(4 * 1024 * 80 + 1, seed=77)
    specs = [(_rand(256, 128, seed=1) * 40.0, 4), (_rand(1024, 256, seed=2) * 1e-5, 5), (_rand(80, 256, seed=3), 4),
             (_rand(80, 256, seed=4), 5), (_rand(256, 80, seed=5), 4), (_rand(256, 80, seed=6), 5), (_rand(36, 44, seed=7), 4),
             (_rand(36, 44, seed=8), 5), (_rand(256, 256, 5, seed=9), 6), (_rand(256, 256, 5, seed=10), 7),
             (_rand(80, 256, 5, seed=11), 6), (_rand(80, 256, 5, seed=12), 7), (_rand(256, 80, 5, seed=13), 6),
             (_rand(256, 80, 5, seed=14), 7), (_rand(48, 36, 9, seed=15) * 3e4, 6), (_rand(48, 36, 9, seed=16), 7),
             (_rand(40, 12, 3, seed=17), 6), (_rand(40, 12, 1, seed=18), 7),
             (big[1:].view(1024, 320), 4), (big[1:].view(1024, 64, 5), 7)]          # data_ptr % 16 == 4
    # the images of ONE weight, adjacent in the table as ops.PlaneTable lists them: the batched refresh measures max|w| once (the
    # first entry) and the others copy it -- forward / data-gradient, row-major / K16-major (modes 8, 9: against the single entry point)
    shared, shared_c = _rand(512, 256, seed=20) * 7.0, _rand(96, 64, 5, seed=21) * 1e-3
    specs += [(shared, 5), (shared, 4), (shared, 9), (shared, 8), (shared_c, 7), (shared_c, 6)]
    imgs, rows, blk = [], [], 0
    for w, mode in specs:
        if mode in (4, 8): R, C, c2, taps = w.shape[0], w.shape[1], 0, 0
        elif mode in (5, 9): R, C, c2, taps = w.shape[1], w.shape[0], 0, 0
        elif mode == 6: R, C, c2, taps = w.shape[0], w.shape[1] * w.shape[2], w.shape[1], w.shape[2]
        else: R, C, c2, taps = w.shape[1], w.shape[0] * w.shape[2], w.shape[0], w.shape[2]
        amax = w.abs().max().item()
        k = 11 - int(np.floor(np.log2(amax)))
        ref, Rr, Cp = _h3_image_ref(w, mode - 4 if mode >= 8 else mode, k)
        assert Rr == R and lib.ttts_split_image_bytes(R, C, mode, c2, taps) == 4 * R * Cp + 16
        a = torch.full((2 * R * Cp + 8,), 0x7e00, dtype=torch.int16, device=_dev())
        b = torch.full_like(a, 0x7e00)
        assert lib.ttts_weight_split(_p(w), _p(a), R, C, mode, c2, taps, _stream()) == 0
        rows.append([w.data_ptr(), b.data_ptr(), R, C, mode, c2, taps, blk])
        blk += lib.ttts_weight_split_units(R, C, mode, c2)
        imgs.append((a, b, ref, amax, (tuple(w.shape), mode)))
    table = torch.tensor(rows, dtype=torch.int64).to(_dev())
    assert lib.ttts_weight_split_batched(_p(table), len(rows), blk, _stream()) == 0
    torch.cuda.synchronize()
    for a, b, ref, amax, what in imgs:
        n_img = ref.numel() + 2                                 # planes + the float that holds max|w| (the rest of the tail is padding)
        assert torch.equal(a[:n_img], b[:n_img]), what          # single and batched entry points: the same bits
        for img in (a, b):
            if what[1] < 8:                                     # (the K16-major layouts hold the same values in another order)
                assert torch.equal(img[: ref.numel()], ref), what
            assert img[ref.numel():ref.numel() + 2].view(torch.float32).item() == amax, what


@pytest.mark.parametrize("causal,Tq,Tk,lens", [(1, 200, 200, [200, 131, 64]), (0, 150, 70, [70, 33, 1]), (0, 33, 129, [129, 128, 5])])
def test_attention_forms_agree(causal, Tq, Tk, lens):
    """fp32-MFMA and split-precision attention (forward, weights output, backward) against an fp64 reference, dropout off,
    and bit-identical dropout masks between the two forms with dropout on."""
    from transformertts_amd import _lib
    from transformertts_amd.ops import _p, _off, _stream
    lib = _lib.load()
    B, H, d = len(lens), 2, 128
    q, kv, do = _rand(B, Tq, d, seed=1), _rand(B, Tk, 2 * d, seed=2), _rand(B, Tq, d, seed=3)
    kl = torch.tensor(lens, dtype=torch.int64, device=_dev())
    qd = q.double().view(B, Tq, H, 64).transpose(1, 2).requires_grad_()
    kd = kv[..., :d].double().reshape(B, Tk, H, 64).transpose(1, 2).requires_grad_()
    vd = kv[..., d:].double().reshape(B, Tk, H, 64).transpose(1, 2).requires_grad_()
    s = qd @ kd.transpose(-1, -2) / 8.0
    mask = torch.arange(Tk, device=_dev())[None, None, None, :] >= kl[:, None, None, None]
    if causal:
        mask = mask | (torch.arange(Tk, device=_dev())[None, :] > torch.arange(Tq, device=_dev())[:, None])
    p_ref = torch.softmax(s.masked_fill(mask, float("-inf")), -1)
    o_ref = (p_ref @ vd).transpose(1, 2).reshape(B, Tq, d)
    o_ref.backward(do.double())
    dq_ref = qd.grad.transpose(1, 2).reshape(B, Tq, d)
    dkv_ref = torch.cat([kd.grad.transpose(1, 2).reshape(B, Tk, d), vd.grad.transpose(1, 2).reshape(B, Tk, d)], -1)
    outs = {}
    for name in ("", "_x6"):
        fwd, bwd = getattr(lib, "ttts_attention_fwd" + name), getattr(lib, "ttts_attention_bwd" + name)
        for p_drop in (0.0, 0.25):
            o = torch.empty(B, Tq, d, device=_dev()); lse = torch.empty(B, H, Tq, device=_dev())
            attn = None if causal else torch.empty(B, H, Tq, Tk, device=_dev())
            assert fwd(_p(q), _off(kv, 0), _off(kv, d), _p(o), _p(lse), _p(attn), _p(kl), B, H, Tq, Tk, d, 2 * d, 2 * d, d,
                       causal, 0.125, p_drop, 99, None, _stream()) == 0
            dq, dkv, delta = torch.empty_like(q), torch.empty_like(kv), torch.empty_like(lse)
            assert bwd(_p(q), _off(kv, 0), _off(kv, d), _p(o), _p(do), _p(lse), _p(delta), _p(dq), _off(dkv, 0), _off(dkv, d),
                       _p(kl), B, H, Tq, Tk, d, 2 * d, 2 * d, d, d, 2 * d, 2 * d, causal, 0.125, p_drop, 99, None, _stream()) == 0
            outs[(name, p_drop)] = (o, attn, dq, dkv)
            if p_drop == 0.0:
                assert _rel(o, o_ref) < TOL and _rel(dq, dq_ref) < TOL and _rel(dkv, dkv_ref) < TOL
                if attn is not None:
                    assert _rel(attn, p_ref) < TOL
    a, b = outs[("", 0.25)], outs[("_x6", 0.25)]
    assert _rel(b[0], a[0]) < TOL and _rel(b[2], a[2]) < TOL and _rel(b[3], a[3]) < TOL
    if a[1] is not None:      # same counter-based mask in both forms: identical zero pattern
        assert torch.equal(a[1] == 0, b[1] == 0)
        # ... and the BACKWARD kernels regenerate exactly the mask the forward applied: an fp64 reference that drops the
        # weights the returned (post-dropout) weights show as dropped must reproduce output and all three gradients
        keep = (a[1] != 0) | mask.expand_as(a[1])            # masked keys are zero with or without dropout
        for t in (qd, kd, vd):
            t.grad = None
        pd = torch.softmax((qd @ kd.transpose(-1, -2) / 8.0).masked_fill(mask, float("-inf")), -1) * keep / 0.75
        od = (pd @ vd).transpose(1, 2).reshape(B, Tq, d)
        od.backward(do.double())
        dq2 = qd.grad.transpose(1, 2).reshape(B, Tq, d)
        dkv2 = torch.cat([kd.grad.transpose(1, 2).reshape(B, Tk, d), vd.grad.transpose(1, 2).reshape(B, Tk, d)], -1)
        for o, attn, dq, dkv in (a, b):
            assert _rel(o, od) < TOL and _rel(dq, dq2) < TOL and _rel(dkv, dkv2) < TOL, (_rel(o, od), _rel(dq, dq2), _rel(dkv, dkv2))
        drop_rate = 1.0 - float(keep[~mask.expand_as(keep)].double().mean())
        assert abs(drop_rate - 0.25) < 0.02, drop_rate


@pytest.mark.parametrize("causal,Tq,Tk,lens", [(1, 200, 200, [200, 131, 64]), (0, 150, 70, [70, 33, 1]), (0, 33, 129, [129, 128, 5]),
                                               (1, 870, 870, [870, 500])])
@pytest.mark.parametrize("qk_scale", [1.0, 6.0])
def test_fp16x3_attention_forward(causal, Tq, Tk, lens, qk_scale):
    """The fp16x3 forward kernel against fp64 (context, per-head weights, lse via the bf16x6 backward that consumes it),
    with peaked softmaxes (scores of +-100 at qk_scale 6), and the same counter-based dropout mask as the bf16x6 form."""
    from transformertts_amd import _lib, ops
    from transformertts_amd.ops import _p, _off, _stream
    lib = _lib.load()
    B, H, d = len(lens), 2, 128
    q, kv, do = _rand(B, Tq, d, seed=1) * qk_scale, _rand(B, Tk, 2 * d, seed=2), _rand(B, Tq, d, seed=3)
    kv[..., :d] *= qk_scale
    kl = torch.tensor(lens, dtype=torch.int64, device=_dev())
    qd = q.double().view(B, Tq, H, 64).transpose(1, 2).requires_grad_()
    kd = kv[..., :d].double().reshape(B, Tk, H, 64).transpose(1, 2).requires_grad_()
    vd = kv[..., d:].double().reshape(B, Tk, H, 64).transpose(1, 2).requires_grad_()
    s = qd @ kd.transpose(-1, -2) / 8.0
    mask = torch.arange(Tk, device=_dev())[None, None, None, :] >= kl[:, None, None, None]
    if causal:
        mask = mask | (torch.arange(Tk, device=_dev())[None, :] > torch.arange(Tq, device=_dev())[:, None])
    p_ref = torch.softmax(s.masked_fill(mask, float("-inf")), -1)
    o_ref = (p_ref @ vd).transpose(1, 2).reshape(B, Tq, d)
    o_ref.backward(do.double())
    dq_ref = qd.grad.transpose(1, 2).reshape(B, Tq, d)
    o = torch.empty(B, Tq, d, device=_dev()); lse = torch.empty(B, H, Tq, device=_dev())
    attn = None if causal else torch.empty(B, H, Tq, Tk, device=_dev())
    qa, kva, oslots = ops._amax(q), ops._amax(kv), torch.zeros(1024, device=_dev())
    assert lib.ttts_attention_fwd_h3(_p(q), _off(kv, 0), _off(kv, d), _p(o), _p(lse), _p(attn), _p(kl), B, H, Tq, Tk, d, 2 * d,
                                     2 * d, d, causal, 0.125, 0.0, 0, None, _p(qa), _p(kva), _p(kva), _p(oslots), None, _stream()) == 0
    assert _rel(o, o_ref) < TOL, _rel(o, o_ref)
    assert oslots.max().item() == o.abs().max().item()
    if attn is not None:
        assert _rel(attn, p_ref) < TOL
        assert float(attn.sum(-1).sub(1).abs().max()) < 1e-5
    dq, dkv, delta = torch.empty_like(q), torch.empty_like(kv), torch.empty_like(lse)
    assert lib.ttts_attention_bwd_x6(_p(q), _off(kv, 0), _off(kv, d), _p(o), _p(do), _p(lse), _p(delta), _p(dq), _off(dkv, 0),
                                     _off(dkv, d), _p(kl), B, H, Tq, Tk, d, 2 * d, 2 * d, d, d, 2 * d, 2 * d, causal, 0.125, 0.0, 0, None,
                                     _stream()) == 0
    if qk_scale == 1.0:                                                 # the backward recomputes P from this forward's lse
        assert _rel(dq, dq_ref) < TOL
    o6 = torch.empty_like(o); a6 = None if causal else torch.empty_like(attn); l6 = torch.empty_like(lse)
    for f, oo, aa, ll, extra in ((lib.ttts_attention_fwd_h3, o, attn, lse, (_p(qa), _p(kva), _p(kva), None, None)),
                                 (lib.ttts_attention_fwd_x6, o6, a6, l6, ())):
        assert f(_p(q), _off(kv, 0), _off(kv, d), _p(oo), _p(ll), _p(aa), _p(kl), B, H, Tq, Tk, d, 2 * d, 2 * d, d, causal, 0.125, 0.25,
                 99, None, *extra, _stream()) == 0
    assert _rel(o, o6) < TOL and _rel(lse, l6) < TOL
    if attn is not None and qk_scale == 1.0:       # (peaked softmaxes underflow to 0 at slightly different places)
        assert torch.equal(attn == 0, a6 == 0)


@pytest.mark.parametrize("causal,Tq,Tk,lens", [(1, 200, 200, [200, 131, 64]), (0, 150, 70, [70, 33, 1]), (0, 33, 129, [129, 128, 5]),
                                               (1, 870, 870, [870, 500])])
@pytest.mark.parametrize("mag,grow", [(1.0, False), (3e-7, False), (3e-7, True)])
def test_fp16x3_attention_backward(causal, Tq, Tk, lens, mag, grow):
    """dQ, dK, dV of the fp16x3 backward against fp64 for gradients of realistic magnitude (3e-7), and with inputs that FORCE
    the rare branch of the lane-local dS scale tracking (`grow`: the last keys' V rows and the last queries' dO rows are
    1000 x larger, so late tiles outgrow the scale chosen from the first ones and the accumulators are rescaled);
    dropout on: same masks and results as the bf16x6 form."""
    from transformertts_amd import _lib, ops
    from transformertts_amd.ops import _p, _off, _stream
    lib = _lib.load()
    B, H, d = len(lens), 2, 128
    q, kv, do = _rand(B, Tq, d, seed=1), _rand(B, Tk, 2 * d, seed=2), _rand(B, Tq, d, seed=3) * mag
    if grow:
        kv[:, Tk - Tk // 4:, d:] *= 100.0
        do[:, Tq - Tq // 4:] *= 1000.0
    kl = torch.tensor(lens, dtype=torch.int64, device=_dev())
    qd = q.double().view(B, Tq, H, 64).transpose(1, 2).requires_grad_()
    kd = kv[..., :d].double().reshape(B, Tk, H, 64).transpose(1, 2).requires_grad_()
    vd = kv[..., d:].double().reshape(B, Tk, H, 64).transpose(1, 2).requires_grad_()
    sc = qd @ kd.transpose(-1, -2) / 8.0
    mask = torch.arange(Tk, device=_dev())[None, None, None, :] >= kl[:, None, None, None]
    if causal:
        mask = mask | (torch.arange(Tk, device=_dev())[None, :] > torch.arange(Tq, device=_dev())[:, None])
    p_ref = torch.softmax(sc.masked_fill(mask, float("-inf")), -1)
    o_ref = (p_ref @ vd).transpose(1, 2).reshape(B, Tq, d)
    o_ref.backward(do.double())
    dq_ref = qd.grad.transpose(1, 2).reshape(B, Tq, d)
    dkv_ref = torch.cat([kd.grad.transpose(1, 2).reshape(B, Tk, d), vd.grad.transpose(1, 2).reshape(B, Tk, d)], -1)

    qa, kva = ops._amax(q), ops._amax(kv)

    def run(fwd, bwd, p_drop, h3):
        o = torch.empty(B, Tq, d, device=_dev()); lse = torch.empty(B, H, Tq, device=_dev())
        rowstat = torch.empty(3, B, H, Tq, device=_dev())
        extra = (_p(qa), _p(kva), _p(kva), None, _p(rowstat)) if h3 else ()
        assert fwd(_p(q), _off(kv, 0), _off(kv, d), _p(o), _p(lse), None, _p(kl), B, H, Tq, Tk, d, 2 * d, 2 * d, d, causal,
                   0.125, p_drop, 99, None, *extra, _stream()) == 0
        dq, dkv, delta = torch.empty_like(q), torch.empty_like(kv), torch.empty_like(lse)
        args = (_p(q), _off(kv, 0), _off(kv, d), _p(o), _p(do), _p(lse), _p(delta), _p(dq), _off(dkv, 0), _off(dkv, d), _p(kl),
                B, H, Tq, Tk, d, 2 * d, 2 * d, d, d, 2 * d, 2 * d, causal, 0.125, p_drop, 99, None)
        if h3:
            sq, sk = torch.zeros(1024, device=_dev()), torch.zeros(1024, device=_dev())
            assert bwd(*args, _p(ops._amax(do)), _p(sq), _p(sk), _p(qa), _p(kva), _p(kva), _p(rowstat), _stream()) == 0
            dq2, dkv2 = torch.empty_like(q), torch.empty_like(kv)           # and from lse alone (no row statistics)
            args2 = args[:7] + (_p(dq2), _off(dkv2, 0), _off(dkv2, d)) + args[10:]
            assert bwd(*args2, _p(ops._amax(do)), None, None, _p(qa), _p(kva), _p(kva), None, _stream()) == 0
            assert _rel(dq2, dq) < TOL and _rel(dkv2, dkv) < TOL
            assert sq.max().item() == dq.abs().max().item() and sk.max().item() == dkv.abs().max().item()
        else:
            assert bwd(*args, _stream()) == 0
        return dq, dkv
    dq, dkv = run(lib.ttts_attention_fwd_h3, lib.ttts_attention_bwd_h3, 0.0, True)
    assert _rel(dq, dq_ref) < TOL, _rel(dq, dq_ref)
    assert _rel(dkv[..., :d], dkv_ref[..., :d]) < TOL and _rel(dkv[..., d:], dkv_ref[..., d:]) < TOL, \
        (_rel(dkv[..., :d], dkv_ref[..., :d]), _rel(dkv[..., d:], dkv_ref[..., d:]))
    if grow:     # the small early rows on their own, not drowned by the large late ones
        n = Tq - Tq // 4
        assert _rel(dq[:, :n], dq_ref[:, :n]) < 4 * TOL, _rel(dq[:, :n], dq_ref[:, :n])
    a = run(lib.ttts_attention_fwd_h3, lib.ttts_attention_bwd_h3, 0.25, True)
    b = run(lib.ttts_attention_fwd_x6, lib.ttts_attention_bwd_x6, 0.25, False)
    assert _rel(a[0], b[0]) < TOL and _rel(a[1], b[1]) < TOL


@pytest.mark.parametrize("vs,qs,ks", [(1e4, 1.0, 1.0), (1e-5, 1e3, 1e-3), (1e6, 1e-4, 1e4), (1.0, 3e-3, 3e2),
                                      (1e3, 1e3, 1e3)])       # the last: scores of +-1e6, softmax one-hot (an un-normalised
#                                                               decoder input 1000 x too large does this to layer 0)
@pytest.mark.parametrize("causal", [0, 1])
def test_fp16x3_attention_any_magnitude(vs, qs, ks, causal):
    """Q, K and V of any magnitude (the scores stay ordinary because qs * ks = 1): the fp16x3 forward and backward take
    their pre-scales from the operands' measured maxima, so nothing saturates and the results stay fp32-grade.  Round 2's
    fixed 2^4 pre-scale made |V| >= 4096 an inf and lost |Q| ~ 1e-3 below the f16 window."""
    from transformertts_amd import _lib, ops
    from transformertts_amd.ops import _p, _off, _stream
    lib = _lib.load()
    B, H, d, T = 2, 2, 128, 160
    lens = [160, 97]
    q, kv, do = _rand(B, T, d, seed=1) * qs, _rand(B, T, 2 * d, seed=2), _rand(B, T, d, seed=3) * 1e-6
    kv[..., :d] *= ks
    kv[..., d:] *= vs
    kv[0, 5, d + 3] *= 50.0                                    # an outlier value on top
    kl = torch.tensor(lens, dtype=torch.int64, device=_dev())
    qd = q.double().view(B, T, H, 64).transpose(1, 2).requires_grad_()
    kd = kv[..., :d].double().reshape(B, T, H, 64).transpose(1, 2).requires_grad_()
    vd = kv[..., d:].double().reshape(B, T, H, 64).transpose(1, 2).requires_grad_()
    mask = torch.arange(T, device=_dev())[None, None, None, :] >= kl[:, None, None, None]
    if causal:
        mask = mask | (torch.arange(T, device=_dev())[None, :] > torch.arange(T, device=_dev())[:, None])
    p_ref = torch.softmax((qd @ kd.transpose(-1, -2) / 8.0).masked_fill(mask, float("-inf")), -1)
    o_ref = (p_ref @ vd).transpose(1, 2).reshape(B, T, d)
    o_ref.backward(do.double())
    dq_ref = qd.grad.transpose(1, 2).reshape(B, T, d)
    dk_ref, dv_ref = kd.grad.transpose(1, 2).reshape(B, T, d), vd.grad.transpose(1, 2).reshape(B, T, d)
    # q and kv come from different producers in cross-attention; in self-attention one array covers all three
    qa, ka, va = ops._amax(q), ops._amax(kv[..., :d].contiguous()), ops._amax(kv[..., d:].contiguous())
    o, lse = torch.empty(B, T, d, device=_dev()), torch.empty(B, H, T, device=_dev())
    rowstat = torch.empty(3, B, H, T, device=_dev())
    assert lib.ttts_attention_fwd_h3(_p(q), _off(kv, 0), _off(kv, d), _p(o), _p(lse), None, _p(kl), B, H, T, T, d, 2 * d, 2 * d,
                                     d, causal, 0.125, 0.0, 0, None, _p(qa), _p(ka), _p(va), None, _p(rowstat), _stream()) == 0
    assert torch.isfinite(o).all() and _rel(o, o_ref) < TOL, _rel(o, o_ref)
    dq, dkv, delta = torch.empty_like(q), torch.empty_like(kv), torch.empty_like(lse)
    assert lib.ttts_attention_bwd_h3(_p(q), _off(kv, 0), _off(kv, d), _p(o), _p(do), _p(lse), _p(delta), _p(dq), _off(dkv, 0),
                                     _off(dkv, d), _p(kl), B, H, T, T, d, 2 * d, 2 * d, d, d, 2 * d, 2 * d, causal, 0.125, 0.0, 0, None,
                                     _p(ops._amax(do)), None, None, _p(qa), _p(ka), _p(va), _p(rowstat), _stream()) == 0
    assert torch.isfinite(dq).all() and torch.isfinite(dkv).all()
    assert _rel(dkv[..., d:], dv_ref) < TOL, _rel(dkv[..., d:], dv_ref)
    if qs * ks < 100:
        assert _rel(dq, dq_ref) < TOL and _rel(dkv[..., :d], dk_ref) < TOL, (_rel(dq, dq_ref), _rel(dkv[..., :d], dk_ref))
    else:
        # scores of +-1e6: every softmax row is exactly one-hot, the true dQ and dK are exactly zero, and what a recomputing
        # backward returns is the rounding of dS = P (dP - delta): 2^-22 |dO| |V| per weight, times |K| (resp. |Q|) / 8
        noise = 2.0 ** -20 * float(do.abs().max()) * float(kv[..., d:].abs().max()) * 8.0
        assert float(dq_ref.abs().max()) == 0.0
        assert float(dq.abs().max()) < noise * float(kv[..., :d].abs().max()) and float(dkv[..., :d].abs().max()) < noise * float(q.abs().max())
    # one shared array for a packed projection whose parts differ by 1e3 .. 1e8 still works (graceful: the small part
    # keeps an absolute error of 2^-37 of the largest element)
    am_all = torch.maximum(torch.maximum(qa, ka), va)
    if max(vs, qs, ks) / min(vs, qs, ks) <= 1e4:
        assert lib.ttts_attention_fwd_h3(_p(q), _off(kv, 0), _off(kv, d), _p(o), _p(lse), None, _p(kl), B, H, T, T, d, 2 * d,
                                         2 * d, d, causal, 0.125, 0.0, 0, None, _p(am_all), _p(am_all), _p(am_all), None, None,
                                         _stream()) == 0
        assert _rel(o, o_ref) < 4 * TOL, _rel(o, o_ref)


def _base_module(seed=5, cfg_name="base"):
    from oracle.spec import model_config, fill_state
    from transformertts_amd.lightning_module import LightningModule
    cfg = model_config(cfg_name)
    config = {"model": dict(cfg, device="cuda"), "loss": {"stop_weight": 8.0},
              "training": {"num_epochs": 300, "teacher_forcing_mode": "linear", "warmup_steps": 4000,
                           "sync_loss_every_step": False}}
    lm = LightningModule(config).to(_dev())
    lm.model.load_state_dict(fill_state(cfg, seed), strict=True)
    return cfg, lm


@pytest.mark.parametrize("cfg_name,B", [("base", 64), ("scaled", 32)])
def test_full_size_batch_composition_invariance(cfg_name, B):
    """BASELINE batch (64 ragged LJSpeech-shaped utterances; 32 for the scaled model = the per-GPU shard of configs[4];
    eval mode): an utterance's outputs do not depend on what it is batched with -- the masks, the conv / go-frame clipping
    at utterance ends and the padding never leak."""
    from oracle.synth import synth_batch
    cfg, lm = _base_module(cfg_name=cfg_name)
    lm.eval()
    batch = {k: v.to(_dev()) for k, v in synth_batch(B, 100, 870, cfg["n_mels"], cfg["n_phon"], ragged=True, seed=3).items()}
    with torch.no_grad():
        full = lm.model(batch["phoneme"], batch["melspec"], batch["phoneme_lens"], batch["melspec_lens"])
        idx = [0, 17, B // 2 + 8, B - 1]
        pl, ml = batch["phoneme_lens"][idx], batch["melspec_lens"][idx]
        Tp, Tm = int(pl.max()), int(ml.max())
        sub = lm.model(batch["phoneme"][idx][:, :Tp].contiguous(), batch["melspec"][idx][:, :Tm].contiguous(), pl, ml)
    for j, i in enumerate(idx):
        m, p = int(ml[j]), int(pl[j])
        for key in ("pred_melspec", "post_melspec"):
            assert _rel(sub[key][j, :m], full[key][i, :m]) < 1e-5, key
        assert _rel(sub["pred_stop"][j, :m], full["pred_stop"][i, :m]) < 1e-5
        a_sub, a_full = sub["alignments"][-1][j, :, :m, :p], full["alignments"][-1][i, :, :m, :p]
        assert _rel(a_sub, a_full) < 1e-5
        assert torch.allclose(a_full.sum(-1), torch.ones_like(a_full.sum(-1)), atol=1e-5)     # rows of weights sum to 1
        assert float(full["alignments"][-1][i, :, :m, p:].abs().max()) == 0.0 if p < full["alignments"][-1].size(-1) else True


@pytest.mark.parametrize("cfg_name,B", [("base", 64), ("scaled", 32)])
def test_full_size_training_step_is_reproducible(cfg_name, B):
    """Two BASELINE-size training steps from the same state and seeds: bit-identical loss and gradient bucket (fixed-order
    reductions everywhere, no floating-point atomics), and a different dropout seed changes them.  Also for the scaled
    model at its per-GPU shard size (configs[4]: batch 32 x 870 frames, d_model 512, 6+6 layers, 8 heads)."""
    from oracle.synth import synth_batch
    from transformertts_amd import ops
    from transformertts_amd.parallel import FlatGradBucket
    cfg, lm = _base_module(cfg_name=cfg_name)
    lm.train()
    bucket = FlatGradBucket(lm.parameters())
    batch = {k: v.to(_dev()) for k, v in synth_batch(B, 100, 870, cfg["n_mels"], cfg["n_phon"], ragged=True, seed=4).items()}
    bn0 = {k: v.clone() for k, v in lm.state_dict().items() if "running" in k or "num_batches" in k}

    def run(seed):
        lm.load_state_dict(bn0, strict=False)
        ops.seeds.manual_seed(seed)
        torch.manual_seed(seed)
        bucket.zero()
        loss = lm.training_step(batch, 0)
        loss.backward()
        torch.cuda.synchronize()
        return loss.detach().clone(), bucket.flat.clone()

    try:
        l1, g1 = run(7)
        l2, g2 = run(7)
        l3, g3 = run(8)
    finally:
        ops.seeds.follow_torch()          # later tests draw their masks from torch's seed again
    assert torch.isfinite(l1) and torch.equal(l1, l2) and torch.equal(g1, g2)
    assert not torch.equal(g1, g3)
    assert float(g1.abs().max()) > 0


@pytest.mark.parametrize("M,N,K,scale", [(300, 96, 64, 1.0), (129, 256, 256, 1e-4), (1000, 768, 256, 30.0), (257, 1024, 256, 1.0),
                                         (515, 256, 1024, 1.0), (1157, 260, 512, 1.0)])
def test_image_operand_gemm_agrees_with_fp64(M, N, K, scale):
    """ttts_linear_fwd_h3i / ttts_linear_bwd_data_h3i through the C ABI: the activation arrives as an image (f16 hi / lo planes,
    K16-major, per-ROW power-of-two scale: ttts_act_image), the weight as the K16-major fp16x3 image (ttts_weight_split modes
    8 / 9), both staged by LDS-DMA.  Rows of very different magnitude (x 1e3, all zero) must come out as accurate as their
    neighbours -- that is what the per-row scale is for; ragged M / N edges, every epilogue (bias, residual, relu + dropout,
    relu gate, published maxima)."""
    from transformertts_amd import _lib, ops
    from transformertts_amd.ops import _p, _stream
    lib, dev = _lib.load(), _dev()
    x = _rand(M, K, seed=1)* scale
    x[3] *= 1e3
    x[5] = 0.0
    w, b, res = _rand(N, K, seed=2) * K ** -0.5, _rand(N, seed=3), _rand(M, N, seed=4)

    def image_of(t):
        m, k = t.shape
        img, inv = torch.empty(m * k * 2, dtype=torch.int16, device=dev), torch.empty(m, device=dev)
        _lib.check(lib.ttts_act_image(_p(t), _p(img), _p(inv), m, k, _stream()), "act_image")
        return img, inv

    img, inv = image_of(x)
    y = torch.full((M, N), float("nan"), device=dev)
    am = torch.zeros(ops.AMAX_SLOTS, device=dev)
    _lib.check(lib.ttts_linear_fwd_h3i(_p(img), _p(inv), _p(ops._planes(w, 8, N, K)), _p(b), _p(res), _p(y), M, N, K, 0, 0.0, 0, None,
                                       _p(am), _stream()), "fwd_h3i")
    ref = x.double() @ w.double().t() + b.double() + res.double()
    assert _rel(y, ref) < 1e-6
    rows = (y.double() - ref).norm(dim=1) / ref.norm(dim=1)
    assert float(rows.max()) < 1e-6, float(rows.max())                       # the x 1e3 row, the zero row, their neighbours
    assert float(am.max()) == float(y.abs().max())
    # relu + dropout: the mask is a function of (seed, element index) only -- the same as the fp32-operand kernel draws
    y1, y2 = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
    xa = torch.zeros(ops.AMAX_SLOTS, device=dev)
    _lib.check(lib.ttts_amax_partials(_p(x), x.numel(), _p(xa), _stream()), "amax")
    _lib.check(lib.ttts_linear_fwd_h3i(_p(img), _p(inv), _p(ops._planes(w, 8, N, K)), _p(b), None, _p(y1), M, N, K, 1, 0.1, 77, None,
                                       None, _stream()), "fwd_h3i")
    _lib.check(lib.ttts_linear_fwd_h3(_p(x), _p(ops._planes(w, 4, N, K)), _p(b), None, _p(y2), M, N, K, 1, 0.1, 77, None, 0, 0,
                                      _p(xa), None, _stream()), "fwd_h3")
    refr = torch.relu(x.double() @ w.double().t() + b.double())
    live = (refr.abs() > 1e-4 * refr.abs().max())                            # away from the relu's kink both kernels agree on the mask
    assert torch.equal((y1 == 0)[live], (y2 == 0)[live])
    kept = live & (y1 != 0)
    assert _rel(y1[kept], (refr / 0.9)[kept]) < 1e-6
    if N % 32 == 0:      # data gradient dx[m, k] = dy[m, n] . w[n, k] with the relu gate of the layer below and a skip gradient
        dy, h, skip = _rand(M, N, seed=5) * 1e-5, torch.relu(_rand(M, K, seed=6)), _rand(M, K, seed=7) * 1e-5
        dimg, dinv = image_of(dy)
        dx = torch.empty(M, K, device=dev)
        _lib.check(lib.ttts_linear_bwd_data_h3i(_p(dimg), _p(dinv), _p(ops._planes(w, 9, K, N)), _p(skip), _p(dx), M, N, K, _p(h),
                                                1.0 / 0.9, None, _stream()), "bwd_h3i")
        refd = (dy.double() @ w.double()) * (h > 0).double() / 0.9 + skip.double()
        assert _rel(dx, refd) < 1e-6


@pytest.mark.parametrize("raw", [True, False], ids=["fp32-by-dma", "image"])
@pytest.mark.parametrize("M,N,K", [(55680, 1024, 256), (70001, 256, 256), (66000, 1024, 256), (55680, 256, 1024),
                                   (70001, 256, 1024), (16512, 1020, 256)])
def test_dma_gemm_streams_across_tiles(M, N, K, raw):
    """gemm_h3i_kernel (128 x 256 tile, both operands by LDS-DMA; csrc/gemm_h3i.hip) in the regime it runs in at the BASELINE
    batch: MORE tiles than the 512 resident workgroups, so every workgroup walks several tiles, the operand stream crosses the
    tile boundary under the epilogue and the residual / gate operands are loaded with the next tile's k-tiles requested --
    where round 4's ordering race lived (a few hundred wrong elements per launch, only where a workgroup has a next tile).
    Through the C ABI, both operand forms (ttts_linear_*_h3d: fp32 rows split in place in LDS; ttts_linear_*_h3i: an image) and
    every epilogue the training step instantiates: <res, gate, drop> = <0,0,0> (+ relu, published maxima), <1,0,0>, <1,0,1>,
    <0,0,1>, <0,1,0>, <1,1,0>.  The WHOLE output is compared with fp64 (row chunks), a second run with the caches disturbed
    must give the same bits, and the published maximum must be max|y| exactly.  (55 680, 256, 1024) has 435 tiles -- one per
    workgroup: the FFN2 forward of the step as it stands; (70 001, 256, 1024) is its multi-tile neighbour; (16 512, 1020, 256)
    has a ragged last column block.)"""
    from transformertts_amd import _lib, ops
    from transformertts_amd.ops import _p, _stream
    lib, dev = _lib.load(), _dev()
    x, w, b = _rand(M, K, seed=31), _rand(N, K, seed=32, scale=K ** -0.5), _rand(N, seed=33)
    res = _rand(M, N, seed=34)
    xa = ops._amax(x)
    pl = ops._planes(w, 8, N, K)
    if not raw:
        img, inv = torch.empty(M * K * 2, dtype=torch.int16, device=dev), torch.empty(M, device=dev)
        _lib.check(lib.ttts_act_image(_p(x), _p(img), _p(inv), M, K, _stream()), "act_image")

    def fwd(y, residual=None, act=0, p=0.0, seed=0, y_am=None):
        if raw:
            return lib.ttts_linear_fwd_h3d(_p(x), _p(pl), _p(b), _p(residual), _p(y), M, N, K, act, p, seed, None, _p(xa), _p(y_am), _stream())
        return lib.ttts_linear_fwd_h3i(_p(img), _p(inv), _p(pl), _p(b), _p(residual), _p(y), M, N, K, act, p, seed, None, _p(y_am), _stream())

    w64, b64 = w.double(), b.double()
    CH = 8192

    def whole(y, fn):
        """rel-L2 of y against fn(row slice) -> fp64 reference rows, over every row of the output"""
        num = den = 0.0
        for r0 in range(0, M, CH):
            sl = slice(r0, min(M, r0 + CH))
            ref = fn(sl)
            num += float(((y[sl].double() - ref) ** 2).sum())
            den += float((ref ** 2).sum())
        return (num / den) ** 0.5

    def disturb():
        torch.randn(1 << 24, device=dev)                      # 64 MB of something else through L2 / the Infinity Cache

    # <0,0,0>: bias + relu, published maxima, same bits on a second run
    y = torch.full((M, N), float("nan"), device=dev)
    slots = torch.zeros(ops.AMAX_SLOTS, device=dev)
    assert fwd(y, act=1, y_am=slots) == 0
    assert torch.isfinite(y).all()
    assert whole(y, lambda sl: torch.relu(x[sl].double() @ w64.t() + b64)) < TOL
    assert slots.max().item() == y.abs().max().item()
    y2 = torch.full((M, N), 3.0, device=dev)
    disturb()
    assert fwd(y2, act=1) == 0
    assert torch.equal(y, y2)
    # <1,0,0>: residual
    assert fwd(y, residual=res) == 0
    assert whole(y, lambda sl: x[sl].double() @ w64.t() + b64 + res[sl].double()) < TOL
    disturb()
    assert fwd(y2, residual=res) == 0
    assert torch.equal(y, y2)
    # <1,0,1> and <0,0,1>: dropout (+ residual).  Every element is either the kept value or the dropped one; the keep rate is
    # 1 - p; the mask is a function of (seed, element) only, i.e. the one gemm_h3_kernel draws for the same seed.
    p_drop = 0.1
    yh = torch.empty(M, N, device=dev)
    assert lib.ttts_linear_fwd_h3(_p(x), _p(ops._planes(w, 4, N, K)), _p(b), None, _p(yh), M, N, K, 0, p_drop, 4321, None, 0, 0,
                                  _p(xa), None, _stream()) == 0
    for residual in (res, None):
        assert fwd(y, residual=residual, p=p_drop, seed=4321) == 0
        num = den = kept_n = 0.0
        agree = True
        for r0 in range(0, M, CH):
            sl = slice(r0, min(M, r0 + CH))
            lin = x[sl].double() @ w64.t() + b64
            add = residual[sl].double() if residual is not None else torch.zeros_like(lin)
            d_keep = (y[sl].double() - (lin / (1 - p_drop) + add)).abs()
            d_drop = (y[sl].double() - add).abs()
            kept = d_keep < d_drop
            num += float((torch.where(kept, d_keep, d_drop) ** 2).sum())
            den += float((lin ** 2).sum())
            kept_n += float(kept.sum())
            sure = lin.abs() > 1e-3                                  # (a product that cancels to ~0 looks dropped either way)
            agree = agree and bool(torch.equal(kept[sure], (yh[sl] != 0)[sure]))
        assert (num / den) ** 0.5 < TOL and abs(kept_n / (M * N) - (1 - p_drop)) < 2e-3 and agree
        disturb()
        assert fwd(y2, residual=residual, p=p_drop, seed=4321) == 0
        assert torch.equal(y, y2)
    del yh
    # data gradients with the same output width: dh (M x N) = dy (M x K) . Wt, Wt = weight (K, N) of a Linear N -> K:
    # <0,1,0> the relu gate of h (FFN2's data gradient), <1,1,0> gate + skip gradient
    dy = _rand(M, K, seed=35) * 1e-5
    hgate = torch.relu(_rand(M, N, seed=36))
    skip = _rand(M, N, seed=37) * 1e-5
    wt = w.t().contiguous()                                          # (K, N)
    plt = ops._planes(wt, 9, N, K)
    dya = ops._amax(dy)
    if not raw:
        dimg, dinv = torch.empty(M * K * 2, dtype=torch.int16, device=dev), torch.empty(M, device=dev)
        _lib.check(lib.ttts_act_image(_p(dy), _p(dimg), _p(dinv), M, K, _stream()), "act_image")

    def bwd(dh, residual=None, gate=None, am=None):
        if raw:
            return lib.ttts_linear_bwd_data_h3d(_p(dy), _p(plt), _p(residual), _p(dh), M, K, N, _p(gate), 1.25, _p(dya), _p(am), _stream())
        return lib.ttts_linear_bwd_data_h3i(_p(dimg), _p(dinv), _p(plt), _p(residual), _p(dh), M, K, N, _p(gate), 1.25, _p(am), _stream())

    wt64 = wt.double()
    for residual in (None, skip):
        dslots = torch.zeros(ops.AMAX_SLOTS, device=dev)
        assert bwd(y, residual=residual, gate=hgate, am=dslots) == 0

        def ref_rows(sl):
            r = (dy[sl].double() @ wt64) * (hgate[sl] > 0).double() * 1.25
            return r + residual[sl].double() if residual is not None else r
        assert whole(y, ref_rows) < TOL
        assert dslots.max().item() == y.abs().max().item()
        disturb()
        assert bwd(y2, residual=residual, gate=hgate) == 0
        assert torch.equal(y, y2)
