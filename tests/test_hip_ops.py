"""Per-kernel parity: every C-ABI entry point (called through transformertts_amd.ops) against a plain
torch reference of the same op evaluated in fp64 on the CPU.  Tolerance 2e-5 rel-L2 unless stated
(measured 5e-8 .. 8e-7: fp32 MFMA = exact fp32 fma chains; the end-to-end gate of north_star is 1e-4)."""
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2

pytestmark = pytest.mark.gpu
TOL = 5e-6


def _dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def _g(t):
    return t.detach().to(_dev()).requires_grad_(t.is_floating_point())


@pytest.mark.parametrize("M,N,K,act,res", [(300, 256, 256, 0, False), (1000, 80, 256, 0, True), (777, 1024, 256, 1, False),
                                            (513, 256, 1024, 0, True), (64, 256, 80, 1, False), (4099, 768, 256, 0, False),
                                            (130, 16, 128, 0, False), (24577, 512, 256, 1, False),
                                            (26001, 256, 1024, 0, True), (50003, 80, 256, 0, False)])
def test_linear_fwd_bwd(M, N, K, act, res):
    from transformertts_amd import ops
    x, w, b = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=K ** -0.5), _rand(N, seed=3, scale=0.1)
    r = _rand(M, N, seed=4) if res else None
    dy = _rand(M, N, seed=5)
    xd, wd, bd = x.double().requires_grad_(), w.double().requires_grad_(), b.double().requires_grad_()
    rd = r.double().requires_grad_() if res else None
    ref = F.linear(xd, wd, bd)
    if act:
        ref = F.relu(ref)
    if res:
        ref = ref + rd
    ref.backward(dy.double())
    xg, wg, bg = _g(x), _g(w), _g(b)
    rg = _g(r) if res else None
    y = ops.linear(xg, wg, bg, residual=rg, act=act)
    y.backward(dy.to(_dev()))
    assert rel_l2(y, ref) < TOL
    assert rel_l2(xg.grad, xd.grad) < TOL
    assert rel_l2(wg.grad, wd.grad) < TOL
    assert rel_l2(bg.grad, bd.grad) < TOL
    if res:
        assert rel_l2(rg.grad, rd.grad) < TOL


def test_linear_go_frame_shift():
    from transformertts_amd import ops
    B, T, K, N = 3, 37, 80, 256
    x, w, b = _rand(B, T, K, seed=1), _rand(N, K, seed=2, scale=K ** -0.5), _rand(N, seed=3, scale=0.1)
    dy = _rand(B, T, N, seed=5)
    xs = torch.cat((torch.zeros(B, 1, K), x[:, :-1]), dim=1).double()
    wd, bd = w.double().requires_grad_(), b.double().requires_grad_()
    ref = F.relu(F.linear(xs, wd, bd))
    ref.backward(dy.double())
    wg, bg = _g(w), _g(b)
    y = ops.linear(x.to(_dev()), wg, bg, act=1, row_shift=-1, T=T)
    y.backward(dy.to(_dev()))
    assert rel_l2(y, ref) < TOL
    assert rel_l2(wg.grad, wd.grad) < TOL
    assert rel_l2(bg.grad, bd.grad) < TOL


@pytest.mark.parametrize("B,T,cin,cout,act,training", [(3, 37, 128, 128, 0, True), (2, 300, 80, 256, 2, True),
                                                        (2, 131, 256, 80, 0, True), (4, 50, 256, 256, 2, True),
                                                        (2, 64, 256, 256, 2, False), (5, 1, 16, 128, 2, True),
                                                        (31, 870, 256, 256, 2, True), (61, 433, 80, 256, 0, True)])
def test_conv_bn_fwd_bwd(B, T, cin, cout, act, training):
    from transformertts_amd import ops
    k = 5
    x = _rand(B, T, cin, seed=1)
    w = _rand(cout, cin, k, seed=2, scale=(cin * k) ** -0.5)
    b = _rand(cout, seed=3, scale=0.1)
    gamma = 0.8 + 0.4 * torch.rand(cout, generator=torch.Generator().manual_seed(4))
    beta = _rand(cout, seed=5, scale=0.1)
    rm, rv = _rand(cout, seed=6, scale=0.1), 0.5 + torch.rand(cout, generator=torch.Generator().manual_seed(7))
    dz = _rand(B, T, cout, seed=8)
    xd, wd, bd, gd, bed = [t.double().requires_grad_() for t in (x, w, b, gamma, beta)]
    rmd, rvd = rm.double().clone(), rv.double().clone()
    y = F.conv1d(xd.transpose(1, 2), wd, bd, padding=2)
    y = F.batch_norm(y, rmd, rvd, gd, bed, training=training, momentum=0.1, eps=1e-5).transpose(1, 2)
    ref = torch.tanh(y) if act == 2 else y
    if training:
        ref.backward(dz.double())
    xg, wg, bg, gg, beg = _g(x), _g(w), _g(b), _g(gamma), _g(beta)
    rmg, rvg = rm.to(_dev()), rv.to(_dev())
    nbt = torch.zeros((), dtype=torch.int64, device=_dev())
    z = ops.conv_bn(xg, wg, bg, gg, beg, rmg, rvg, nbt, training, 0.1, 1e-5, act, 0.0, 0)
    assert rel_l2(z, ref) < TOL
    if training:
        z.backward(dz.to(_dev()))
        assert int(nbt.item()) == 1
        assert rel_l2(rmg, rmd) < TOL and rel_l2(rvg, rvd) < TOL
        assert rel_l2(xg.grad, xd.grad) < TOL
        assert rel_l2(wg.grad, wd.grad) < TOL
        assert rel_l2(gg.grad, gd.grad) < TOL
        assert rel_l2(beg.grad, bed.grad) < TOL
        # conv bias in front of train-mode BN: analytically zero gradient
        assert bg.grad.abs().max().item() < 1e-3 * dz.abs().sum().item() / cout
    else:
        assert int(nbt.item()) == 0 and torch.equal(rmg.cpu(), rm)


@pytest.mark.parametrize("M,d", [(5, 128), (1000, 256), (333, 512), (7, 1024)])
def test_layernorm(M, d):
    from transformertts_amd import ops
    x, g, b, dy = _rand(M, d, seed=1) * 2 + 0.3, 1 + _rand(d, seed=2, scale=0.2), _rand(d, seed=3, scale=0.1), _rand(M, d, seed=4)
    xd, gd, bd = x.double().requires_grad_(), g.double().requires_grad_(), b.double().requires_grad_()
    ref = F.layer_norm(xd, (d,), gd, bd, 1e-5)
    ref.backward(dy.double())
    xg, gg, bg = _g(x), _g(g), _g(b)
    y = ops.layer_norm(xg, gg, bg, 1e-5)
    y.backward(dy.to(_dev()))
    assert rel_l2(y, ref) < TOL
    assert rel_l2(xg.grad, xd.grad) < TOL
    assert rel_l2(gg.grad, gd.grad) < TOL
    assert rel_l2(bg.grad, bd.grad) < TOL


def _image_decode(img, inv, M, K):
    """activation image (int16 [K/16][M][32]: 16 f16 hi, 16 f16 lo per (k-tile, row)) and per-row inverse scales -> fp64 (M, K)"""
    h = img.view(torch.float16).view(K // 16, M, 2, 16).double()
    v = (h[:, :, 0, :] + h[:, :, 1, :]).permute(1, 0, 2).reshape(M, K)
    return v * inv.double().view(M, 1)


@pytest.mark.parametrize("M,d", [(777, 256), (130, 512), (65, 1024)])
def test_layernorm_backward_with_fused_dropout_backward(M, d):
    """ttts_layernorm_bwd_drop == ttts_layernorm_bwd followed by ttts_dropout_bwd on its dx, bit for bit (dx, the dropped
    copy, the parameter gradients), and the published maxima are those of the dropped copy."""
    from transformertts_amd import _lib, ops
    from transformertts_amd.ops import _p, _stream
    lib = _lib.load()
    dev = _dev()
    x, g, dy = (_rand(M, d, seed=1) * 2 + 0.3).to(dev), (1 + _rand(d, seed=2, scale=0.2)).to(dev), _rand(M, d, seed=4).to(dev)
    b = torch.zeros(d, device=dev)
    y, mean, rstd = torch.empty_like(x), torch.empty(M, device=dev), torch.empty(M, device=dev)
    ysl = torch.zeros(1024, device=dev)
    yimg, yinv = torch.empty(M * d * 2, dtype=torch.int16, device=dev), torch.empty(M, device=dev)
    assert lib.ttts_layernorm_fwd(_p(x), _p(g), _p(b), _p(y), _p(mean), _p(rstd), M, d, 1e-5, _p(ysl), _p(yimg), _p(yinv), _stream()) == 0
    assert float(ysl.max()) == float(y.abs().max())              # the forward's own maxima of y (slot-wise atomic max)
    # the image operand the forward leaves for the GEMMs that read y: hi + lo reproduce y to 22 bits of each ROW's maximum
    dec = _image_decode(yimg, yinv, M, d)
    assert float(((dec - y.double()).abs().max(dim=1).values / y.double().abs().max(dim=1).values).max()) < 2.0 ** -21
    nb = lib.ttts_layernorm_bwd_workspace_bytes(d)
    outs = []
    for fused in (False, True):
        dx, dg, db, ws = torch.empty_like(x), torch.empty(d, device=dev), torch.empty(d, device=dev), ops._ws(nb, dev)
        dacc, am = torch.empty_like(x), torch.zeros(1024, device=dev)
        gimg, ginv = torch.empty(M * d * 2, dtype=torch.int16, device=dev), torch.empty(M, device=dev)
        if fused:
            assert lib.ttts_layernorm_bwd_drop(_p(dy), _p(x), _p(mean), _p(rstd), _p(g), _p(dx), _p(dg), _p(db), _p(ws),
                                               ws.numel() * 4, M, d, 0, _p(dacc), 0.3, 4242, None, _p(am), _p(gimg), _p(ginv), None,
                                               _stream()) == 0
        else:
            assert lib.ttts_layernorm_bwd(_p(dy), _p(x), _p(mean), _p(rstd), _p(g), _p(dx), _p(dg), _p(db), _p(ws),
                                          ws.numel() * 4, M, d, 0, _p(gimg), _p(ginv), None, _stream()) == 0
            assert lib.ttts_dropout_bwd(_p(dx), _p(dacc), dx.numel(), 0.3, 4242, None, None, _stream()) == 0
        # the image is that of the tensor the producing Linear's backward consumes: dacc behind a residual dropout, dx without
        want = (dacc if fused else dx).double()
        err = (_image_decode(gimg, ginv, M, d) - want).abs().max(dim=1).values / want.abs().max(dim=1).values.clamp_min(1e-300)
        assert float(err.max()) < 2.0 ** -21
        outs.append((dx, dg, db, dacc, am))
    for a, c in zip(outs[0][:4], outs[1][:4]):
        assert torch.equal(a, c)
    assert float(outs[1][4].max()) == float(outs[1][3].abs().max())
    frac = float((outs[1][3] == 0).float().mean())
    assert abs(frac - 0.3) < 0.02, frac


def _ref_attention(q, k, v, lens, causal):
    """q,k,v (B,H,T,head_dim) fp64; q.k^T with q pre-scaled by sqrt(1/head_dim); -inf masks; softmax; weights @ v"""
    B, H, Tq, _ = q.shape
    Tk = k.shape[2]
    s = (q * math.sqrt(1.0 / q.shape[-1])) @ k.transpose(-1, -2)
    dead = torch.arange(Tk).view(1, 1, 1, Tk) >= lens.view(B, 1, 1, 1)
    if causal:
        dead = dead | torch.triu(torch.ones(Tq, Tk, dtype=torch.bool), diagonal=1).view(1, 1, Tq, Tk)
    a = torch.softmax(s.masked_fill(dead, float("-inf")), dim=-1)
    return a @ v, a


@pytest.mark.parametrize("B,H,T,causal,lens", [(2, 2, 40, True, [40, 17]), (3, 4, 300, True, [300, 129, 1]),
                                                (2, 4, 100, False, [100, 33]), (1, 1, 129, True, [129]),
                                                (2, 2, 257, False, [257, 200])])
def test_self_attention(B, H, T, causal, lens):
    from transformertts_amd import ops
    d = H * 64
    qkv = _rand(B, T, 3 * d, seed=1)
    do = _rand(B, T, d, seed=2)
    lens_t = torch.tensor(lens, dtype=torch.int64)
    qd = qkv.double().requires_grad_()
    q, k, v = [t.view(B, T, H, 64).transpose(1, 2) for t in qd.split(d, dim=-1)]
    o, _ = _ref_attention(q, k, v, lens_t, causal)
    ref = o.transpose(1, 2).reshape(B, T, d)
    ref.backward(do.double())
    qg = _g(qkv)
    out = ops.SelfAttentionFn.apply(qg, lens_t.to(_dev()), H, causal, 0.0, 0)
    out.backward(do.to(_dev()))
    assert rel_l2(out, ref) < TOL
    assert rel_l2(qg.grad, qd.grad) < TOL


@pytest.mark.parametrize("B,H,Tq,Tk,lens", [(2, 2, 40, 12, [12, 5]), (2, 4, 300, 60, [60, 31]), (3, 4, 130, 100, [100, 64, 1]),
                                             (1, 2, 70, 161, [161])])
def test_cross_attention(B, H, Tq, Tk, lens):
    from transformertts_amd import ops
    d = H * 64
    q_, kv_, do = _rand(B, Tq, d, seed=1), _rand(B, Tk, 2 * d, seed=2), _rand(B, Tq, d, seed=3)
    lens_t = torch.tensor(lens, dtype=torch.int64)
    qd, kvd = q_.double().requires_grad_(), kv_.double().requires_grad_()
    q = qd.view(B, Tq, H, 64).transpose(1, 2)
    k, v = [t.view(B, Tk, H, 64).transpose(1, 2) for t in kvd.split(d, dim=-1)]
    o, a = _ref_attention(q, k, v, lens_t, False)
    ref = o.transpose(1, 2).reshape(B, Tq, d)
    ref.backward(do.double())
    qg, kvg = _g(q_), _g(kv_)
    out, attn = ops.CrossAttentionFn.apply(qg, kvg, lens_t.to(_dev()), H, 0.0, 0)
    out.backward(do.to(_dev()))
    assert rel_l2(out, ref) < TOL
    assert rel_l2(attn, a) < TOL
    assert torch.all(attn.sum(-1).sub(1).abs() < 1e-5)
    for b, n in enumerate(lens):          # zero mass on padded keys
        assert float(attn[b, :, :, n:].abs().sum()) == 0.0
    assert rel_l2(qg.grad, qd.grad) < TOL
    assert rel_l2(kvg.grad, kvd.grad) < TOL
    # the same attention without the weights output (single-pass online softmax path)
    q2, kv2 = _g(q_), _g(kv_)
    out2, none_w = ops.CrossAttentionFn.apply(q2, kv2, lens_t.to(_dev()), H, 0.0, 0, False)
    out2.backward(do.to(_dev()))
    assert none_w.numel() == 0
    assert rel_l2(out2, ref) < TOL and rel_l2(q2.grad, qd.grad) < TOL and rel_l2(kv2.grad, kvd.grad) < TOL


def test_attention_on_random_shapes():
    """Seeded random batch sizes, head counts, lengths (tile edges at 32 / 64 / 128 and ragged key lengths down to 1), causal
    and not, with and without the returned weights: forward, weights and every gradient against fp64."""
    import random
    from transformertts_amd import ops
    rng = random.Random(20260905)
    dev = _dev()
    for case in range(14):
        B, H = rng.choice([1, 2, 3]), rng.choice([1, 2, 4, 8])
        T = rng.choice([1, 31, 32, 33, 63, 64, 65, 127, 128, 129, 200, 333])
        causal = rng.random() < 0.5
        lens = [T] + [rng.randint(1, T) for _ in range(B - 1)]
        d = H * 64
        qkv, do = _rand(B, T, 3 * d, seed=case), _rand(B, T, d, seed=100 + case)
        lens_t = torch.tensor(lens, dtype=torch.int64)
        qd = qkv.double().requires_grad_()
        q, k, v = [t.view(B, T, H, 64).transpose(1, 2) for t in qd.split(d, dim=-1)]
        ref = _ref_attention(q, k, v, lens_t, causal)[0].transpose(1, 2).reshape(B, T, d)
        ref.backward(do.double())
        qg = _g(qkv)
        out = ops.self_attention(qg, lens_t.to(dev), H, causal, 0.0, 0)
        out.backward(do.to(dev))
        assert rel_l2(out, ref) < TOL and rel_l2(qg.grad, qd.grad) < TOL, ("self", B, H, T, causal, lens)
    for case in range(12):
        B, H = rng.choice([1, 2, 3]), rng.choice([1, 2, 4])
        Tq, Tk = rng.choice([1, 33, 64, 130, 257]), rng.choice([1, 17, 64, 100, 129, 190])
        lens = [Tk] + [rng.randint(1, Tk) for _ in range(B - 1)]
        d = H * 64
        q_, kv_, do = _rand(B, Tq, d, seed=200 + case), _rand(B, Tk, 2 * d, seed=300 + case), _rand(B, Tq, d, seed=400 + case)
        lens_t = torch.tensor(lens, dtype=torch.int64)
        qd, kvd = q_.double().requires_grad_(), kv_.double().requires_grad_()
        qq = qd.view(B, Tq, H, 64).transpose(1, 2)
        kk, vv = [t.view(B, Tk, H, 64).transpose(1, 2) for t in kvd.split(d, dim=-1)]
        o, a = _ref_attention(qq, kk, vv, lens_t, False)
        ref = o.transpose(1, 2).reshape(B, Tq, d)
        ref.backward(do.double())
        for need_w in (True, False):
            qg, kvg = _g(q_), _g(kv_)
            out, attn = ops.cross_attention(qg, kvg, lens_t.to(dev), H, 0.0, 0, need_w)
            out.backward(do.to(dev))
            assert rel_l2(out, ref) < TOL and rel_l2(kvg.grad, kvd.grad) < TOL, ("cross", B, H, Tq, Tk, lens, need_w)
            if Tk == 1:     # one key: the weights are constant 1 and dq is exactly zero in fp64 -- absolute bound instead
                assert qg.grad.abs().max().item() < 1e-6 * do.abs().max().item()
            else:
                assert rel_l2(qg.grad, qd.grad) < TOL, ("cross dq", B, H, Tq, Tk, lens, need_w)
            if need_w:
                assert rel_l2(attn, a) < TOL


@pytest.mark.parametrize("hd", [128, 192])
def test_attention_with_heads_wider_than_64(hd):
    """head_dim > 64 (the reference takes any nhead: /root/reference/model/model.py:139-161): ops.self_attention /
    cross_attention run as tensor algebra on library GEMMs (ops._attention_wide_heads) with the conventions of the kernels --
    ragged key lengths, causal mask, weights returned, an utterance without keys gives zeros -- forward, weights and every
    gradient against fp64; with dropout the weights are the dropped ones (as the kernels return them)."""
    from transformertts_amd import ops
    dev = _dev()
    for case, (B, H, Tq, Tk, causal, lens) in enumerate([(2, 2, 70, 70, True, [70, 13]), (3, 1, 33, 33, False, [33, 1, 20]),
                                                         (2, 2, 50, 19, False, [19, 7]), (2, 1, 5, 9, False, [9, 0])]):
        d = H * hd
        lens_t = torch.tensor(lens, dtype=torch.int64)
        if Tq == Tk:
            qkv, do = _rand(B, Tq, 3 * d, seed=case), _rand(B, Tq, d, seed=50 + case)
            qd = qkv.double().requires_grad_()
            q, k, v = [t.view(B, Tq, H, hd).transpose(1, 2) for t in qd.split(d, dim=-1)]
            ref = _ref_attention(q, k, v, lens_t, causal)[0].transpose(1, 2).reshape(B, Tq, d)
            ref.backward(do.double())
            qg = _g(qkv)
            out = ops.self_attention(qg, lens_t.to(dev), H, causal, 0.0, 0)
            out.backward(do.to(dev))
            assert rel_l2(out, ref) < TOL and rel_l2(qg.grad, qd.grad) < TOL, ("self", hd, B, H, Tq, causal, lens)
        else:
            q_, kv_, do = _rand(B, Tq, d, seed=60 + case), _rand(B, Tk, 2 * d, seed=70 + case), _rand(B, Tq, d, seed=80 + case)
            qd, kvd = q_.double().requires_grad_(), kv_.double().requires_grad_()
            qq = qd.view(B, Tq, H, hd).transpose(1, 2)
            kk, vv = [t.view(B, Tk, H, hd).transpose(1, 2) for t in kvd.split(d, dim=-1)]
            o, a = _ref_attention(qq, kk, vv, lens_t, False)
            ref = o.transpose(1, 2).reshape(B, Tq, d)
            if min(lens) == 0:                       # an utterance without keys: zeros (the reference softmax has no answer there)
                live = (lens_t > 0).double()
                ref, a = torch.nan_to_num(ref) * live[:, None, None], torch.nan_to_num(a) * live[:, None, None, None]
            ref.backward(do.double())
            qg, kvg = _g(q_), _g(kv_)
            out, attn = ops.cross_attention(qg, kvg, lens_t.to(dev), H, 0.0, 0, True)
            out.backward(do.to(dev))
            assert torch.isfinite(out).all() and torch.isfinite(qg.grad).all() and torch.isfinite(kvg.grad).all()
            assert rel_l2(out, ref) < TOL and rel_l2(attn, a) < TOL, ("cross", hd, B, H, Tq, Tk, lens)
            if min(lens) > 0:
                assert rel_l2(qg.grad, qd.grad) < TOL and rel_l2(kvg.grad, kvd.grad) < TOL
    # dropout: the returned weights are the dropped ones, rescaled; about p of them are zero
    q_, kv_ = _rand(2, 40, 2 * hd, seed=91), _rand(2, 30, 4 * hd, seed=92)
    out, attn = ops.cross_attention(q_.to(dev), kv_.to(dev), torch.tensor([30, 30], device=dev), 2, 0.25, 7, True)
    zero = float((attn == 0).float().mean())
    assert 0.2 < zero < 0.3 and abs(float(attn.sum(-1).mean()) - 1.0) < 0.05


def test_fanout_sums_consumer_gradients_in_one_launch():
    """ops.fanout: n aliases whose gradients meet in one ttts_add / ttts_add3 launch -- autograd's own accumulation up to
    the association order of a three-term fp32 sum --, also with an unused handle and without gradient tracking."""
    from transformertts_amd import ops
    x = _rand(7, 33, 64, seed=1)
    w = [_rand(7, 33, 64, seed=10 + i) for i in range(3)]
    for n in (2, 3):
        ref = _g(x)
        sum((ref * w[i].to(_dev())).sum() * (i + 1.0) for i in range(n)).backward()
        xg = _g(x)
        hs = ops.fanout(xg, n)
        assert len(hs) == n and all(h.data_ptr() == xg.data_ptr() for h in hs)
        sum((hs[i] * w[i].to(_dev())).sum() * (i + 1.0) for i in range(n)).backward()
        assert rel_l2(xg.grad, ref.grad) < 1e-7 and (xg.grad - ref.grad).abs().max().item() <= 4e-7 * ref.grad.abs().max().item()
    xg = _g(x)
    a, b, c = ops.fanout(xg, 3)
    ((a * 2.0).sum() + (c * 3.0).sum()).backward()              # the middle handle has no consumer
    assert torch.equal(xg.grad, torch.full_like(xg, 5.0))
    with torch.no_grad():
        hs = ops.fanout(_g(x), 3)
    assert all(not h.requires_grad or h.grad_fn is None for h in hs)


def test_embedding_posenc_heads_add():
    from transformertts_amd import ops
    dev = _dev()
    V, d, B, T = 30, 128, 3, 21
    ids = torch.randint(0, V, (B, T), generator=torch.Generator().manual_seed(1))
    tab, dout = _rand(V, d, seed=2), _rand(B, T, d, seed=3)
    td = tab.double().requires_grad_()
    F.embedding(ids, td).backward(dout.double())
    tg = _g(tab)
    e = ops.EmbeddingFn.apply(ids.to(dev), tg)
    e.backward(dout.to(dev))
    assert torch.equal(e.cpu(), F.embedding(ids, tab))
    assert rel_l2(tg.grad, td.grad) < TOL

    from oracle.spec import sinusoid_table
    pe = sinusoid_table(100, d)
    x, alpha = _rand(B, T, d, seed=4), torch.tensor([1.3])
    xd, ad = x.double().requires_grad_(), alpha.double().requires_grad_()
    ref = xd + ad * pe[:T].double().unsqueeze(0)
    ref.backward(dout.double())
    xg, ag = _g(x), _g(alpha)
    y = ops.PosEncFn.apply(xg, pe.to(dev), ag, 0.0, 0)
    y.backward(dout.to(dev))
    assert rel_l2(y, ref) < TOL and rel_l2(xg.grad, xd.grad) < TOL and rel_l2(ag.grad, ad.grad) < TOL

    K, N, M = 128, 16, B * T
    wm, bm, ws, bs = _rand(N, K, seed=5, scale=K ** -0.5), _rand(N, seed=6), _rand(1, K, seed=7, scale=K ** -0.5), _rand(1, seed=8)
    dmel, dstop = _rand(B, T, N, seed=9), _rand(B, T, seed=10)
    ps = [t.double().requires_grad_() for t in (x, wm, bm, ws, bs)]
    mel_r, stop_r = F.linear(ps[0], ps[1], ps[2]), F.linear(ps[0], ps[3], ps[4]).squeeze(-1)
    (mel_r * dmel.double()).sum().add((stop_r * dstop.double()).sum()).backward()
    gs = [_g(t) for t in (x, wm, bm, ws, bs)]
    mel, stop = ops.HeadsFn.apply(*gs)
    ((mel * dmel.to(dev)).sum() + (stop * dstop.to(dev)).sum()).backward()
    assert rel_l2(mel, mel_r) < TOL and rel_l2(stop, stop_r) < TOL
    for a, b in zip(gs, ps):
        assert rel_l2(a.grad, b.grad) < TOL

    z = ops.AddFn.apply(x.to(dev), dout.to(dev))
    assert torch.equal(z.cpu(), x + dout)


def test_dropout_masks_are_consistent_and_calibrated():
    """Dropout cannot match torch's CPU Philox stream; instead: (i) kept fraction ~ 1-p, kept values scaled by
    1/(1-p); (ii) backward regenerates exactly the forward mask (gradient zero where the output was dropped)."""
    from transformertts_amd import ops
    dev = _dev()
    M, K, N, p = 2048, 256, 256, 0.5
    x, w = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=K ** -0.5)
    r = _rand(M, N, seed=3)
    xg, wg = _g(x), _g(w)
    y0 = ops.linear(xg, wg, None, residual=r.to(dev), drop_p=0.0)
    y1 = ops.linear(xg, wg, None, residual=r.to(dev), drop_p=p, seed=1234)
    acc0 = (y0 - r.to(dev)).detach()
    acc1 = (y1 - r.to(dev)).detach()
    kept = acc1.abs() > 1e-6 * acc0.abs().mean()   # dropped entries come back as residual exactly
    frac = kept.float().mean().item()
    assert abs(frac - (1 - p)) < 0.01
    assert rel_l2(acc1[kept], acc0[kept] / (1 - p)) < 1e-5
    dy = torch.ones_like(y1)
    (gx,) = torch.autograd.grad(y1, xg, dy)
    ref_gx = (kept.float() / (1 - p)) @ w.to(dev)
    assert rel_l2(gx, ref_gx) < 1e-5
    y2 = ops.linear(xg, wg, None, residual=r.to(dev), drop_p=p, seed=1235)
    assert (((y2 - r.to(dev)).abs() > 1e-9) != kept).float().mean().item() > 0.3   # new seed, new mask

    # attention-weight dropout: rows of the returned (post-dropout) weights average to ~1, zeros ~ p
    B, H, Tq, Tk = 2, 2, 200, 64
    q_, kv_ = _rand(B, Tq, 128, seed=4), _rand(B, Tk, 256, seed=5)
    lens = torch.full((B,), Tk, dtype=torch.int64, device=dev)
    qg, kvg = _g(q_), _g(kv_)
    _, a0 = ops.CrossAttentionFn.apply(qg, kvg, lens, H, 0.0, 0)
    o1, a1 = ops.CrossAttentionFn.apply(qg, kvg, lens, H, 0.1, 77)
    z = (a1 == 0).float().mean().item()
    assert abs(z - 0.1) < 0.01
    keep = a1 != 0
    assert rel_l2(a1[keep], a0[keep] / 0.9) < 1e-5
    # backward uses the same mask: compare with an explicit computation from the returned weights
    do = _rand(B, Tq, 128, seed=6).to(dev)
    (gkv,) = torch.autograd.grad(o1, kvg, do)
    v_grad_ref = torch.einsum("bhqk,bqhd->bkhd", a1, do.view(B, Tq, H, 64)).reshape(B, Tk, 128)
    assert rel_l2(gkv[:, :, 128:], v_grad_ref) < 1e-5


@pytest.mark.parametrize("B,T,C,lens", [(3, 37, 16, [37, 20, 9]), (4, 300, 80, [300, 211, 95, 1])])
def test_loss_kernels(B, T, C, lens):
    from oracle import oracle_loss
    from transformertts_amd.loss import TransformerTTSLoss
    pred, post, stop, mel = _rand(B, T, C, seed=1), _rand(B, T, C, seed=2), _rand(B, T, seed=3, scale=2.0), _rand(B, T, C, seed=4)
    lens_t = torch.tensor(lens, dtype=torch.int64)
    pd, qd, sd = [t.double().requires_grad_() for t in (pred, post, stop)]
    ref = oracle_loss({"pred_melspec": pd, "post_melspec": qd, "pred_stop": sd}, mel.double(), lens_t)
    w = torch.tensor([1.0, 0.3, -0.7, 2.0], dtype=torch.float64)       # exercise all four upstream gradients
    (w[0] * ref["total"] + w[1] * ref["pred_mel"] + w[2] * ref["post_mel"] + w[3] * ref["stop"]).backward()
    pg, qg, sg = _g(pred), _g(post), _g(stop)
    out = TransformerTTSLoss(8.0).to(_dev())({"pred_melspec": pg, "post_melspec": qg, "pred_stop": sg}, mel.to(_dev()),
                                             lens_t.to(_dev()))
    for k in ("total", "pred_mel", "post_mel", "stop"):
        assert abs(out[k].item() - ref[k].item()) < 2e-6 * max(1.0, abs(ref[k].item())), k
    wd = w.float().to(_dev())
    (wd[0] * out["total"] + wd[1] * out["pred_mel"] + wd[2] * out["post_mel"] + wd[3] * out["stop"]).backward()
    assert rel_l2(pg.grad, pd.grad) < TOL and rel_l2(qg.grad, qd.grad) < TOL and rel_l2(sg.grad, sd.grad) < TOL
    for b, n in enumerate(lens):           # padded frames get exactly zero gradient
        assert float(pg.grad[b, n:].abs().sum()) == 0.0 and float(sg.grad[b, n:].abs().sum()) == 0.0


def test_scheduled_sampling_mix_kernel(golden_dir):
    """Bit-exact against the reference's own mixed tensors (tests/golden/helpers.npz) for the injected uniform draw."""
    import os
    import numpy as np
    from transformertts_amd import ops
    g = np.load(os.path.join(golden_dir, "helpers.npz"))
    pred, mel, lens = (torch.from_numpy(g["ss/pred"]), torch.from_numpy(g["ss/mel"]), torch.from_numpy(g["ss/lens"]))
    dev = _dev()
    for p_tf in (1.0, 0.7, 0.05):
        u = torch.from_numpy(g[f"ss/u_{p_tf}"]).view(pred.shape[0], -1)
        out = ops.sched_sampling_mix(pred.to(dev), mel.to(dev), u.to(dev), lens.to(dev), p_tf, 8)
        assert torch.equal(out.cpu(), torch.from_numpy(g[f"ss/mixed_{p_tf}"])), p_tf


def test_block_mask_shim_matches_reference_windowing():
    """`utils.util.block_mask` (reference utils/util.py:103-111) for an injected draw: the (B,T,1) bool mask equals the
    max-pool dilation of (u < 1 - p_tf), window L_bar with padding L_bar // 2, cut to T frames."""
    import torch.nn.functional as F
    import transformertts_amd.utils.util as U
    g = torch.Generator().manual_seed(5)
    B, T = 3, 53
    u = torch.rand(B, 1, T, generator=g)
    mel = torch.zeros(B, T, 16, device=_dev())
    U._uniform_draw = lambda b, t, device: u.to(device)
    try:
        for p_tf, l_bar in ((0.9, 8), (0.5, 8), (0.8, 4)):
            got = U.block_mask(mel, p_tf, l_bar)
            want = F.max_pool1d((u < (1 - p_tf)).float(), kernel_size=l_bar, stride=1, padding=l_bar // 2)
            want = want.squeeze(1).bool().unsqueeze(-1)[:, :T, :]
            assert got.dtype == torch.bool and got.shape == (B, T, 1) and torch.equal(got.cpu(), want), (p_tf, l_bar)
    finally:
        U._uniform_draw = None
    assert not bool(U.block_mask(mel, 1.0, 8).any())         # p_tf = 1: pure teacher forcing, in-kernel draw


def test_loss_kernel_matches_reference_fixture(golden_dir):
    """The fused loss against the values the reference's own TransformerTTSLoss produced (tests/golden/helpers.npz)."""
    import os
    import numpy as np
    from transformertts_amd.loss import TransformerTTSLoss
    g = np.load(os.path.join(golden_dir, "helpers.npz"))
    dev = _dev()
    outs = {"pred_melspec": torch.from_numpy(g["ss/pred"]).to(dev), "post_melspec": torch.from_numpy(g["loss/post"]).to(dev),
            "pred_stop": torch.from_numpy(g["loss/stop_logits"]).to(dev)}
    ls = TransformerTTSLoss(8.0).to(dev)(outs, torch.from_numpy(g["ss/mel"]).to(dev), torch.from_numpy(g["ss/lens"]).to(dev))
    for k in ("total", "pred_mel", "post_mel", "stop"):
        assert abs(ls[k].item() - float(g[f"loss/{k}"])) < 2e-6 * max(1.0, abs(float(g[f"loss/{k}"]))), k


def test_in_kernel_uniform_draw_and_step_state():
    """u = None: the mix draws its own uniforms.  The replaced fraction follows 1 - (1 - q)^8-ish block statistics of
    the reference's max-pool dilation, the draw is a pure function of (seed, step-state seed word), and the step state
    overrides p_tf from device memory (what a captured graph relies on)."""
    from transformertts_amd import ops
    dev = _dev()
    B, T, C = 8, 4000, 16
    pred = torch.ones(B, T, C, device=dev)
    mel = torch.zeros(B, T, C, device=dev)
    lens = torch.full((B,), T, dtype=torch.int64, device=dev)
    for p_tf in (1.0, 0.9, 0.5):
        out = ops.sched_sampling_mix(pred, mel, None, lens, p_tf, 8, seed=123)
        frac = out[..., 0].mean().item()
        want = 1.0 - p_tf ** 8                      # a frame keeps the truth only if all 8 draws of its window are >= 1-p_tf
        assert abs(frac - want) < 0.02, (p_tf, frac, want)
        assert torch.equal(out, ops.sched_sampling_mix(pred, mel, None, lens, p_tf, 8, seed=123))
        if p_tf < 1.0:
            assert not torch.equal(out, ops.sched_sampling_mix(pred, mel, None, lens, p_tf, 8, seed=124))
    st = ops.StepState(dev)
    st.push(seed=0xABCDEF0123456789, lr=0.0, p_tf=0.5, step=1)
    with st:
        a = ops.sched_sampling_mix(pred, mel, None, lens, 1.0, 8, seed=123)       # by-value p_tf = 1.0 is ignored
    assert abs(a[..., 0].mean().item() - (1.0 - 0.5 ** 8)) < 0.02
    st.push(seed=0x1111, lr=0.0, p_tf=0.5, step=2)
    with st:
        b = ops.sched_sampling_mix(pred, mel, None, lens, 1.0, 8, seed=123)
    assert not torch.equal(a, b)                                                  # new seed word, new draw
    # dropout sites follow the seed word too, and backward regenerates the forward mask under the same word
    x = torch.randn(64, 256, device=dev, requires_grad=True)
    w = torch.randn(256, 256, device=dev)
    with st:
        y1 = ops.linear(x, w, None, drop_p=0.5, seed=7)
        (g1,) = torch.autograd.grad(y1.sum(), x)
    st.push(seed=0x2222, lr=0.0, p_tf=0.5, step=3)
    with st:
        y2 = ops.linear(x, w, None, drop_p=0.5, seed=7)
    assert not torch.equal(y1 == 0, y2 == 0)
    keep = (y1 != 0).float() * 2.0
    assert rel_l2(g1, keep @ w) < 1e-5


def test_flat_adam_matches_torch_adam():
    """FlatAdam (fused clip + Adam over flat buffers) against torch.optim.Adam + clip_grad_norm_ in fp64 on CPU."""
    from transformertts_amd.optim import FlatAdam
    from transformertts_amd.utils.util import get_noam_scheduler
    dev = _dev()
    shapes = [(100, 256), (768,), (256, 80, 5), (1,), (33, 7)]
    ps = [torch.nn.Parameter(_rand(*s, seed=10 + i).to(dev)) for i, s in enumerate(shapes)]
    ref = [torch.nn.Parameter(p.detach().cpu().double()) for p in ps]
    opt = FlatAdam(ps, lr=1.0, betas=(0.9, 0.98), eps=1e-9, max_grad_norm=1.0)
    ropt = torch.optim.Adam(ref, lr=1.0, betas=(0.9, 0.98), eps=1e-9)
    lam = get_noam_scheduler(256, 4000)
    sch = torch.optim.lr_scheduler.LambdaLR(opt, lam)
    rsch = torch.optim.lr_scheduler.LambdaLR(ropt, lam)
    for step in range(4):
        opt.zero_grad()
        for i, (p, r) in enumerate(zip(ps, ref)):
            g = _rand(*p.shape, seed=100 * step + i, scale=(3.0 if step % 2 else 0.01))   # clipped and unclipped steps
            p.grad.copy_(g.to(dev))
            r.grad = g.double()
        torch.nn.utils.clip_grad_norm_(ref, 1.0)
        opt.step(); ropt.step(); sch.step(); rsch.step()
    for p, r in zip(ps, ref):
        assert p.data_ptr() >= opt.flat_params.data_ptr()       # still a view of the flat buffer
        assert rel_l2(p, r) < 1e-5
