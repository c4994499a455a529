"""HEAD-IMAGE operands through the C ABI: the in-projection GEMM that writes q / k / v as f16 hi / lo pieces with a power-of-two
scale per (row, 64-column head) (ttts_linear_fwd_h3d_img, csrc/gemm_h3i.hip), the stand-alone conversion (ttts_head_image) and
the attention kernels that stage such operands by LDS-DMA (ttts_attention_fwd_img / ttts_attention_bwd_img,
csrc/attention_img.hip) -- each against fp64 torch on the same inputs (torch F.multi_head_attention_forward /
scaled_dot_product_attention semantics: torch/nn/functional.py:6206-6629, reached from model/layers.py:54-74), with the
fp32-operand fp16x3 kernels beside them where the two must agree exactly (dropout masks)."""
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


def guarded(shape, fill, guard_rows=1):
    from conftest import guarded as g
    return g(shape, fill, guard_rows)


def _decode(img, inv):
    """head image (M, N 4-byte cells) + inverse scales (N / 64, M) -> float64 (M, N)"""
    M, N = img.shape
    h = img.view(torch.float16).view(M, N // 64, 2, 64).double()
    return ((h[:, :, 0] + h[:, :, 1]) * inv.t().double()[:, :, None]).reshape(M, N)


def _himg(x2d):
    from transformertts_amd import _lib
    from transformertts_amd.ops import _p, _stream
    M, N = x2d.shape
    img = torch.full((M, N), float("nan"), device=_dev())
    inv = torch.full((N // 64, M), float("nan"), device=_dev())
    _lib.check(_lib.load().ttts_head_image(_p(x2d), N, _p(img), N, _p(inv), M, N, _stream()), "ttts_head_image")
    return img, inv


@pytest.mark.parametrize("M,N,K", [(300, 768, 256), (1000, 256, 256), (777, 512, 512), (6400, 1536, 512), (70001, 768, 256),
                                   (129, 192, 96)])
def test_in_projection_writes_head_images(M, N, K):
    """ttts_linear_fwd_h3d_img: x W^T + b decoded from the image against fp64 -- rows and heads of very different magnitude (x 30,
    x 0.03, a head 1000 x smaller, all zero) as accurate as their neighbours, which is what the per-(row, head) scale is for; the inverse scales are powers of
    two that put every head row's maximum in [2^11, 2^12); the section maxima are exact; a second run gives the same bits;
    (70 001, 768, 256) runs several tiles per workgroup with a ragged last row block, (129, 192, 96) a ragged column block."""
    from transformertts_amd import _lib, ops
    from transformertts_amd.ops import _p, _stream
    lib, dev = _lib.load(), _dev()
    x, w, b = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=K ** -0.5), _rand(N, seed=3)
    # (the INPUT keeps its per-tensor scale -- ttts_linear_fwd_h3d's operand form: rows within 2^-15 of max|x| keep 22 bits --
    # so the row factors stay inside that window; what is under test is the OUTPUT side)
    x[3] *= 30.0
    x[7] *= 0.03
    x[5] = 0.0
    b[64:128] = 0.0                                     # with x[5] = 0: one head row that is exactly zero
    w[128:192] *= 1e-3                                  # a head whose outputs are 1000 x smaller than its neighbours'
    b[128:192] *= 1e-3
    nsec = 3 if N % 192 == 0 else 1
    pl = ops._planes(w, 8, N, K)
    xa = ops._amax(x)

    def run():
        # guard rows behind every output: the section maxima (one more section of slots: where the publish of a head past the
        # last column -- (129, 192, 96): hcol = 192 of a 256-column tile, section 3 of 3 -- landed before round 5's fix), the
        # inverse scales (one more head plane) and the image (one more row block)
        img, chk_img = guarded((M, N), float("nan"), guard_rows=128)
        inv, chk_inv = guarded((N // 64, M), float("nan"))
        am, chk_am = guarded((nsec, ops.AMAX_SLOTS), 0.0)
        _lib.check(lib.ttts_linear_fwd_h3d_img(_p(x), _p(pl), _p(b), _p(img), _p(inv), M, N, K, _p(xa), _p(am), N // nsec if nsec > 1 else 0,
                                               _stream()), "fwd_h3d_img")
        torch.cuda.synchronize()
        chk_img(); chk_inv(); chk_am()
        return img, inv, am
    img, inv, am = run()
    ref = x.double() @ w.double().t() + b.double()
    y = _decode(img, inv)
    assert torch.isfinite(y).all()
    assert _rel(y, ref) < 1e-6, _rel(y, ref)
    heads = (y - ref).view(M, N // 64, 64).norm(dim=2) / ref.view(M, N // 64, 64).norm(dim=2).clamp_min(1e-300)
    assert float(heads.max()) < 2e-6, float(heads.max())                       # every (row, head) on its own, the tiny rows included
    assert float(y[5, 64:128].abs().max()) == 0.0 and float(inv[1, 5]) == 1.0    # the zero head row: zeros, scale 1
    # scales: exact powers of two, maximum of each head row of the stored values in [2^11, 2^12)
    man, _ = torch.frexp(inv)
    assert bool((man == 0.5).all())
    stored = img.view(torch.float16).view(M, N // 64, 2, 64)[:, :, 0].float().abs().amax(dim=2)       # hi plane
    nz = ref.view(M, N // 64, 64).abs().amax(dim=2) > 1e-30
    assert bool(((stored >= 2048.0) & (stored <= 4096.0))[nz].all())
    for s in range(nsec):
        sec = ref[:, s * (N // nsec):(s + 1) * (N // nsec)]
        got = float(am[s].max())
        assert abs(got - float(sec.abs().max())) <= 2e-6 * got
    torch.randn(1 << 23, device=dev)
    img2, inv2, am2 = run()
    assert torch.equal(img.view(torch.int32), img2.view(torch.int32)) and torch.equal(inv, inv2) and torch.equal(am, am2)
    # the stand-alone conversion of the fp32 result agrees with its own decode, and with the epilogue's image wherever the fp32
    # rounding of y did not move a value across a binade
    y32 = ref.float().contiguous()
    img3, inv3 = _himg(y32)
    assert _rel(_decode(img3, inv3), y32.double()) < 3e-7


def _reference(q, kv, do, lens, causal, H, keep=None, p_drop=0.0):
    B, Tq, d = q.shape
    Tk = kv.shape[1]
    kl = torch.tensor(lens, dtype=torch.int64, device=_dev())
    qd = q.double().view(B, Tq, H, 64).transpose(1, 2).requires_grad_()
    kd = kv[..., :d].double().reshape(B, Tk, H, 64).transpose(1, 2).requires_grad_()
    vd = kv[..., d:].double().reshape(B, Tk, H, 64).transpose(1, 2).requires_grad_()
    mask = torch.arange(Tk, device=_dev())[None, None, None, :] >= kl[:, None, None, None]
    if causal:
        mask = mask | (torch.arange(Tk, device=_dev())[None, :] > torch.arange(Tq, device=_dev())[:, None])
    p = torch.softmax((qd @ kd.transpose(-1, -2) / 8.0).masked_fill(mask, float("-inf")), -1)
    if keep is not None:
        p = p * (keep | mask.expand_as(p)) / (1.0 - p_drop)
    o = (p @ vd).transpose(1, 2).reshape(B, Tq, d)
    o.backward(do.double())
    dq = qd.grad.transpose(1, 2).reshape(B, Tq, d)
    dkv = torch.cat([kd.grad.transpose(1, 2).reshape(B, Tk, d), vd.grad.transpose(1, 2).reshape(B, Tk, d)], -1)
    return o.detach(), p.detach(), dq, dkv, mask


def _run_img(q, kv, do, lens, causal, H, p_drop=0.0, seed=0, want_attn=True):
    """forward + backward on head images of q (B,Tq,d) and packed kv (B,Tk,2d)"""
    from transformertts_amd import _lib, ops
    from transformertts_amd.ops import _p, _off, _stream
    lib, dev = _lib.load(), _dev()
    B, Tq, d = q.shape
    Tk = kv.shape[1]
    kl = torch.tensor(lens, dtype=torch.int64, device=dev)
    qi, qinv = _himg(q.reshape(B * Tq, d).contiguous())
    kvi, kvinv = _himg(kv.reshape(B * Tk, 2 * d).contiguous())
    va = ops._amax(kv[..., d:].contiguous())
    checks = []

    def G(shape, fill):       # every output with a sentinel block behind it (conftest.guarded)
        t, chk = guarded(shape, fill)
        checks.append(chk)
        return t
    o = G((B, Tq, d), float("nan"))
    stat = G((6, B, H, Tq), float("nan"))
    attn = G((B, H, Tq, Tk), float("nan")) if (want_attn and not causal) else None
    oslots = G((1, ops.AMAX_SLOTS), 0.0)[0]
    HK = H * B * Tk
    _lib.check(lib.ttts_attention_fwd_img(_p(qi), _off(kvi, 0), _off(kvi, d), _p(qinv), _off(kvinv, 0), _off(kvinv, HK), _p(o), _p(stat[0]),
                                          _p(attn), _p(kl), B, H, Tq, Tk, d, 2 * d, 2 * d, d, causal, 0.125, p_drop, seed, None, _p(va),
                                          _p(oslots), _p(stat[1:]), 0, 0, 0, _stream()), "fwd_img")
    dq, dkv = G(tuple(q.shape), float("nan")), G(tuple(kv.shape), float("nan"))
    delta = G((B, H, Tq), 0.0)
    sq, sk = G((1, ops.AMAX_SLOTS), 0.0)[0], G((1, ops.AMAX_SLOTS), 0.0)[0]
    _lib.check(lib.ttts_attention_bwd_img(_p(qi), _off(kvi, 0), _off(kvi, d), _p(qinv), _off(kvinv, 0), _off(kvinv, HK), _p(o), _p(do),
                                          _p(stat[1:]), _p(delta), _p(dq), _off(dkv, 0), _off(dkv, d), _p(kl), B, H, Tq, Tk, d, 2 * d, 2 * d,
                                          d, d, 2 * d, 2 * d, causal, 0.125, p_drop, seed, None, _p(ops._amax(do)), _p(sq), _p(sk), None, 1,
                                          0, 0, 0, _stream()), "bwd_img")
    out = {"o": o, "attn": attn, "lse": stat[0], "dq": dq, "dkv": dkv, "o_amax": oslots, "dq_amax": sq, "dkv_amax": sk}
    if Tq >= 64:
        # the same backward with the query range of the dK / dV kernel split over 3 workgroups per key block: partial sums + one
        # fixed-order reduction -- equal to fp32 rounding of the sum order, maxima published by the reduction
        dq2, dkv2 = G(tuple(q.shape), float("nan")), G(tuple(kv.shape), float("nan"))
        part = G((3, B, Tk, 2 * d), float("nan"))
        sk2 = G((1, ops.AMAX_SLOTS), 0.0)[0]
        _lib.check(lib.ttts_attention_bwd_img(_p(qi), _off(kvi, 0), _off(kvi, d), _p(qinv), _off(kvinv, 0), _off(kvinv, HK), _p(o), _p(do),
                                              _p(stat[1:]), _p(delta), _p(dq2), _off(dkv2, 0), _off(dkv2, d), _p(kl), B, H, Tq, Tk, d, 2 * d,
                                              2 * d, d, d, 2 * d, 2 * d, causal, 0.125, p_drop, seed, None, _p(ops._amax(do)), None, _p(sk2),
                                              _p(part), 3, 0, 0, 0, _stream()), "bwd_img split")
        out["dkv_split"], out["dkv_split_amax"] = dkv2, sk2
        assert torch.equal(dq2, dq)
    torch.cuda.synchronize()
    for chk in checks:
        chk()
    return out


CASES = [(1, 200, 200, [200, 131, 64]), (0, 150, 70, [70, 33, 1]), (0, 33, 129, [129, 128, 5]), (1, 870, 870, [870, 500]),
         (0, 870, 100, [100, 61]), (1, 1, 1, [1]), (0, 64, 64, [64, 0])]


@pytest.mark.parametrize("causal,Tq,Tk,lens", CASES)
@pytest.mark.parametrize("qk_scale", [1.0, 6.0])
def test_attention_on_head_images(causal, Tq, Tk, lens, qk_scale):
    """forward (context, per-head weights, lse) and backward (dq, dk, dv) against fp64, ragged lengths incl. a single key and an
    empty utterance, peaked softmaxes (scores of +-100 at qk_scale 6), published maxima exact; with dropout on, the masks are the
    ones the fp32-operand fp16x3 kernels draw for the same seed, and the backward regenerates them."""
    from transformertts_amd import _lib, ops
    from transformertts_amd.ops import _p, _off, _stream
    lib = _lib.load()
    B, H, d = len(lens), 2, 128
    q, kv, do = _rand(B, Tq, d, seed=1) * qk_scale, _rand(B, Tk, 2 * d, seed=2), _rand(B, Tq, d, seed=3)
    kv[..., :d] *= qk_scale
    o_ref, p_ref, dq_ref, dkv_ref, mask = _reference(q, kv, do, lens, causal, H)
    live = torch.tensor([l > 0 for l in lens], device=_dev())        # (an utterance without keys: torch gives NaN rows, we give zeros)
    r = _run_img(q, kv, do, lens, causal, H)
    for k in ("o", "dq", "dkv"):
        assert torch.isfinite(r[k]).all(), k
    assert _rel(r["o"][live], o_ref[live]) < TOL, _rel(r["o"][live], o_ref[live])
    assert float(r["o"][~live].abs().max() if (~live).any() else 0.0) == 0.0
    assert r["o_amax"].max().item() == r["o"].abs().max().item()
    if r["attn"] is not None:
        assert _rel(r["attn"][live], p_ref[live]) < TOL
        assert float(r["attn"][live].sum(-1).sub(1).abs().max()) < 1e-5
        assert float(r["attn"].masked_select(mask.expand_as(r["attn"])).abs().max() if mask.any() else 0.0) == 0.0
    assert _rel(r["dq"][live], dq_ref[live]) < TOL, _rel(r["dq"][live], dq_ref[live])
    assert _rel(r["dkv"][live][..., :d], dkv_ref[live][..., :d]) < TOL and _rel(r["dkv"][live][..., d:], dkv_ref[live][..., d:]) < TOL, \
        (_rel(r["dkv"][live][..., :d], dkv_ref[live][..., :d]), _rel(r["dkv"][live][..., d:], dkv_ref[live][..., d:]))
    assert r["dq_amax"].max().item() == r["dq"].abs().max().item() and r["dkv_amax"].max().item() == r["dkv"].abs().max().item()
    if "dkv_split" in r:
        assert torch.isfinite(r["dkv_split"]).all()
        assert _rel(r["dkv_split"][live], dkv_ref[live]) < TOL and _rel(r["dkv_split"], r["dkv"]) < 1e-6
        assert r["dkv_split_amax"].max().item() == r["dkv_split"].abs().max().item()
    if min(lens) == 0 or Tq == 1:
        return
    # dropout: the kept weights ARE the fp32-operand kernel's for the same seed, and the fp64 reference that drops exactly the
    # weights the returned maps show as dropped reproduces the context and all three gradients
    p_drop = 0.25
    rd = _run_img(q, kv, do, lens, causal, H, p_drop=p_drop, seed=99)
    qa, kva = ops._amax(q), ops._amax(kv)
    o3, lse3 = torch.empty_like(q), torch.empty(B, H, Tq, device=_dev())
    a3 = None if causal else torch.empty(B, H, Tq, Tk, device=_dev())
    assert lib.ttts_attention_fwd_h3(_p(q), _off(kv, 0), _off(kv, d), _p(o3), _p(lse3), _p(a3), _p(torch.tensor(lens, device=_dev())), B, H,
                                     Tq, Tk, d, 2 * d, 2 * d, d, causal, 0.125, p_drop, 99, None, _p(qa), _p(kva), _p(kva), None, None,
                                     _stream()) == 0
    assert _rel(rd["o"], o3) < TOL and _rel(rd["lse"], lse3) < TOL
    if a3 is not None:
        if qk_scale == 1.0:
            assert torch.equal(rd["attn"] == 0, a3 == 0)
        keep = rd["attn"] != 0
        od, _, dq2, dkv2, _ = _reference(q, kv, do, lens, causal, H, keep=keep, p_drop=p_drop)
        assert _rel(rd["o"], od) < TOL and _rel(rd["dq"], dq2) < TOL and _rel(rd["dkv"], dkv2) < TOL, \
            (_rel(rd["o"], od), _rel(rd["dq"], dq2), _rel(rd["dkv"], dkv2))
    else:
        dq3, dkv3, delta3 = torch.empty_like(q), torch.empty_like(kv), torch.empty_like(lse3)
        rs3 = torch.empty(3, B, H, Tq, device=_dev())
        assert lib.ttts_attention_fwd_h3(_p(q), _off(kv, 0), _off(kv, d), _p(o3), _p(lse3), None, _p(torch.tensor(lens, device=_dev())), B, H,
                                         Tq, Tk, d, 2 * d, 2 * d, d, causal, 0.125, p_drop, 99, None, _p(qa), _p(kva), _p(kva), None, _p(rs3),
                                         _stream()) == 0
        assert lib.ttts_attention_bwd_h3(_p(q), _off(kv, 0), _off(kv, d), _p(o3), _p(do), _p(lse3), _p(delta3), _p(dq3), _off(dkv3, 0),
                                         _off(dkv3, d), _p(torch.tensor(lens, device=_dev())), B, H, Tq, Tk, d, 2 * d, 2 * d, d, d, 2 * d,
                                         2 * d, causal, 0.125, p_drop, 99, None, _p(ops._amax(do)), None, None, _p(qa), _p(kva), _p(kva),
                                         _p(rs3), _stream()) == 0
        assert _rel(rd["dq"], dq3) < TOL and _rel(rd["dkv"], dkv3) < TOL, (_rel(rd["dq"], dq3), _rel(rd["dkv"], dkv3))


@pytest.mark.parametrize("causal,Tq,Tk,lens", [(1, 200, 200, [200, 131, 64]), (0, 150, 70, [70, 33, 1]), (1, 870, 870, [870, 500])])
@pytest.mark.parametrize("mag,grow", [(3e-7, False), (3e-7, True)])
def test_attention_backward_on_head_images_small_and_growing_gradients(causal, Tq, Tk, lens, mag, grow):
    """gradients of realistic magnitude (3e-7), and inputs that FORCE the rare branch of the lane-local dS scale tracking (the
    last keys' V rows and the last queries' dO rows 100 - 1000 x larger: late tiles outgrow the scale chosen from the first ones)"""
    B, H, d = len(lens), 2, 128
    q, kv, do = _rand(B, Tq, d, seed=1), _rand(B, Tk, 2 * d, seed=2), _rand(B, Tq, d, seed=3) * mag
    if grow:
        kv[:, Tk - Tk // 4:, d:] *= 100.0
        do[:, Tq - Tq // 4:] *= 1000.0
    _, _, dq_ref, dkv_ref, _ = _reference(q, kv, do, lens, causal, H)
    r = _run_img(q, kv, do, lens, causal, H, want_attn=False)
    assert _rel(r["dq"], dq_ref) < TOL, _rel(r["dq"], dq_ref)
    assert _rel(r["dkv"][..., :d], dkv_ref[..., :d]) < TOL and _rel(r["dkv"][..., d:], dkv_ref[..., d:]) < TOL
    if grow:     # the small early rows on their own, not drowned by the large late ones
        n = Tq - Tq // 4
        assert _rel(r["dq"][:, :n], dq_ref[:, :n]) < 4 * TOL, _rel(r["dq"][:, :n], dq_ref[:, :n])


@pytest.mark.parametrize("vs,qs,ks", [(1e4, 1.0, 1.0), (1e-5, 1e3, 1e-3), (1e6, 1e-4, 1e4), (1.0, 3e-3, 3e2), (1e3, 1e3, 1e3)])
@pytest.mark.parametrize("causal", [0, 1])
@pytest.mark.parametrize("ragged_rows", [False, True])
def test_attention_on_head_images_any_magnitude(vs, qs, ks, causal, ragged_rows):
    """Q, K and V of any magnitude -- per tensor (as test_fp16x3_attention_any_magnitude) and, `ragged_rows`, per ROW: value rows
    spread over six decades, query and key rows over two (softmax rows become peaked; what must hold is accuracy against fp64
    on the same operands), an outlier value, an all-zero value row and an all-zero key row -- the per-(row, head) scales keep
    22 bits for every row.  The last tensor case has scores of +-1e6: every softmax row is one-hot, dq and dk are exactly zero."""
    if ragged_rows and qs * ks >= 100:
        pytest.skip("row factors on top of saturated scores leave some rows unsaturated: no closed-form bound")
    B, H, d, T = 2, 2, 128, 160
    lens = [160, 97]
    q, kv, do = _rand(B, T, d, seed=1) * qs, _rand(B, T, 2 * d, seed=2), _rand(B, T, d, seed=3) * 1e-6
    kv[..., :d] *= ks
    kv[..., d:] *= vs
    kv[0, 5, d + 3] *= 50.0
    if ragged_rows:
        g = torch.Generator().manual_seed(11)
        fv = (10.0 ** (torch.rand(B, T, 1, generator=g) * 6 - 3)).to(_dev())
        kv[..., d:] *= fv                                          # value rows over six decades
        fq = (10.0 ** (torch.rand(B, T, 1, generator=g) * 2 - 1)).to(_dev())
        q *= fq
        kv[..., :d] /= (10.0 ** (torch.rand(B, T, 1, generator=g) * 2 - 1)).to(_dev())
        kv[1, 7, d:] = 0.0
        kv[0, 9, :d] = 0.0
    o_ref, _, dq_ref, dkv_ref, _ = _reference(q, kv, do, lens, causal, H)
    r = _run_img(q, kv, do, lens, causal, H, want_attn=False)
    for k in ("o", "dq", "dkv"):
        assert torch.isfinite(r[k]).all(), k
    assert _rel(r["o"], o_ref) < TOL, _rel(r["o"], o_ref)
    assert _rel(r["dkv"][..., d:], dkv_ref[..., d:]) < TOL, _rel(r["dkv"][..., d:], dkv_ref[..., d:])
    if qs * ks < 100:
        assert _rel(r["dq"], dq_ref) < TOL and _rel(r["dkv"][..., :d], dkv_ref[..., :d]) < TOL, \
            (_rel(r["dq"], dq_ref), _rel(r["dkv"][..., :d], dkv_ref[..., :d]))
    else:
        noise = 2.0 ** -20 * float(do.abs().max()) * float(kv[..., d:].abs().max()) * 8.0
        assert float(dq_ref.abs().max()) == 0.0
        assert float((r["dq"] - dq_ref.float()).abs().max()) < noise * float(kv[..., :d].abs().max())
        assert float((r["dkv"][..., :d] - dkv_ref[..., :d].float()).abs().max()) < noise * float(q.abs().max())


def test_full_size_attention_on_head_images_is_reproducible():
    """BASELINE shapes (64 x 4 heads x 870 frames causal; 870 x 100 cross with weights), dropout on: two runs give the same bits
    (the LDS-DMA ring has no ordering left to chance), everything finite, rows of weights sum to 1 / (1 - p) times their kept mass."""
    B, H, d = 64, 4, 256
    for causal, Tq, Tk in ((1, 870, 870), (0, 870, 100)):
        lens = [Tk] * (B // 2) + [max(1, Tk - 7 * i) for i in range(B // 2)]
        q, kv, do = _rand(B, Tq, d, seed=5), _rand(B, Tk, 2 * d, seed=6), _rand(B, Tq, d, seed=7) * 1e-5
        a = _run_img(q, kv, do, lens, causal, H, p_drop=0.1, seed=5)
        torch.randn(1 << 24, device=_dev())
        b = _run_img(q, kv, do, lens, causal, H, p_drop=0.1, seed=5)
        for k in ("o", "dq", "dkv", "lse"):
            assert torch.isfinite(a[k]).all() and torch.equal(a[k], b[k]), k
        if a["attn"] is not None:
            assert torch.equal(a["attn"], b["attn"])


def _decoder_stack(layers=3, d=256, h=4, ff=512, scales=(1.0, 30.0, 0.01)):
    from transformertts_amd.model import layers as L
    torch.manual_seed(21)
    dec = L.TransformerDecoder(L.TransformerDecoderLayer(d, h, ff, dropout=0.0), layers)
    for p in dec.parameters():                 # (deep copies of one layer: make the layers differ)
        torch.nn.init.normal_(p, std=0.06) if p.dim() > 1 else torch.nn.init.normal_(p, mean=0.3, std=0.2)
    with torch.no_grad():                      # K/V projections of very different magnitude per layer: ONE image scale serves them all
        for layer, sc in zip(dec.layers, scales):
            layer.multihead_attn.in_proj_weight[d:] *= sc
            layer.multihead_attn.in_proj_bias[d:] *= sc
    return dec.cuda().train()


def test_one_kv_projection_for_all_decoder_layers():
    """ops.cross_kv_projection: the K/V projections of every decoder layer's cross-attention as ONE head-image GEMM over the
    encoder memory (N = layers x 2 d_model) on a STACKED weight image, one data-gradient GEMM of reduction depth N, one
    weight-gradient GEMM reduced into each layer's rows -- against the reference's structure (model/layers.py:54-74: one
    projection per layer; torch.nn.TransformerDecoder in fp64 with the same state dict): outputs, per-head alignments, input
    and parameter gradients; and against this library's own per-layer path (FUSED_CROSS_KV off).  The layers' K/V weights differ
    by x30 / x0.01: rows of the stacked image share one power-of-two scale, their per-(row, head) output scales do the rest."""
    from conftest import rel_l2
    from transformertts_amd import ops
    d, h, ff, B, Tq, Tk = 256, 4, 512, 3, 150, 70
    dec = _decoder_stack(3, d, h, ff)
    ref = torch.nn.TransformerDecoder(torch.nn.TransformerDecoderLayer(d, h, ff, dropout=0.0, batch_first=True), 3)
    ref.load_state_dict(dec.state_dict(), strict=True)
    ref = ref.double().train()
    g = torch.Generator().manual_seed(8)
    tgt, mem, dy = torch.randn(B, Tq, d, generator=g), torch.randn(B, Tk, d, generator=g), torch.randn(B, Tq, d, generator=g)
    tl, ml = [150, 97, 33], [70, 41, 9]
    t64, m64 = tgt.double().requires_grad_(True), mem.double().requires_grad_(True)
    causal = torch.triu(torch.ones(Tq, Tq, dtype=torch.bool), 1)
    tk = torch.arange(Tq)[None, :] >= torch.tensor(tl)[:, None]
    mk = torch.arange(Tk)[None, :] >= torch.tensor(ml)[:, None]
    y_ref = ref(t64, m64, tgt_mask=causal, tgt_key_padding_mask=tk, memory_key_padding_mask=mk)
    y_ref.backward(dy.double())
    live = (~tk).unsqueeze(-1).double()

    def run(fused):
        ops.FUSED_CROSS_KV = fused
        try:
            for p in dec.parameters():
                p.grad = None
            tc, mc = tgt.cuda().requires_grad_(True), mem.cuda().requires_grad_(True)
            y, al = dec(tc, mc, tgt_lens=torch.tensor(tl).cuda(), memory_lens=torch.tensor(ml).cuda())
            y.backward(dy.cuda())
            return y.detach(), [a.detach() for a in al], tc.grad, mc.grad, {n: p.grad.clone() for n, p in dec.named_parameters()}
        finally:
            ops.FUSED_CROSS_KV = True
    assert ops.cross_kv_ok(mem.cuda(), [l.multihead_attn for l in dec.layers], h)
    yf, af, dtf, dmf, gf = run(True)
    yu, au, dtu, dmu, gu = run(False)
    # vs torch fp64 (padded query rows are not comparable: torch's -inf rows; compare live rows)
    assert rel_l2(yf.cpu().double() * live, y_ref.detach() * live) < 2e-5
    assert rel_l2(dmf.cpu(), m64.grad) < 2e-5 and rel_l2(dtf.cpu().double() * live, t64.grad * live) < 2e-5
    for (n, q) in ref.named_parameters():
        if q.grad.norm() > 1e-9 and not n.endswith("in_proj_bias"):
            assert rel_l2(gf[n].cpu(), q.grad) < 3e-5, (n, rel_l2(gf[n].cpu(), q.grad))
    # vs the per-layer path of this library: same arithmetic up to the weight image's shared scale
    assert rel_l2(yf, yu) < 5e-6 and rel_l2(dmf, dmu) < 5e-6 and rel_l2(dtf, dtu) < 5e-6
    for a, b in zip(af, au):
        assert rel_l2(a, b) < 5e-6
    for n in gf:
        if gu[n].norm() > 1e-9:
            assert rel_l2(gf[n], gu[n]) < 1e-5, (n, rel_l2(gf[n], gu[n]))
    # a second fused run gives the same bits (fixed-order reductions, no atomics on values)
    y2, _, dt2, dm2, g2 = run(True)
    assert torch.equal(yf, y2) and torch.equal(dmf, dm2) and all(torch.equal(gf[n], g2[n]) for n in gf)


def test_head_image_is_not_a_tensor():
    """`linear(..., head_image_sections=n)` returns a HeadImage: typed-fp32 cells that hold f16 pairs must not be sliced,
    cloned, hooked or fed to the fp32 kernels by accident -- the wrapper has no tensor interface, and the attention
    wrappers refuse a mix of image and fp32 operands."""
    from transformertts_amd import ops
    x, w, b = _rand(2, 40, 256, seed=1), _rand(768, 256, seed=2, scale=0.06), _rand(768, seed=3)
    qkv = ops.linear(x, w, b, head_image_sections=3)
    assert isinstance(qkv, ops.HeadImage) and not isinstance(qkv, torch.Tensor) and tuple(qkv.shape) == (2, 40, 768)
    for misuse in (lambda: qkv[..., :256], lambda: qkv.clone(), lambda: qkv.contiguous(), lambda: qkv + 1, lambda: torch.isfinite(qkv),
                   lambda: qkv.register_hook(print)):
        with pytest.raises((TypeError, AttributeError)):
            misuse()
    lens = torch.tensor([40, 17], device=_dev())
    o = ops.self_attention(qkv, lens, 4, True, 0.0, 0)
    ref = ops.self_attention(ops.linear(x, w, b, publish_amax=True), lens, 4, True, 0.0, 0)
    assert _rel(o, ref) < 1e-5
    q = ops.linear(x, w[:256].contiguous(), b[:256].contiguous(), head_image_sections=1)
    kv32 = ops.linear(x, w[256:].contiguous(), b[256:].contiguous(), publish_amax=True)
    with pytest.raises(ValueError):
        ops.cross_attention(q, kv32, lens, 4, 0.0, 0)
