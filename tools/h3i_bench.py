#!/usr/bin/env python3
"""gemm_h3i_kernel (image activation operand, LDS-DMA, two workgroups per CU) against gemm_h3 (fp32 operand split in the
loader): correctness against fp64 on small and ragged shapes, then interleaved timing rounds (HIP events) on the shapes of the
training step.  The image is made once outside the timed region (in the step its producer writes it); `act_image` is timed on
its own line.   usage: python tools/h3i_bench.py [M [d_model]]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformertts_amd import _lib, ops
from transformertts_amd.ops import _p, _stream

lib = _lib.load()
dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 55680


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def amax_of(x):
    out = torch.zeros(ops.AMAX_SLOTS, device=dev)
    _lib.check(lib.ttts_amax_partials(_p(x), x.numel(), _p(out), _stream()), "amax")
    return out


def image_of(x):
    m, k = x.shape
    if k > 1024:                      # no image form for rows this wide: those columns of the table fall back to the fp32 call
        return None, None
    img = torch.empty(m, k, 2, dtype=torch.int16, device=dev)
    inv = torch.empty(m, dtype=torch.float32, device=dev)
    _lib.check(lib.ttts_act_image(_p(x), _p(img), _p(inv), m, k, _stream()), "act_image")
    return img, inv


def rel(a, b):
    return float((a.double() - b).norm() / b.norm())


# ---------------------------------------------------------------- correctness
torch.manual_seed(0)
worst = 0.0
for (m, n, k, scale) in [(300, 96, 64, 1.0), (129, 256, 256, 1e-4), (1000, 768, 256, 30.0), (257, 1024, 256, 1.0), (515, 256, 1024, 1.0),
                         (128 * 9 + 5, 260, 512, 1.0)]:
    x = torch.randn(m, k, device=dev) * scale
    x[3] *= 1e3                              # rows of very different magnitude: the scale is per row
    x[5] = 0.0
    w = torch.randn(n, k, device=dev) * k ** -0.5
    b = torch.randn(n, device=dev)
    res = torch.randn(m, n, device=dev)
    img, inv = image_of(x)
    y = torch.full((m, n), float("nan"), device=dev)
    am = torch.zeros(ops.AMAX_SLOTS, device=dev)
    _lib.check(lib.ttts_linear_fwd_h3i(_p(img), _p(inv), _p(ops._planes(w, 8, n, k)), _p(b), _p(res), _p(y), m, n, k, 0, 0.0, 0, None,
                                       _p(am), _stream()), "fwd_h3i")
    ref = x.double() @ w.double().t() + b.double() + res.double()
    e = rel(y, ref)
    # row-wise: the small rows must be as accurate as the big ones (that is what the per-row scale buys)
    rows = ((y.double() - ref).norm(dim=1) / ref.norm(dim=1).clamp_min(1e-300)).max().item()
    ok_amax = abs(float(am.max()) - float(y.abs().max())) <= 1e-6 * float(y.abs().max())
    print(f"fwd  M={m} N={n} K={k} scale={scale:g}: rel-L2 {e:.2e} worst row {rows:.2e} amax ok {ok_amax}", flush=True)
    worst = max(worst, e, rows if torch.isfinite(torch.tensor(rows)) else 1.0)
    assert ok_amax
    yd_ = torch.full((m, n), float("nan"), device=dev)
    _lib.check(lib.ttts_linear_fwd_h3d(_p(x), _p(ops._planes(w, 8, n, k)), _p(b), _p(res), _p(yd_), m, n, k, 0, 0.0, 0, None,
                                       _p(amax_of(x)), None, _stream()), "fwd_h3d")
    e = rel(yd_, ref)
    print(f"h3d  M={m} N={n} K={k}: rel-L2 {e:.2e}", flush=True)
    worst = max(worst, e)
    # relu + dropout epilogue: same mask as the fp32-operand kernel (the mask is a function of seed and element index)
    y1, y2 = torch.empty(m, n, device=dev), torch.empty(m, n, device=dev)
    _lib.check(lib.ttts_linear_fwd_h3i(_p(img), _p(inv), _p(ops._planes(w, 8, n, k)), _p(b), None, _p(y1), m, n, k, 1, 0.1, 77, None,
                                       None, _stream()), "fwd_h3i")
    _lib.check(lib.ttts_linear_fwd_h3(_p(x), _p(ops._planes(w, 4, n, k)), _p(b), None, _p(y2), m, n, k, 1, 0.1, 77, None, 0, 0,
                                      _p(amax_of(x)), None, _stream()), "fwd_h3")
    assert torch.equal(y1 == 0, y2 == 0) or rel(y1, y2.double()) < 1e-5
    # data gradient with relu gate: dx[m, k] = dy[m, n] . w[n, k], gated by h > 0
    dy = torch.randn(m, n, device=dev) * 1e-5
    if n % 32 == 0:
        h = torch.relu(torch.randn(m, k, device=dev))
        dimg, dinv = image_of(dy)
        dx = torch.empty(m, k, device=dev)
        _lib.check(lib.ttts_linear_bwd_data_h3i(_p(dimg), _p(dinv), _p(ops._planes(w, 9, k, n)), None, _p(dx), m, n, k, _p(h), 1.0 / 0.9,
                                                None, _stream()), "bwd_h3i")
        refd = (dy.double() @ w.double()) * (h > 0).double() / 0.9
        e = rel(dx, refd)
        print(f"dgrad M={m} N={n} K={k}: rel-L2 {e:.2e}", flush=True)
        worst = max(worst, e)
        dx2 = torch.empty(m, k, device=dev)
        _lib.check(lib.ttts_linear_bwd_data_h3d(_p(dy), _p(ops._planes(w, 9, k, n)), None, _p(dx2), m, n, k, _p(h), 1.0 / 0.9,
                                                _p(amax_of(dy)), None, _stream()), "bwd_h3d")
        e = rel(dx2, refd)
        print(f"dgrad h3d M={m} N={n} K={k}: rel-L2 {e:.2e}", flush=True)
        worst = max(worst, e)
torch.cuda.synchronize()
print(f"worst rel-L2 {worst:.2e}")
assert worst < 5e-6, worst

# ---------------------------------------------------------------- timing
d = int(sys.argv[2]) if len(sys.argv) > 2 else 256
f = 4 * d
x = torch.randn(M, d, device=dev); h = torch.relu(torch.randn(M, f, device=dev)); skip = torch.randn(M, d, device=dev)
w1 = torch.randn(f, d, device=dev) * d ** -0.5; w2 = torch.randn(d, f, device=dev) * f ** -0.5; wq = torch.randn(3 * d, d, device=dev) * d ** -0.5
wo = torch.randn(d, d, device=dev) * d ** -0.5
b1 = torch.randn(f, device=dev); b2 = torch.randn(d, device=dev); bq = torch.randn(3 * d, device=dev)
dy = torch.randn(M, d, device=dev) * 1e-5; dh = torch.randn(M, f, device=dev) * 1e-5
yf = torch.empty(M, f, device=dev); yd = torch.empty(M, d, device=dev); yq = torch.empty(M, 3 * d, device=dev)
xa, ha, dya, dha = amax_of(x), amax_of(h), amax_of(dy), amax_of(dh)
am = torch.zeros(ops.AMAX_SLOTS, device=dev)
p1, p2, pq, po = (ops._planes(w1, 4, f, d).clone(), ops._planes(w2, 4, d, f).clone(), ops._planes(wq, 4, 3 * d, d).clone(),
                  ops._planes(wo, 4, d, d).clone())
p1t, p2t = ops._planes(w1, 5, d, f).clone(), ops._planes(w2, 5, f, d).clone()
k1, k2, kq, ko = (ops._planes(w1, 8, f, d).clone(), ops._planes(w2, 8, d, f).clone(), ops._planes(wq, 8, 3 * d, d).clone(),
                  ops._planes(wo, 8, d, d).clone())
k1t, k2t = ops._planes(w1, 9, d, f).clone(), ops._planes(w2, 9, f, d).clone()
xi, xv = image_of(x); hi, hv = image_of(h); dyi, dyv = image_of(dy); dhi, dhv = image_of(dh)
scr_i, scr_v = torch.empty_like(xi), torch.empty_like(xv)
cases = [
    (f"ffn1 fwd  relu+drop+amax  N={f} K={d}", 2.0 * M * f * d,
     lambda: lib.ttts_linear_fwd_h3(_p(x), _p(p1), _p(b1), None, _p(yf), M, f, d, 1, 0.1, 77, None, 0, 0, _p(xa), _p(am), _stream()),
     lambda: lib.ttts_linear_fwd_h3i(_p(xi), _p(xv), _p(k1), _p(b1), None, _p(yf), M, f, d, 1, 0.1, 77, None, _p(am), _stream()),
     lambda: lib.ttts_linear_fwd_h3d(_p(x), _p(k1), _p(b1), None, _p(yf), M, f, d, 1, 0.1, 77, None, _p(xa), _p(am), _stream())),
    (f"inproj fwd bias+amax      N={3 * d} K={d}", 2.0 * M * 3 * d * d,
     lambda: lib.ttts_linear_fwd_h3(_p(x), _p(pq), _p(bq), None, _p(yq), M, 3 * d, d, 0, 0.0, 0, None, 0, 0, _p(xa), _p(am), _stream()),
     lambda: lib.ttts_linear_fwd_h3i(_p(xi), _p(xv), _p(kq), _p(bq), None, _p(yq), M, 3 * d, d, 0, 0.0, 0, None, _p(am), _stream()),
     lambda: lib.ttts_linear_fwd_h3d(_p(x), _p(kq), _p(bq), None, _p(yq), M, 3 * d, d, 0, 0.0, 0, None, _p(xa), _p(am), _stream())),
    (f"outproj fwd res+drop      N={d} K={d}", 2.0 * M * d * d,
     lambda: lib.ttts_linear_fwd_h3(_p(x), _p(po), _p(b2), _p(skip), _p(yd), M, d, d, 0, 0.1, 79, None, 0, 0, _p(xa), None, _stream()),
     lambda: lib.ttts_linear_fwd_h3i(_p(xi), _p(xv), _p(ko), _p(b2), _p(skip), _p(yd), M, d, d, 0, 0.1, 79, None, None, _stream()),
     lambda: lib.ttts_linear_fwd_h3d(_p(x), _p(ko), _p(b2), _p(skip), _p(yd), M, d, d, 0, 0.1, 79, None, _p(xa), None, _stream())),
    (f"ffn2 fwd  res+drop        N={d} K={f}", 2.0 * M * f * d,
     lambda: lib.ttts_linear_fwd_h3(_p(h), _p(p2), _p(b2), _p(skip), _p(yd), M, d, f, 0, 0.1, 78, None, 0, 0, _p(ha), None, _stream()),
     lambda: lib.ttts_linear_fwd_h3i(_p(hi), _p(hv), _p(k2), _p(b2), _p(skip), _p(yd), M, d, f, 0, 0.1, 78, None, None, _stream()),
     lambda: lib.ttts_linear_fwd_h3d(_p(h), _p(k2), _p(b2), _p(skip), _p(yd), M, d, f, 0, 0.1, 78, None, _p(ha), None, _stream())),
    (f"ffn2 dgrad gate+amax      N={f} K={d}", 2.0 * M * f * d,
     lambda: lib.ttts_linear_bwd_data_h3(_p(dy), _p(p2t), None, _p(yf), M, d, f, _p(h), 1.0 / 0.9, _p(dya), _p(am), _stream()),
     lambda: lib.ttts_linear_bwd_data_h3i(_p(dyi), _p(dyv), _p(k2t), None, _p(yf), M, d, f, _p(h), 1.0 / 0.9, _p(am), _stream()),
     lambda: lib.ttts_linear_bwd_data_h3d(_p(dy), _p(k2t), None, _p(yf), M, d, f, _p(h), 1.0 / 0.9, _p(dya), _p(am), _stream())),
    (f"ffn1 dgrad residual       N={d} K={f}", 2.0 * M * f * d,
     lambda: lib.ttts_linear_bwd_data_h3(_p(dh), _p(p1t), _p(skip), _p(yd), M, f, d, None, 1.0, _p(dha), None, _stream()),
     lambda: lib.ttts_linear_bwd_data_h3i(_p(dhi), _p(dhv), _p(k1t), _p(skip), _p(yd), M, f, d, None, 1.0, None, _stream()),
     lambda: lib.ttts_linear_bwd_data_h3d(_p(dh), _p(k1t), _p(skip), _p(yd), M, f, d, None, 1.0, _p(dha), None, _stream())),
]
def img_ok(name):
    return f <= 1024 or "K=%d" % f not in name


res = {}
for rnd in range(3):
    for name, fl, fa, fb, fc in cases:
        res.setdefault(name, [[], [], []])
        res[name][0].append(timeit(fa))
        res[name][1].append(timeit(fb) if img_ok(name) else float("nan"))
        res[name][2].append(timeit(fc))
for name, fl, fa, fb, fc in cases:
    a, b, c = min(res[name][0]), min(res[name][1]), min(res[name][2])
    print(f"{name:38s} h3 {a:6.1f} us {fl / a / 1e6:5.0f} TF | image {b:6.1f} us {fl / b / 1e6:5.0f} TF ({a / b:.2f}x) | "
          f"fp32 by DMA {c:6.1f} us {fl / c / 1e6:5.0f} TF ({a / c:.2f}x)", flush=True)
t = timeit(lambda: lib.ttts_act_image(_p(x), _p(scr_i), _p(scr_v), M, d, _stream()))
print(f"act_image {M} x {d}: {t:.1f} us")
