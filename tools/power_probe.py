#!/usr/bin/env python3
"""Socket power and shader clock while ONE kernel runs back to back (development aid for DESIGN 10.3 (d)).

A thread polls the amdgpu hwmon files (power1_input in uW, freq1_input = sclk in Hz; readable by an ordinary user) every
10 ms while the main thread keeps the stream full of one launch for ~2.5 s; the card that belongs to this process is the one
whose power moves.  Prints, per kernel: launch time, mean / max power, mean shader clock over the second half of the window
(the first half lets the power controller settle).   usage: python tools/power_probe.py [M]"""
import glob
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformertts_amd import _lib, ops
from transformertts_amd.ops import _p, _stream

lib = _lib.load()
dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 55680
CARD = None
HW = sorted(d for d in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*") if os.path.exists(d + "/power1_input"))


def read(path):
    try:
        with open(path) as f:
            return float(f.read())
    except Exception:  # noqa: BLE001
        return float("nan")


class Poller(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.stop, self.rows = False, []

    def run(self):
        while not self.stop:
            self.rows.append((time.perf_counter(), [read(d + "/power1_input") * 1e-6 for d in HW],
                              [read(d + "/freq1_input") * 1e-6 for d in HW]))
            time.sleep(0.01)


def sustained(name, fn, seconds=2.5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    n = max(50, int(seconds * 1e3 / max(e0.elapsed_time(e1), 1e-3)))
    p = Poller(); p.start()
    t0 = time.perf_counter()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    p.stop = True; p.join()
    rows = [r for r in p.rows if t0 + 0.5 * (t1 - t0) <= r[0] <= t1]
    if not rows or not HW:
        print(f"{name:44s} {e0.elapsed_time(e1) / n * 1e3:8.1f} us   (no hwmon samples)"); return
    mean_p = [sum(r[1][i] for r in rows) / len(rows) for i in range(len(HW))]
    i = CARD if CARD is not None else max(range(len(HW)), key=lambda j: mean_p[j])
    clk = [r[2][i] for r in rows]
    print(f"{name:44s} {e0.elapsed_time(e1) / n * 1e3:8.1f} us   {mean_p[i]:6.0f} W mean {max(r[1][i] for r in rows):6.0f} W max   "
          f"sclk {sum(clk) / len(clk):5.0f} MHz (min {min(clk):.0f})   [{len(rows)} samples, {os.path.basename(HW[i])}]", flush=True)


def amax_of(x):
    out = torch.zeros(ops.AMAX_SLOTS, device=dev)
    _lib.check(lib.ttts_amax_partials(_p(x), x.numel(), _p(out), _stream()), "amax")
    return out


def image_of(x):
    m, k = x.shape
    img = torch.empty(m, k, 2, dtype=torch.int16, device=dev)
    inv = torch.empty(m, dtype=torch.float32, device=dev)
    _lib.check(lib.ttts_act_image(_p(x), _p(img), _p(inv), m, k, _stream()), "act_image")
    return img, inv


print(f"hwmon dirs: {len(HW)}; cap {read(HW[0] + '/power1_cap') * 1e-6:.0f} W" if HW else "no hwmon", flush=True)
torch.manual_seed(0)
d, f = 256, 1024
x = torch.randn(M, d, device=dev); h = torch.relu(torch.randn(M, f, device=dev))
w1 = torch.randn(f, d, device=dev) * d ** -0.5; w2 = torch.randn(d, f, device=dev) * f ** -0.5
b1 = torch.randn(f, device=dev); b2 = torch.randn(d, device=dev)
yf = torch.empty(M, f, device=dev); yd = torch.empty(M, d, device=dev)
dy = torch.randn(M, d, device=dev) * 1e-5
xa, ha, dya = amax_of(x), amax_of(h), amax_of(dy)
p1, p2 = ops._planes(w1, 4, f, d).clone(), ops._planes(w2, 4, d, f).clone()
k1, k2 = ops._planes(w1, 8, f, d).clone(), ops._planes(w2, 8, d, f).clone()
xi, xv = image_of(x); hi, hv = image_of(h)

# which hwmon directory is this process's card: the one whose power rises when the card is loaded (the host's other cards
# belong to other tenants and may be busy)
CARD = None
if HW:
    time.sleep(1.0)
    idle = [read(d + "/power1_input") for d in HW]
    t_end = time.perf_counter() + 0.8
    while time.perf_counter() < t_end:              # ~0.8 s of copies, synchronised in chunks so that the queue stays short
        for _ in range(200):
            yf.copy_(h)
        torch.cuda.synchronize()
    for _ in range(400):
        yf.copy_(h)
    time.sleep(0.01)
    busy = [read(d + "/power1_input") for d in HW]
    torch.cuda.synchronize()
    CARD = max(range(len(HW)), key=lambda j: busy[j] - idle[j])
    print(f"this card: {HW[CARD]} (idle {idle[CARD] * 1e-6:.0f} W -> loaded {busy[CARD] * 1e-6:.0f} W)", flush=True)
time.sleep(1.0)
sustained("idle (a 4-byte fill)", lambda: yd[:1].zero_(), 1.5)
sustained("copy 228 MB (HBM stream)", lambda: yf.copy_(h))
sustained("gemm_h3_wide   FFN1 fwd N=1024 K=256",
          lambda: lib.ttts_linear_fwd_h3(_p(x), _p(p1), _p(b1), None, _p(yf), M, f, d, 0, 0.0, 0, None, 0, 0, _p(xa), None, _stream()))
sustained("gemm_h3i image FFN1 fwd N=1024 K=256",
          lambda: lib.ttts_linear_fwd_h3i(_p(xi), _p(xv), _p(k1), _p(b1), None, _p(yf), M, f, d, 0, 0.0, 0, None, None, _stream()))
sustained("gemm_h3i raw   FFN1 fwd N=1024 K=256",
          lambda: lib.ttts_linear_fwd_h3d(_p(x), _p(k1), _p(b1), None, _p(yf), M, f, d, 0, 0.0, 0, None, _p(xa), None, _stream()))
sustained("gemm_h3<256,256,2,4> FFN2 fwd N=256 K=1024",
          lambda: lib.ttts_linear_fwd_h3(_p(h), _p(p2), _p(b2), None, _p(yd), M, d, f, 0, 0.0, 0, None, 0, 0, _p(ha), None, _stream()))
sustained("gemm_h3i image FFN2 fwd N=256 K=1024",
          lambda: lib.ttts_linear_fwd_h3i(_p(hi), _p(hv), _p(k2), _p(b2), None, _p(yd), M, d, f, 0, 0.0, 0, None, None, _stream()))

# weight gradient 1024 x 256 and LayerNorm through the ops layer (their workspaces live there)
xg = x.clone().requires_grad_(True)
w1g = w1.clone().requires_grad_(True)
ln_w = torch.ones(d, device=dev, requires_grad=True); ln_b = torch.zeros(d, device=dev, requires_grad=True)
sustained("layernorm fwd 55680 x 256 (ops.layer_norm)", lambda: ops.layer_norm(x, ln_w.detach(), ln_b.detach(), 1e-5))

B, H, T = 64, 4, 870
if M == B * T:
    qkv = torch.randn(B, T, 3 * d, device=dev)
    lens = torch.full((B,), T, dtype=torch.int64, device=dev)
    with torch.no_grad():
        sustained("self-attention fwd causal B=64 H=4 T=870", lambda: ops.self_attention(qkv, lens, H, True, 0.1, 5))
    qg = qkv.clone().requires_grad_(True)

    def fwd_bwd():
        o = ops.self_attention(qg, lens, H, True, 0.1, 5)
        o = o[0] if isinstance(o, tuple) else o
        o.backward(o.detach())
        qg.grad = None
    sustained("self-attention fwd + dq + dkv", fwd_bwd)

# the MFMA-only microbenchmark (its own process) under the same poller
exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "micro", "mfma_shape")
if os.path.exists(exe) and HW:
    import subprocess
    torch.cuda.synchronize()
    p = Poller(); p.start()
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120).stdout
    p.stop = True; p.join()
    mean_all = [sum(r[1][i] for r in p.rows) / len(p.rows) for i in range(len(HW))]
    i = CARD
    print("mfma_shape (MFMAs alone): power over its whole run: mean %.0f W, max %.0f W; sclk min %.0f MHz" % (
        mean_all[i], max(r[1][i] for r in p.rows), min(r[2][i] for r in p.rows)))
    print(out)

# the whole training step (bench.py in its own process, 300 timed steps): power and clock over the samples taken while it steps
if HW and "--no-step" not in sys.argv:
    import subprocess
    torch.cuda.synchronize()
    del x, h, yf, yd, xi, hi
    torch.cuda.empty_cache()
    p = Poller(); p.start()
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), "--steps", "300",
                        "--warmup", "5", "--sustain", "0", "--no-probe", "--no-cpu-baseline", "--no-alignments-figure"],
                       capture_output=True, text=True, timeout=600)
    p.stop = True; p.join()
    pw = [row[1][CARD] for row in p.rows]
    hot = [row for row in p.rows if row[1][CARD] >= 0.9 * max(pw)]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    ms = __import__("json").loads(line[-1])["ms_per_step"] if line else float("nan")
    print("training step (bench.py, %.2f ms/step): %d samples at >= 90 %% of the run's maximum power: mean %.0f W, sclk mean %.0f MHz (min %.0f, max %.0f)" % (
        ms, len(hot), sum(row[1][CARD] for row in hot) / max(1, len(hot)), sum(row[2][CARD] for row in hot) / max(1, len(hot)),
        min(row[2][CARD] for row in hot), max(row[2][CARD] for row in hot)))
