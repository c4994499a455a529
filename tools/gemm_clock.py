#!/usr/bin/env python3
"""In-kernel clock of the forward GEMM kernels under sustained load (MI355X_MICROARCH.md, DVFS give-back item 6): needs a library
built with -DTTTS_CLOCK_STAMPS (bash tools/build_variant.sh <out.so> -DTTTS_CLOCK_STAMPS; TTTS_LIB=<out.so> tools/gemm_clock.py).
Each kernel runs back to back on random data for >= 2 s; the stamps of the LAST launch give, per workgroup,
delta s_memtime / delta s_memrealtime x 100 MHz; the median over workgroups is the clock the kernel held."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from transformertts_amd import _lib, ops
from transformertts_amd.ops import _p, _stream
lib = _lib.load(); dev = torch.device("cuda:0")
raw = ctypes.CDLL(_lib.LIB_PATH)
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 2.5

def clock(reader):
    buf = (ctypes.c_ulonglong * 1024)()
    getattr(raw, reader)(buf, ctypes.c_size_t(1024))
    a = np.frombuffer(buf, dtype=np.uint64).reshape(512, 2).astype(np.float64)
    a = a[a[:, 1] > 0]
    return np.median(a[:, 0] / a[:, 1]) * 0.1, len(a)       # cycles per 10 ns tick -> GHz

def sustained(f):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); n = 0
    e0.record()
    while time.perf_counter() - t0 < secs:
        for _ in range(50): f()
        n += 50
        torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

M = 55680
for N, K, what in ((1024, 256, "FFN1 forward"), (768, 256, "in-projection"), (256, 1024, "FFN2 forward")):
    x, w, b = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev) * K ** -0.5, torch.randn(N, device=dev)
    y = torch.empty(M, N, device=dev); xa = ops._amax(x)
    p4, p8 = ops._planes(w, 4, N, K).clone(), ops._planes(w, 8, N, K).clone()
    us = sustained(lambda: lib.ttts_linear_fwd_h3(_p(x), _p(p4), _p(b), None, _p(y), M, N, K, 0, 0.0, 0, None, 0, 0, _p(xa), None, _stream()))
    ghz, nw = clock("ttts_dbg_read_clock_h3_wide")
    print(f"gemm_h3_wide_kernel  {what:14s} M={M} N={N:4d} K={K:4d}: {us:6.1f} us per launch sustained, in-kernel clock {ghz:.3f} GHz (median of {nw} workgroups)")
    us = sustained(lambda: lib.ttts_linear_fwd_h3d(_p(x), _p(p8), _p(b), None, _p(y), M, N, K, 0, 0.0, 0, None, _p(xa), None, _stream()))
    ghz, nw = clock("ttts_dbg_read_clock_h3i")
    print(f"gemm_h3i_kernel<raw> {what:14s} M={M} N={N:4d} K={K:4d}: {us:6.1f} us per launch sustained, in-kernel clock {ghz:.3f} GHz (median of {nw} workgroups)")
