#!/usr/bin/env python3
"""Which tile of gemm_h3 is fastest on a shape?  Times ttts_linear_fwd_h3 (bias + residual epilogue) with every tile forced, on the
step's shapes at a given row count; needs a library built with -DTTTS_TUNE (bash tools/build_variant.sh /tmp/tune.so -DTTTS_TUNE;
TTTS_LIB=/tmp/tune.so python3 tools/tile_sweep.py 13920 1600).  HIP events around 20 launches with 1 ms of idle in front of each
batch (the step's GEMMs do not run back to back)."""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformertts_amd import _lib, ops
from transformertts_amd.ops import _p, _stream

lib = _lib.load()
force = ctypes.CDLL(_lib.LIB_PATH).ttts_dbg_force_h3_tile
dev = torch.device("cuda:0")
TILES = {0: "auto", 1: "64x64", 2: "128x128", 3: "64x128", 6: "256x256", 7: "256x128/8w", 8: "256x128 pair"}
rows = [int(a) for a in (sys.argv[1:sys.argv.index("conv")] if "conv" in sys.argv else sys.argv[1:])] or [13920, 1600]
for M in rows:
    for N, K in ((256, 256), (768, 256), (1024, 256), (256, 1024), (512, 256), (256, 768)):
        x, w, b = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev) * K ** -0.5, torch.randn(N, device=dev)
        res, y = torch.randn(M, N, device=dev), torch.empty(M, N, device=dev)
        pl, xa = ops._planes(w, 4, N, K), ops._amax(x)
        out = []
        for t, name in TILES.items():
            force(t)
            f = lambda: _lib.check(lib.ttts_linear_fwd_h3(_p(x), _p(pl), _p(b), _p(res), _p(y), M, N, K, 0, 0.0, 0, None, 0, 0, _p(xa), None,
                                                          _stream()), "fwd")      # noqa: E731
            try:
                for _ in range(3):
                    f()
            except RuntimeError:
                out.append(f"{name} n/a")
                continue
            best = 1e9
            for _ in range(5):
                torch.cuda.synchronize()
                time.sleep(0.001)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    f()
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
            out.append(f"{name} {best:6.1f}")
        force(0)
        print(f"M={M:6d} N={N:5d} K={K:5d}: " + " | ".join(out), flush=True)

# convolutions (k = 5, 256 -> 256 channels: the post-net / the encoder pre-net) at B x T rows: `conv B T [B T ...]` behind the row counts
if "conv" in sys.argv:
    cv = [int(a) for a in sys.argv[sys.argv.index("conv") + 1:]]
    for B, T in zip(cv[0::2], cv[1::2]):
        cin = cout = 256
        x, w, b = torch.randn(B, T, cin, device=dev), torch.randn(cout, cin, 5, device=dev) * (5 * cin) ** -0.5, torch.randn(cout, device=dev)
        y = torch.empty(B, T, cout, device=dev)
        pl, xa = ops._planes(w, 6, cout, 5 * cin, cin, 5), ops._amax(x)
        out = []
        for t, name in TILES.items():
            force(t)
            f = lambda: _lib.check(lib.ttts_conv1d_fwd_h3(_p(x), _p(pl), _p(b), _p(y), B, T, cin, cout, 5, _p(xa), None, _stream()), "conv")   # noqa: E731
            try:
                for _ in range(3):
                    f()
            except RuntimeError:
                out.append(f"{name} n/a")
                continue
            best = 1e9
            for _ in range(5):
                torch.cuda.synchronize()
                time.sleep(0.001)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    f()
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
            out.append(f"{name} {best:6.1f}")
        force(0)
        print(f"conv k5 256->256  B={B:4d} T={T:4d} (M={B * T:6d}): " + " | ".join(out), flush=True)

