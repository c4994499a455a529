#!/usr/bin/env python3
"""Where a wave of gemm_h3i_kernel spends its ticks (development aid; needs a library built with -DTTTS_H3I_STAMPS:
bash tools/build_variant.sh /tmp/h3i_stamps.so -DTTTS_H3I_STAMPS, then TTTS_LIB=/tmp/h3i_stamps.so python tools/h3i_stamps.py [M N K])."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from transformertts_amd import _lib, ops
from transformertts_amd.ops import _p, _stream

lib = _lib.load()
dev = torch.device("cuda:0")
M, N, K = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (55680, 1024, 256)
x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * K ** -0.5; b = torch.randn(N, device=dev)
y = torch.empty(M, N, device=dev)
img = torch.empty(M, K, 2, dtype=torch.int16, device=dev); inv = torch.empty(M, device=dev)
lib.ttts_act_image(_p(x), _p(img), _p(inv), M, K, _stream())
pl = ops._planes(w, 8, N, K).clone()
run = lambda: lib.ttts_linear_fwd_h3i(_p(img), _p(inv), _p(pl), _p(b), None, _p(y), M, N, K, 0, 0.0, 0, None, None, _stream())
for _ in range(5):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
raw = ctypes.CDLL(_lib.LIB_PATH)
n = 512 * 4 * 8
buf = (ctypes.c_ulonglong * n)()
raw.ttts_dbg_h3i_read_stamps(buf, ctypes.c_size_t(n))
# rows per tile as csrc/gemm_h3i.hip h3i_big_tile picks them: the 256-row tile has eight waves per workgroup (at most 256 of them)
NW = 8 if (N >= 1024 and K <= 256 and (-(-M // 256)) * (-(-N // 256)) >= 256) else 4
st = np.frombuffer(buf, dtype=np.uint64).reshape(2048 // NW, NW, 8).astype(np.float64)
live = st[:, 0, 5] > 0
st = st[live]
print(f"M={M} N={N} K={K}: {e0.elapsed_time(e1) * 1e3:.1f} us (with stamps), {int(live.sum())} workgroups")
tiles, kts = st[:, :, 5].mean(), st[:, :, 6].mean()
print(f"tiles per workgroup {tiles:.2f}, k-tiles {kts:.1f}; whole kernel {st[:, :, 4].mean():.0f} ticks (s_memtime: 100 MHz)")
names = ["wait k-tile (vmcnt + barrier)", "issue DMAs", "fragments + products", "epilogue"]
for i, nm in enumerate(names):
    per = st[:, :, i].mean()
    print(f"  {nm:32s} {per:9.0f} ticks per wave = {per / st[:, :, 4].mean() * 100:5.1f} %   per k-tile {per / kts:7.1f}" if i < 3 else
          f"  {nm:32s} {per:9.0f} ticks per wave = {per / st[:, :, 4].mean() * 100:5.1f} %   per tile {per / tiles:7.1f}")
pass
print("  by wave (wait, issue, compute, epilogue):", [[int(st[:, wv, i].mean()) for i in range(4)] for wv in range(NW)])
