#!/usr/bin/env python3
"""Where a wave of attn_fwd_img_kernel (causal) spends its ticks (development aid; needs a library built with -DTTTS_AIMG_STAMPS:
bash tools/build_variant.sh /tmp/aimg.so -DTTTS_AIMG_STAMPS, then TTTS_LIB=/tmp/aimg.so python tools/aimg_stamps.py [B H T p_drop])."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from transformertts_amd import _lib, ops
from transformertts_amd.ops import _p, _off, _stream

lib = _lib.load()
dev = torch.device("cuda:0")
B, H, T = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (64, 4, 870)
p_drop = float(sys.argv[4]) if len(sys.argv) > 4 else 0.1
d = H * 64
qkv = torch.randn(B * T, 3 * d, device=dev)
img, inv = torch.empty_like(qkv), torch.empty(3 * H, B * T, device=dev)
_lib.check(lib.ttts_head_image(_p(qkv), 3 * d, _p(img), 3 * d, _p(inv), B * T, 3 * d, _stream()), "head_image")
va = ops._amax(qkv[:, 2 * d:].contiguous())
lens = torch.full((B,), T, dtype=torch.int64, device=dev)
o = torch.empty(B, T, d, device=dev); stat = torch.empty(6, B, H, T, device=dev)
HM = H * B * T
run = lambda: lib.ttts_attention_fwd_img(_off(img, 0), _off(img, d), _off(img, 2 * d), _off(inv, 0), _off(inv, HM), _off(inv, 2 * HM), _p(o),
                                         _p(stat[0]), None, _p(lens), B, H, T, T, 3 * d, 3 * d, 3 * d, d, 1, 0.125, p_drop, 5, None, _p(va), None,
                                         _p(stat[1:]), 0, 0, 0, _stream())
for _ in range(5):
    assert run() == 0
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    run()
e1.record(); torch.cuda.synchronize()
print(f"B={B} H={H} T={T} p={p_drop}: {e0.elapsed_time(e1) * 100:.1f} us per launch")
raw = ctypes.CDLL(_lib.LIB_PATH)
if hasattr(raw, "ttts_dbg_aimg_read_stamps"):
    n = 2048 * 4 * 8
    buf = (ctypes.c_ulonglong * n)()
    raw.ttts_dbg_aimg_read_stamps(buf, ctypes.c_size_t(n))
    st = np.frombuffer(buf, dtype=np.uint64).reshape(2048, 4, 8).astype(np.float64)
    live = st[:, 0, 5] > 0
    st = st[live]
    tot = st[:, :, 5]
    print(f"{int(live.sum())} workgroups stamped; whole kernel {tot.mean():.0f} ticks per wave (max {tot.max():.0f}); sub-tiles per wave {st[:, :, 6].mean():.1f}")
    names = ["wait tile (vmcnt + barrier)", "issue DMAs", "scores + key scales", "softmax + dropout", "split + V^T reads + products", None, None, "prologue"]
    for i, nm in enumerate(names):
        if nm is None:
            continue
        per = st[:, :, i].mean()
        print(f"  {nm:32s} {per:9.0f} ticks per wave = {per / tot.mean() * 100:5.1f} %   per sub-tile {per / st[:, :, 6].mean():7.1f}")
    acc = sum(st[:, :, i].mean() for i in (0, 1, 2, 3, 4, 7))
    print(f"  unaccounted (epilogue, loop overhead) {tot.mean() - acc:9.0f} ticks = {(tot.mean() - acc) / tot.mean() * 100:5.1f} %")
    print("  by wave (wait, issue, scores, softmax, pv, total):", [[int(st[:, wv, i].mean()) for i in (0, 1, 2, 3, 4, 5)] for wv in range(4)])
    # heaviest workgroups (the causal blocks with most tiles)
    heavy = st[st[:, 0, 6] >= st[:, 0, 6].max() - 1]
    print(f"  heaviest workgroups ({len(heavy)}): total {heavy[:, :, 5].mean():.0f}, wait {heavy[:, :, 0].mean():.0f}, scores {heavy[:, :, 2].mean():.0f}, softmax {heavy[:, :, 3].mean():.0f}, pv {heavy[:, :, 4].mean():.0f}")
