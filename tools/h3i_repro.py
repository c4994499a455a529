#!/usr/bin/env python3
"""Race screen for gemm_h3i_kernel: the same launch repeated must give the same bits, for every epilogue kind and under load
from a second stream (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformertts_amd import _lib, ops
from transformertts_amd.ops import _p, _stream

lib = _lib.load()
dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 55680
torch.manual_seed(0)


def image_of(x):
    m, k = x.shape
    img = torch.empty(m * k * 2, dtype=torch.int16, device=dev); inv = torch.empty(m, device=dev)
    _lib.check(lib.ttts_act_image(_p(x), _p(img), _p(inv), m, k, _stream()), "act_image")
    return img, inv


bad = 0
for (N, K, kind) in [(1024, 256, "gate_amax"), (1024, 256, "gate"), (256, 256, "gate_amax"), (256, 256, "gate"), (1024, 256, "gate_res"),
                     (1024, 256, "bias_amax"), (1024, 256, "res")]:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * K ** -0.5; b = torch.randn(N, device=dev)
    res = torch.randn(M, N, device=dev); h = torch.relu(torch.randn(M, N, device=dev))
    img, inv = image_of(x)
    img2, inv2 = image_of(x)
    assert torch.equal(img, img2) and torch.equal(inv, inv2)
    pl = ops._planes(w, 8, N, K).clone()
    plt = ops._planes(w.t().contiguous(), 9, N, K).clone()        # (K_lin=N... ) data-gradient form: rows = output width N
    outs = []
    for rep in range(12):
        y = torch.full((M, N), float("nan"), device=dev)
        am = torch.zeros(ops.AMAX_SLOTS, device=dev)
        if kind == "relu_drop_amax":
            rc = lib.ttts_linear_fwd_h3i(_p(img), _p(inv), _p(pl), _p(b), None, _p(y), M, N, K, 1, 0.1, 77, None, _p(am), _stream())
        elif kind == "bias_amax":
            rc = lib.ttts_linear_fwd_h3i(_p(img), _p(inv), _p(pl), _p(b), None, _p(y), M, N, K, 0, 0.0, 0, None, _p(am), _stream())
        elif kind == "res_drop":
            rc = lib.ttts_linear_fwd_h3i(_p(img), _p(inv), _p(pl), _p(b), _p(res), _p(y), M, N, K, 0, 0.1, 78, None, None, _stream())
        elif kind == "res":
            rc = lib.ttts_linear_fwd_h3i(_p(img), _p(inv), _p(pl), _p(b), _p(res), _p(y), M, N, K, 0, 0.0, 0, None, None, _stream())
        elif kind == "gate_amax":     # dx[M, N] = dy[M, K] . wT ; here the "weight" planes are those of a (K -> N) data gradient
            rc = lib.ttts_linear_bwd_data_h3i(_p(img), _p(inv), _p(plt), None, _p(y), M, K, N, _p(h), 1.0 / 0.9, _p(am), _stream())
        elif kind == "gate":
            rc = lib.ttts_linear_bwd_data_h3i(_p(img), _p(inv), _p(plt), None, _p(y), M, K, N, _p(h), 1.0 / 0.9, None, _stream())
        else:
            rc = lib.ttts_linear_bwd_data_h3i(_p(img), _p(inv), _p(plt), _p(res), _p(y), M, K, N, _p(h), 1.0 / 0.9, None, _stream())
        assert rc == 0, _lib.last_error()
        outs.append((y, am))
    torch.cuda.synchronize()
    nb = sum(0 if (torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][1], o[1])) else 1 for o in outs[1:])
    nan = int(torch.isnan(outs[0][0]).sum())
    if kind.startswith("gate"):
        rows = slice(0, 4096)
        ref = (x[rows].double() @ w.double().t()) * (h[rows] > 0).double() / 0.9
        if kind == "gate_res":
            ref = ref + res[rows].double()
        errs = [float((o[0][rows].double() - ref).abs().max()) for o in outs]
        print("   max abs error vs fp64 on rows 0..4095 per repeat:", " ".join(f"{e:.1e}" for e in errs))
    if nb:
        d = (outs[0][0] != [o for o in outs[1:] if not torch.equal(outs[0][0], o[0])][0][0]).nonzero()
        print(f"N={N} K={K} {kind}: {nb} of 11 repeats differ; {d.shape[0]} elements, first {d[:4].tolist()}, rows {sorted(set((d[:, 0] // 128).tolist()))[:8]} (128-row panels)")
    else:
        print(f"N={N} K={K} {kind}: 12 launches identical (nan {nan})")
    bad += nb
print("differences:", bad)
