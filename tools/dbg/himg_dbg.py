import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from transformertts_amd import _lib, ops
from transformertts_amd.ops import _p, _stream
lib = _lib.load(); dev = torch.device("cuda:0")
M, N, K = 300, 768, 256
g = torch.Generator().manual_seed(1)
x = torch.randn(M, K, generator=g).to(dev); w = (torch.randn(N, K, generator=g) * K ** -0.5).to(dev); b = torch.randn(N, generator=g).to(dev)
pl = ops._planes(w, 8, N, K); xa = ops._amax(x)
img = torch.full((M, N), float("nan"), device=dev); inv = torch.full((N // 64, M), float("nan"), device=dev)
am = torch.zeros(3, 1024, device=dev)
_lib.check(lib.ttts_linear_fwd_h3d_img(_p(x), _p(pl), _p(b), _p(img), _p(inv), M, N, K, _p(xa), _p(am), 256, _stream()), "x")
torch.cuda.synchronize()
print("inv nonfinite:", (~torch.isfinite(inv)).sum().item(), "of", inv.numel())
bad = (~torch.isfinite(inv)).nonzero()
print(bad[:20].tolist())
h = img.view(torch.float16).view(M, N // 64, 2, 64)
nf = ~torch.isfinite(h.float())
print("img nonfinite:", nf.sum().item(), "of", nf.numel())
idx = nf.nonzero()
print(idx[:20].tolist())
if idx.numel():
    print("rows:", idx[:, 0].unique()[:40].tolist(), "heads:", idx[:, 1].unique().tolist(), "planes:", idx[:, 2].unique().tolist(), "d:", idx[:, 3].unique()[:64].tolist())
ref = x.double() @ w.double().t() + b.double()
y = ((h[:, :, 0].double() + h[:, :, 1].double()) * inv.t().double()[:, :, None]).reshape(M, N)
ok = torch.isfinite(y)
print("rel err on finite:", float((y[ok] - ref[ok]).norm() / ref[ok].norm()))
print("amax", am.max(dim=1).values.tolist(), [float(ref[:, i*256:(i+1)*256].abs().max()) for i in range(3)])
torch.set_printoptions(precision=4, linewidth=200)
print("y[0,:32]  ", y[0, :32].float())
print("ref[0,:32]", ref[0, :32].float())
print("y[1,:16]  ", y[1, :16].float())
print("ref[1,:16]", ref[1, :16].float())
# search: for row 0 head 0, find permutation
yy = y[0, :64].float(); rr = ref[0, :64].float()
perm = [int((rr - v).abs().argmin()) for v in yy]
print("perm row0:", perm)
rowmatch = [int(((ref[:128, :64].float() - y[r, :64].float()) ** 2).sum(1).argmin()) for r in range(0, 40)]
print("rowmatch:", rowmatch)
hb = img.view(torch.int16).view(M, N // 64, 2, 64)
for (r, hd, pl_, dd) in idx[:6].tolist():
    sc = 1.0 / float(inv[hd, r])
    want = (ref[r, hd * 64: hd * 64 + 64] * sc).float()
    print("row", r, "head", hd, "d", dd, "scale 2^", torch.log2(torch.tensor(sc)).item(), "rowmax*sc", float(want.abs().max()))
    print("  hi bits", [hex(v & 0xffff) for v in hb[r, hd, 0, dd - 2: dd + 6].tolist()])
    print("  want hi", [hex(int(v) & 0xffff) for v in want[dd - 2: dd + 6].half().view(torch.int16).tolist()])
    print("  lo bits", [hex(v & 0xffff) for v in hb[r, hd, 1, dd - 2: dd + 6].tolist()])
    print("  want   ", want[dd - 2: dd + 6].tolist())
