"""Masks the kernels do not derive from lengths -- a key-padding mask with holes, a `tgt_mask` that is not the causal mask, a
`memory_mask`, an encoder `mask`: arguments the reference's layers accept (model/layers.py:29-74 on torch's
nn.TransformerDecoderLayer / nn.TransformerEncoder) and its model never passes.  The reference here is torch's own classes in
fp64 (the reference's layers ARE those classes plus the returned alignments), loaded with the same state dict.  Gate 1e-4
(north_star), measured ~1e-6."""
import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu
GATE = 1e-4


def _pair(kind, d, h, ff, layers=2):
    from transformertts_amd.model import layers as L
    torch.manual_seed(11)
    if kind == "dec":
        ours = L.TransformerDecoderLayer(d, h, ff, dropout=0.0)
        ref = torch.nn.TransformerDecoderLayer(d, h, ff, dropout=0.0, batch_first=True)
    else:
        ours = L.TransformerEncoder(L.TransformerEncoderLayer(d, h, ff, dropout=0.0), layers)
        for p in ours.parameters():                 # (deep copies of one layer: make the layers differ)
            torch.nn.init.normal_(p, std=0.08) if p.dim() > 1 else torch.nn.init.normal_(p, mean=0.5, std=0.2)
        ref = torch.nn.TransformerEncoder(torch.nn.TransformerEncoderLayer(d, h, ff, dropout=0.0, batch_first=True), layers,
                                          enable_nested_tensor=False)
    ref.load_state_dict(ours.state_dict(), strict=True)
    return ours.cuda().train(), ref.double().train()       # train mode, dropout 0: torch's inference fast path stays out of it


def _holes(B, T, lens, holes, gen):
    """key-padding mask: keys past lens[b] dead plus `holes` random dead keys among the live ones (never key 0)"""
    kpm = torch.arange(T)[None, :] >= torch.tensor(lens)[:, None]
    for b in range(B):
        idx = torch.randperm(lens[b] - 1, generator=gen)[:holes] + 1
        kpm[b, idx] = True
    return kpm


@pytest.mark.parametrize("float_masks", [False, True])
def test_decoder_layer_with_arbitrary_masks_matches_torch(float_masks):
    d, h, ff, B, Tq, Tk = 128, 2, 256, 3, 37, 29
    ours, ref = _pair("dec", d, h, ff)
    g = torch.Generator().manual_seed(5)
    tgt = torch.randn(B, Tq, d, generator=g)
    mem = torch.randn(B, Tk, d, generator=g)
    tgt_kpm = _holes(B, Tq, [37, 30, 21], 3, g)
    mem_kpm = _holes(B, Tk, [29, 17, 8], 2, g)
    i, j = torch.arange(Tq)[:, None], torch.arange(Tq)[None, :]
    tgt_mask = (j > i) | (i - j > 6)                                  # a causal BAND: not the causal mask
    mem_mask = torch.rand(Tq, Tk, generator=g) < 0.3
    mem_mask[:, 0] = False                                            # (every query keeps a live key: torch gives NaN otherwise)
    if float_masks:                                                   # float masks are ADDED to the scores
        tgt_mask = torch.randn(Tq, Tq, generator=g).masked_fill(tgt_mask, float("-inf"))
        mem_mask = torch.randn(B * h, Tq, Tk, generator=g).masked_fill(mem_mask[None].expand(B * h, -1, -1), float("-inf"))
    t64, m64 = tgt.double().requires_grad_(True), mem.double().requires_grad_(True)
    y_ref = ref(t64, m64, tgt_mask=tgt_mask if tgt_mask.dtype == torch.bool else tgt_mask.double(),
                memory_mask=mem_mask if mem_mask.dtype == torch.bool else mem_mask.double(),
                tgt_key_padding_mask=tgt_kpm, memory_key_padding_mask=mem_kpm)
    x1 = ref.norm1(t64 + ref._sa_block(t64, tgt_mask if tgt_mask.dtype == torch.bool else tgt_mask.double(), tgt_kpm))
    _, w_ref = ref.multihead_attn(x1, m64, m64, attn_mask=mem_mask if mem_mask.dtype == torch.bool else mem_mask.double(),
                                  key_padding_mask=mem_kpm, need_weights=True, average_attn_weights=False)
    dy = torch.randn(B, Tq, d, generator=g)
    y_ref.backward(dy.double())

    tc, mc = tgt.cuda().requires_grad_(True), mem.cuda().requires_grad_(True)
    y, w = ours(tc, mc, tgt_mask=tgt_mask.cuda(), memory_mask=mem_mask.cuda(), tgt_key_padding_mask=tgt_kpm.cuda(),
                memory_key_padding_mask=mem_kpm.cuda(), tgt_is_causal=False)
    y.backward(dy.cuda())
    assert rel_l2(y.detach().cpu(), y_ref.detach()) < GATE
    assert rel_l2(w.detach().cpu(), w_ref.detach()) < GATE
    assert rel_l2(tc.grad.cpu(), t64.grad) < GATE and rel_l2(mc.grad.cpu(), m64.grad) < GATE
    for (n, p), (_, q) in zip(ours.named_parameters(), ref.named_parameters()):
        assert rel_l2(p.grad.cpu(), q.grad) < GATE, n


def test_encoder_with_mask_and_holes_matches_torch():
    d, h, ff, B, T = 128, 2, 192, 3, 41
    ours, ref = _pair("enc", d, h, ff)
    g = torch.Generator().manual_seed(6)
    x = torch.randn(B, T, d, generator=g)
    kpm = _holes(B, T, [41, 33, 12], 4, g)
    i, j = torch.arange(T)[:, None], torch.arange(T)[None, :]
    mask = (i - j).abs() > 9                                          # a band around the diagonal
    x64 = x.double().requires_grad_(True)
    y_ref = ref(x64, mask=mask, src_key_padding_mask=kpm)
    dy = torch.randn(B, T, d, generator=g)
    y_ref.backward(dy.double())
    xc = x.cuda().requires_grad_(True)
    y = ours(xc, mask=mask.cuda(), src_key_padding_mask=kpm.cuda())
    y.backward(dy.cuda())
    assert rel_l2(y.detach().cpu(), y_ref.detach()) < GATE
    assert rel_l2(xc.grad.cpu(), x64.grad) < GATE
    for (n, p), (_, q) in zip(ours.named_parameters(), ref.named_parameters()):
        assert rel_l2(p.grad.cpu(), q.grad) < GATE, n


def test_prefix_and_causal_mask_tensors_take_the_kernels():
    """mask TENSORS that say what lengths / the causal flag say run on the same kernels: same bits as the lengths call"""
    d, h, ff, B, Tq, Tk = 128, 2, 256, 3, 50, 23
    ours, _ = _pair("dec", d, h, ff)
    ours.eval()
    g = torch.Generator().manual_seed(7)
    tgt, mem = torch.randn(B, Tq, d, generator=g).cuda(), torch.randn(B, Tk, d, generator=g).cuda()
    tl, ml = torch.tensor([50, 31, 7]).cuda(), torch.tensor([23, 23, 5]).cuda()
    tgt_kpm = torch.arange(Tq).cuda()[None, :] >= tl[:, None]
    mem_kpm = torch.arange(Tk).cuda()[None, :] >= ml[:, None]
    causal_b = torch.triu(torch.ones(Tq, Tq, dtype=torch.bool, device="cuda"), 1)
    causal_f = torch.zeros(Tq, Tq, device="cuda").masked_fill(causal_b, float("-inf"))
    with torch.no_grad():
        y0, w0 = ours(tgt, mem, tgt_is_causal=True, tgt_lens=tl, memory_lens=ml)
        for m in (causal_b, causal_f):
            y1, w1 = ours(tgt, mem, tgt_mask=m, tgt_key_padding_mask=tgt_kpm, memory_key_padding_mask=mem_kpm)
            assert torch.equal(y0, y1) and torch.equal(w0, w1)
