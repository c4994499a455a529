import sys, os, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from test_hip_graph import _setup, _state, _same
from transformertts_amd.step import TrainStep
from transformertts_amd import ops
from transformertts_amd.workload import synth_batch
mode = sys.argv[1] if len(sys.argv) > 1 else "AB"
if "L" in mode:
    cfg, lm, opt, sch = _setup("tiny", 3, 0)
    ts = TrainStep(lm, opt, sch, graph=True, seed=2, lattice=(8, 16) if "P" in mode else None)
    for i, (Tp, Tm) in enumerate([(12, 40), (10, 35), (14, 44), (9, 33), (11, 47), (13, 41)] if "V" in mode else [(12, 40)] * 6):
        b = {k: v.to("cuda") for k, v in synth_batch(3, Tp, Tm, cfg["n_mels"], cfg["n_phon"], ragged=True, seed=i).items()}
        ts(b)
    torch.cuda.synchronize()
    print("lattice phase done, graphs", ts.n_graphs, flush=True)
    if "K" not in mode:
        del ts, lm, opt, sch
cfg, lmA, optA, schA = _setup("tiny", 3, 0)
_, lmB, optB, schB = _setup("tiny", 4, 0)
_, lmR, optR, schR = _setup("tiny", 4, 0)
batch = {k: v.to("cuda") for k, v in synth_batch(3, 12, 40, cfg["n_mels"], cfg["n_phon"], ragged=True, seed=8).items()}
tsA = TrainStep(lmA, optA, schA, batch, graph=("A" in mode), seed=1)
tsB = TrainStep(lmB, optB, schB, batch, graph=("B" in mode), seed=2)
tsR = TrainStep(lmR, optR, schR, batch, graph=False, seed=2)
torch.cuda.memory._record_memory_history(max_entries=200000)
print("pool handles", torch.cuda.graph_pool_handle(), torch.cuda.graph_pool_handle(), flush=True)
if "N" in mode:
    import transformertts_amd.step as S
    _orig = torch.cuda.graph
    class _NoPool(_orig):
        def __init__(self, g, pool=None, **kw):
            super().__init__(g, pool=None, **kw)
    S.torch.cuda.graph = _NoPool
for i in range(6):
    if i == 3:
        snap = torch.cuda.memory_snapshot()
        pools = {}
        for seg in snap:
            pools.setdefault(tuple(seg.get("segment_pool_id", (0, 0))), []).append((seg["address"], seg["total_size"]))
        print("pools:", {k: (len(v), sum(x[1] for x in v)) for k, v in pools.items()}, flush=True)
        print("tsA pool", tsA._pool, "tsB pool", tsB._pool, flush=True)
        def where(t):
            a = t.data_ptr()
            for k, v in pools.items():
                for base, size in v:
                    if base <= a < base + size:
                        return k
            return None
        print("lossB in", where(tsB._cur.losses["full"]), "lossA in", where(tsA._cur.losses["full"]) if tsA._cur.graphs else None,
              "arena in", where(ops._amax_arenas[torch.device("cuda", 0)].buf), "B planes.dev in", where(tsB._planes.dev), flush=True)
    if i == 3 and "D" in mode:
        if "1" not in mode:
            del tsA, lmA, optA, schA
            tsA = None
        if "2" not in mode:
            gc.collect()
        if "S" in mode:
            snap = torch.cuda.memory._snapshot()
            for seg in snap["segments"]:
                pid = tuple(seg.get("segment_pool_id", (0, 0)))
                if pid == (0, 0):
                    continue
                for blk in seg["blocks"]:
                    if blk["state"] != "inactive":
                        fr = [f"{f['filename'].split('/')[-1]}:{f['line']}:{f['name']}" for f in blk.get("frames", [])][:12]
                        print("LIVE in pool", pid, blk["state"], blk["size"], hex(blk.get("address", 0)), " <- ".join(fr), flush=True)
        if "3" not in mode:
            torch.cuda.empty_cache()
    la = tsA() if (tsA is not None and not ("D" in mode and i >= 3)) else None
    if i == 2 and "W" in mode:
        import traceback
        snapW = torch.cuda.memory_snapshot()
        rangesA = [(sg["address"], sg["address"] + sg["total_size"]) for sg in snapW if tuple(sg.get("segment_pool_id", (0, 0))) == tuple(tsA._pool)]
        print("A pool ranges", len(rangesA), flush=True)
        _orig_p = ops._p
        seen = set()
        def _p_watch(t):
            if t is not None:
                a = t.data_ptr()
                for lo, hi in rangesA:
                    if lo <= a < hi:
                        key = (a, tuple(t.shape))
                        if key not in seen:
                            seen.add(key)
                            fr = traceback.extract_stack(limit=6)[:-1]
                            print("B TOUCHES A-POOL:", hex(a), tuple(t.shape), " <- ".join(f"{f.name}:{f.lineno}" for f in fr), flush=True)
            return _orig_p(t)
        ops._p = _p_watch
    lb = tsB(); lr = tsR()
    torch.cuda.synchronize()
    if "X" in mode and i >= 3:
        rows = []
        for (n, pb), (_, pr) in zip(lmB.model.named_parameters(), lmR.model.named_parameters()):
            d = (pb.detach() - pr.detach()).abs().max().item()
            gb, gr = pb.grad, pr.grad
            dg = (gb - gr).abs().max().item() if gb is not None and gr is not None else -1
            if d > 0 or dg > 0:
                rows.append((n, d, dg, float(pr.grad.abs().max()) if pr.grad is not None else 0))
        print("step", i, "differing params:", len(rows), "of", len(list(lmB.model.parameters())), flush=True)
        for r in rows[:40]:
            print("   ", r, flush=True)
        for (n, bb), (_, br) in zip(lmB.model.named_buffers(), lmR.model.named_buffers()):
            if bb.dtype.is_floating_point and not torch.equal(bb, br):
                print("    buffer", n, (bb - br).abs().max().item(), flush=True)
    print(i, mode, "lossA", None if la is None else float(la), "lossB", float(lb), "lossR", float(lr), "same", _same(_state(lmB, optB), _state(lmR, optR)), flush=True)
