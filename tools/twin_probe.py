#!/usr/bin/env python3
"""What would running the encoder of both forwards of a step as ONE batch of 2 B buy?  (the no-grad forward and the grad forward
of training_step encode the SAME phonemes, lightning_module.py:53-59,77.)  Replays captured graphs of `encode` (no grad, train
mode): two passes at B against one pass at 2 B.  usage: python3 tools/twin_probe.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformertts_amd import ops
from transformertts_amd.model import TransformerTTS
from transformertts_amd.workload import model_config

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
cfg = model_config("base")
m = TransformerTTS(**cfg, device="cuda").to(dev).train()
g = torch.Generator().manual_seed(1)


def graph_of(nb, passes):
    ph = torch.randint(0, 100, (nb, 100), generator=g).to(dev)
    lens = torch.full((nb,), 100, dtype=torch.int64, device=dev)
    with torch.no_grad():
        for _ in range(2):
            for _ in range(passes):
                m.encode(ph, lens)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(passes):
                m.encode(ph, lens)
    return gr


for name, nb, passes in (("two passes at B", B, 2), ("one pass at 2 B", 2 * B, 1)):
    gr = graph_of(nb, passes)
    for _ in range(3):
        gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    print(f"B = {B}: encoder forward, {name}: {e0.elapsed_time(e1) / 50 * 1e3:7.1f} us")
