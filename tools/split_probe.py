#!/usr/bin/env python3
"""Which weights does a captured step still re-split ONE AT A TIME (ttts_weight_split inside the capture) instead of through the
batched refresh?  Runs bench.py's step in-process with the entry point wrapped.  usage: python3 tools/split_probe.py [bench flags]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformertts_amd import _lib

lib = _lib.load()
orig = lib.ttts_weight_split


def logged(w, planes, rows, cols, mode, c2, taps, stream):
    print(f"[split_probe] ttts_weight_split rows {rows} cols {cols} mode {mode} c2 {c2} taps {taps} capturing "
          f"{torch.cuda.is_current_stream_capturing()}", file=sys.stderr)
    return orig(w, planes, rows, cols, mode, c2, taps, stream)


lib.ttts_weight_split = logged
import bench
sys.argv = ["bench.py", "--steps", "2", "--warmup", "5", "--no-cpu-baseline", "--no-probe", "--sustain", "0", "--no-alignments-figure"] + sys.argv[1:]
bench.main()
