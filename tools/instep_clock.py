#!/usr/bin/env python3
"""The clock wgrad_dma_kernel runs at INSIDE the training step (s_memtime / s_memrealtime of its last launch): runs bench.py's
default step with a library built with -DTTTS_WG_STAMPS (TTTS_LIB=<so>), then reads the stamps.  Compare tools/wgrad_stamps.py
(the same kernel back to back: 1.4 GHz; with idle gaps: 1.9 GHz)."""
import ctypes, os, runpy, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.argv = ["bench.py", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-probe", "--sustain", "0", "--no-alignments-figure"]
try:
    runpy.run_path(os.path.join(root, "bench.py"), run_name="__main__")
except SystemExit:
    pass
from transformertts_amd import _lib
raw = ctypes.CDLL(_lib.LIB_PATH)
n = 2048 * 8 * 8
buf = (ctypes.c_ulonglong * n)()
raw.ttts_dbg_wg_read_stamps(buf, ctypes.c_size_t(n))
st = np.frombuffer(buf, dtype=np.uint64).reshape(2048, 8, 8).astype(np.float64)
st = st[st[:, 0, 3] > 0]
tot = st[:, :, 3].mean(); ghz = st[:, :, 3].sum() / st[:, :, 7].sum() / 10.0
print(f"last wgrad_dma launch of the step: {len(st)} workgroups, {tot:.0f} ticks per wave = {tot / ghz / 1e3:.1f} us at {ghz:.2f} GHz", file=sys.stderr)
