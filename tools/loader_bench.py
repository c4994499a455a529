"""Input-side rates: host collate (padded vs ragged) per core, and H2D + device padding.
Usage: python tools/loader_bench.py [B] [reps]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from oracle.collate import oracle_getitem      # noqa: E402  (tools/ is measurement infrastructure)
from oracle.synth import synth_samples         # noqa: E402
from transformertts_amd.dataset import DeviceStager, collate_fn, collate_ragged   # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    torch.set_num_threads(1)
    samples = synth_samples(B, n_mels=80, max_frames=867, seed=1)
    for s in samples:   # LJSpeech-like lengths: 95..870, mean ~566
        pass
    items = [oracle_getitem(s) for s in samples]
    frames = sum(int(it["melspec"].shape[0]) for it in items)
    Tmax = max(int(it["melspec"].shape[0]) for it in items)
    res = {"B": B, "real_frames": frames, "padded_frames": B * Tmax}
    for name, fn in (("collate_fn", collate_fn), ("collate_ragged", collate_ragged)):
        fn(items)
        t = time.perf_counter()
        for _ in range(reps):
            out = fn(items)
        dt = (time.perf_counter() - t) / reps
        res[name + "_ms"] = round(dt * 1e3, 3)
        res[name + "_frames_per_s_core"] = round(frames / dt)
    if torch.cuda.is_available():
        st = DeviceStager("cuda")
        for name, fn in (("padded", collate_fn), ("ragged", collate_ragged)):
            hb = fn(items)
            if name == "padded":
                hb = {k: (v.pin_memory() if isinstance(v, torch.Tensor) else v) for k, v in hb.items()}
            DeviceStager.wait(st.stage(hb))
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(reps):
                d = DeviceStager.wait(st.stage(hb))
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t) / reps
            res[f"stage_{name}_ms"] = round(dt * 1e3, 3)
            res[f"stage_{name}_frames_per_s"] = round(frames / dt)
    print(res)


if __name__ == "__main__":
    main()
