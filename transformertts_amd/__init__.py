"""transformertts_amd -- MI355X-native (gfx950) teacher-forced Transformer-TTS forward/backward.

Drop-in for the reference's `model` package surface:
    from transformertts_amd.model import TransformerTTS
Host code is Python on PyTorch-ROCm (memory, streams, autograd graph, torch.distributed); every
arithmetic step of the path runs in hand-written HIP kernels behind the C ABI of include/ttts_hip.h.
"""
__all__ = ["model", "ops"]
