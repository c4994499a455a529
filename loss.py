"""Top-level `loss` module (`from loss import TransformerTTSLoss`, /root/reference/lightning_module.py:8): re-exports the
fused-HIP loss of `transformertts_amd.loss`."""
from transformertts_amd.loss import TransformerTTSLoss  # noqa: F401
