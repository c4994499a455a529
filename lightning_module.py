"""Top-level `lightning_module` (`from lightning_module import LightningModule`, /root/reference/train.py:7): re-exports the
training_step surface of `transformertts_amd.lightning_module`."""
from transformertts_amd.lightning_module import LightningModule  # noqa: F401
