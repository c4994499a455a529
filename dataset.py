"""Top-level `dataset` (`from dataset import DataModule`, /root/reference/train.py:6): re-exports `transformertts_amd.dataset`."""
from transformertts_amd.dataset import *  # noqa: F401,F403
from transformertts_amd.dataset import DataModule, TransformerTTSDataset, collate_fn  # noqa: F401
