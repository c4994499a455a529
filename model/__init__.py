"""Top-level `model` package: the import the reference's step surface uses (`from model import TransformerTTS`,
/root/reference/lightning_module.py:7, /root/reference/model/__init__.py:1), bound to the MI355X-native implementation in
`transformertts_amd.model`.  With the repository root on `sys.path` the reference's `train.py` / `lightning_module.py`
import lines work unchanged; the sub-modules `model.model`, `model.layers`, `model.module` are aliased too."""
import sys as _sys

from transformertts_amd.model import TransformerTTS  # noqa: F401
from transformertts_amd.model import layers as _layers, model as _model, module as _module

_sys.modules[__name__ + ".model"] = _model
_sys.modules[__name__ + ".layers"] = _layers
_sys.modules[__name__ + ".module"] = _module
model, layers, module = _model, _layers, _module
