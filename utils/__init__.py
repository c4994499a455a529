"""Top-level `utils` package (`from utils.util import ...`, /root/reference/lightning_module.py:9-16): aliases
`transformertts_amd.utils`."""
import sys as _sys

from transformertts_amd.utils import util as _util

_sys.modules[__name__ + ".util"] = _util
util = _util
