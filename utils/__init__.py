"""Top-level `utils` package (`from utils.util import ...` / `from utils.plot import ...`,
/root/reference/lightning_module.py:9-21): aliases `transformertts_amd.utils`."""
import sys as _sys

from transformertts_amd.utils import plot as _plot
from transformertts_amd.utils import util as _util

_sys.modules[__name__ + ".util"] = _util
util = _util
_sys.modules[__name__ + ".plot"] = _plot
plot = _plot
