"""oracle/ -- TEST INFRASTRUCTURE ONLY (never imported by the product path).

A stock-PyTorch **CPU** restatement of the teacher-forced Transformer-TTS hot
path of Orca0917/TransformerTTS (`model/{model,layers,module}.py`, `loss.py`,
`utils/util.py`, `lightning_module.py::training_step`).  It exists to check the
hand-written HIP path in `transformertts_amd/` and to be timed as the
`cpu_baseline` ("port") leg of `bench.py`.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may
import this package.  `transformertts_amd/` must never import it; the product
path raises if the HIP library is missing instead of falling back to anything
in here.

Parity pin: the reference repo ships no tests / golden vectors for this path
(SURVEY.md section 4), and the arithmetic itself lives in un-vendored `torch`
(readme pins 2.2.0; this image has 2.10.0+rocm7.0).  The oracle is therefore
pinned by fixtures generated in the build container from the *live import of the
real reference* (`tests/golden/make_golden.py`, outputs committed under
`tests/golden/*.npz`); `tests/test_oracle_golden.py` checks the restatement
against those fixtures on every run.
"""

from .spec import CONFIGS, model_config, state_spec, fill_state  # noqa: F401
from .ref_model import oracle_forward, oracle_loss, oracle_training_step, oracle_inference, relu_gates  # noqa: F401
from .synth import synth_batch  # noqa: F401
