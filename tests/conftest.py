import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def rel_l2(a, b):
    import inspect
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    den = b.norm().item()
    err = (a - b).norm().item() / (den if den > 0 else 1.0)
    if os.environ.get("TTTS_ERRLOG"):      # development aid: log every measured error with its call site
        fr = inspect.stack()[1]
        with open(os.environ["TTTS_ERRLOG"], "a") as f:
            f.write(f"{err:.3e} {os.path.basename(fr.filename)}:{fr.lineno} {fr.function} {(fr.code_context or [''])[0].strip()[:90]}\n")
    return err


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(REPO, "tests", "golden")
