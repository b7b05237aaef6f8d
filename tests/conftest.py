import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def snb():
    """The product package (directory name has hyphens, so it is imported through importlib)."""
    try:   # let torch create its HIP context first: tests that use torch for device buffers run after many engine contexts
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass
    return importlib.import_module("openmm-nonbonded-slicing_amd")


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle front-end (test infrastructure; builds oracle/libsnb_oracle.so on first use)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc
    orc.lib()
    return orc
