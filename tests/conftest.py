import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "em-spec_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _torch_first():
    """torch must have initialised its HIP runtime before libemspec.so creates an engine: in the other order torch
    finds no device afterwards ("No HIP GPUs are available").  The tests that use torch tensors come after others that
    only need the engine, so the engine fixtures settle the order."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass


@pytest.fixture(scope="session")
def engine():
    """One libemspec engine on cuda:0 for the whole GPU session (no CPU fallback exists)."""
    _torch_first()
    import emspec
    e = emspec.Engine()
    yield e
    e.close()


@pytest.fixture(scope="session")
def diag_engine():
    """Engine of libemspec_diag.so (the product sources + include/emspec_debug.h), for tests that probe internals."""
    _torch_first()
    import emspec
    e = emspec.Engine(diag=True)
    yield e
    e.close()
