import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "bwa-mem-scale_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True)
def _library_switches_follow_the_environment():
    """The library reads its debugging aids / A-B switches (BWAMS_*) from the environment once.  A test that changes one calls
    capi.debug_reload(); after every test — monkeypatch has restored the environment by then or does so right after, so both before
    and after — the library re-reads it, and no test inherits another's switches."""
    def reload():
        try:
            from bwams import capi
            if capi._lib is not None:
                capi.debug_reload()
        except Exception:
            pass
    reload()
    yield
    reload()


@pytest.fixture(scope="session", autouse=True)
def _torch_hip_runtime_first():
    """PyTorch-ROCm bundles its own HIP runtime, and on this image it only finds the GPU when it
    initialises BEFORE the /opt/rocm runtime that libbwams.so links (the other order works, which
    is the order bench.py uses).  A few GPU tests build their inputs with torch, so on a GPU box
    initialise torch's runtime once, up front.  Nothing here touches the product library."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:       # no torch / no GPU: the CPU suite does not need it
        pass
    yield
