import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "bwa-mem-scale_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
