"""CPU-only checks of the C-ABI library: it builds, loads, and exports every symbol
include/bwams.h declares (no compute without a GPU)."""
import ctypes as C
import os
import re

import pytest

from bwams import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    capi.build()
    return capi.lib()


def test_library_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "bwams.h")).read()
    declared = set(re.findall(r"\b(bwams_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(capi.SYMBOLS), declared ^ set(capi.SYMBOLS)
    for s in declared:
        assert hasattr(lib, s), s


def test_record_sizes_match_reference_layouts():
    # SMEM 40 B (FMI_search.h:85-93), SeqPair 56 B (bandedSWA.h:90-99), CP_OCC 64 B
    assert capi.SMEM_DTYPE.itemsize == 40
    assert capi.SEQPAIR_DTYPE.itemsize == 56
    assert C.sizeof(capi.SeedOpt) == 20 and C.sizeof(capi.SwOpt) == 52


def test_fails_loudly_without_a_gpu(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = C.c_void_p()
    rc = lib.bwams_index_open(b"/nonexistent/prefix", 0, C.byref(h))
    assert rc == -2                      # IO error comes first
    from util import toy
    _, idx = toy(3000)
    with pytest.raises(capi.BwamsError) as e:
        capi.Index.from_host(idx, 0)
    assert e.value.code == -1            # BWAMS_ERR_DEVICE: no fallback path exists
    assert lib.bwams_strerror(-1).decode().startswith("no usable gfx950")
