"""Pins of the FM-index pieces of the reference that compile without the un-vendored safestringlib (oracle/ref_harness_fmi.cpp ->
oracle/_ref/libref_fmi.so): the record layouts of src/FMI_search.h (static_asserts in the harness, sizes here), its GET_OCC macro
over the blocks our builders write, and sais.h — the suffix sorter build_index calls (src/FMI_search.cpp:833-840) — against the
suffix array behind the host builder (which the GPU builder is tested against, tests/test_gpu_build.py) and, through the
sampled-SA arrays and the BWT blocks, against the index records themselves."""
import numpy as np
import pytest

from bwams import fmindex, simulate
from oracle import loader

ref = loader.ref_fmi_lib()
pytestmark = pytest.mark.skipif(ref is None, reason="oracle/_ref/libref_fmi.so not built (reference tree absent)")


def test_record_sizes():
    assert [ref.ref_fmi_sizes(i) for i in range(4)] == [64, 40, 128, 16]
    assert loader.SMEM_DTYPE.itemsize == 40


@pytest.mark.parametrize("n,seed,rep", [(3000, 1, 0.3), (20000, 2, 0.15), (65, 3, 0.0), (1, 4, 0.0)])
def test_get_occ_macro_over_our_blocks(n, seed, rep):
    g = simulate.make_genome(n, seed=seed, repeat_frac=rep, repeat_len=100, n_families=2)
    idx = fmindex.build_fmindex(g)
    o = loader.OracleFMI(idx)
    cp = np.ascontiguousarray(idx.cp_occ)
    L = idx.ref_seq_len
    rng = np.random.default_rng(seed)
    pos = np.unique(np.concatenate([rng.integers(0, L + 1, size=400), np.arange(0, min(L + 1, 200)), [L, L - 1, idx.sentinel_index, idx.sentinel_index + 1]]))
    for p in pos:
        for c in range(4):
            assert ref.ref_get_occ(cp.ctypes.data, int(p), c) == o.occ(int(p), c), (p, c)


@pytest.mark.parametrize("n,seed,rep", [(2000, 5, 0.4), (30000, 6, 0.2), (50000, 7, 0.0)])
def test_sais_suffix_array_is_the_index(n, seed, rep):
    """saisxx over the letters of fw + rc (what build_index sorts) gives the suffix array whose samples, BWT and counts the
    index holds: SA[8 i] = sa_ms_byte << 32 | sa_ls_word, BWT[j] = text[SA[j] - 1] as the one-hot strings say, the sentinel row."""
    g = simulate.make_genome(n, seed=seed, repeat_frac=rep, repeat_len=150, n_families=3)
    g[100:160] = 0                                    # a homopolymer run
    idx = fmindex.build_fmindex(g)
    text = fmindex.fw_rc_text(g)
    letters = np.frombuffer(b"ACGT", np.uint8)[text].tobytes()
    N = len(text)
    sa = np.zeros(N + 1, np.int64)
    assert ref.ref_sais(letters, N, sa.ctypes.data) == 0
    assert sa[0] == N and np.array_equal(np.sort(sa), np.arange(N + 1))
    # sampled suffix array
    ms = np.asarray(idx.sa_ms_byte).astype(np.int64) & 0xff
    ls = np.asarray(idx.sa_ls_word).astype(np.int64) & 0xffffffff
    samp = (ms << 32) | ls
    want = sa[::8]
    assert np.array_equal(samp[:len(want)], want)
    # sentinel row and BWT (one-hot strings of the CP_OCC blocks)
    sent = int(np.flatnonzero(sa == 0)[0])
    assert sent == idx.sentinel_index
    bwt = text[np.where(sa > 0, sa - 1, 0)]
    cp = np.asarray(idx.cp_occ).astype(np.uint64)
    rows = np.arange(N + 1)
    blk, bit = rows >> 6, np.uint64(63) - (rows & 63).astype(np.uint64)
    for c in range(4):
        hot = ((cp[blk, 4 + c] >> bit) & np.uint64(1)).astype(bool)
        expect = (bwt == c) & (rows != sent)
        assert np.array_equal(hot, expect), c
    # counts: the loader's count[] (+1) = rows of suffixes starting with a smaller base
    first = text[sa[1:]]
    for c in range(4):
        assert idx.count[c] == 1 + int((first < c).sum())


def test_bseq1_t_of_the_reference_is_the_record_the_host_layer_mirrors():
    """bwa.h compiles on its own with the `scale` build's feature macros (oracle/ref_harness_bwa.cpp -> oracle/_ref/libref_bwa.so):
    the reference's bseq1_t — 64 bytes, sam at 48, perfect at 56 — against the record the Python harness builds for
    mem_process_seqs() (capi.BSEQ1_DTYPE) and the numbers host/bwamem_hip.h asserts for its mirror."""
    import ctypes as C
    import os
    import re
    from bwams import capi
    so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libref_bwa.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref/libref_bwa.so not built (reference tree absent)")
    lib = C.CDLL(so)
    out = (C.c_int * 13)()
    lib.ref_bseq1_layout(out)
    out = list(out)
    d = capi.BSEQ1_DTYPE
    assert out[0] == d.itemsize == 64
    for k, f in enumerate(("l_seq", "id", "strbuf", "name", "comment", "seq", "qual", "sam", "perfect")):
        assert out[1 + k] == d.fields[f][1], f
    assert out[10] == 8 and out[11] == 0 and out[12] == 4                 # bseq1_perfect_t: {flags, location} over `exist`
    # the mirror's static_assert carries the same numbers
    hdr = open(os.path.join(os.path.dirname(so), "..", "..", "bwa-mem-scale_amd", "host", "bwamem_hip.h")).read()
    m = re.search(r"sizeof\(bseq1_t\) == (\d+) && offsetof\(bseq1_t, name\) == (\d+) && offsetof\(bseq1_t, seq\) == (\d+) && offsetof\(bseq1_t, sam\) == (\d+) &&\s*offsetof\(bseq1_t, perfect\) == (\d+)", hdr)
    assert m and [int(x) for x in m.groups()] == [out[0], out[4], out[6], out[8], out[9]]
