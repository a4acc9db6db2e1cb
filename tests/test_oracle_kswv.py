"""The reference's own batched mate-rescue kernel — kswv::getScores8 / getScores16 (src/kswv.cpp, compiled with
-mavx512bw from where it lies into oracle/_ref/libref_kswv.so), driven in the order of mem_sam_pe_batch
(src/bwamem_pair.cpp:903-971) — against ksw_align2 (the reference's ksw.cpp object) and against the restatement in
oracle/ksw_oracle.c, on tasks shaped like mate rescue.  The AVX512 build of the reference runs kswv, the other builds
run ksw_align2 per task; SURVEY.md §8c measured 0 differing SAM lines between them."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import loader
from oracle.loader import KSW_XBYTE, KSW_XSTART, KSW_XSUBO
from util import make_local_cases

PATH = os.path.join(os.path.dirname(loader.__file__), "_ref", "libref_kswv.so")
HAVE = os.path.exists(PATH) and "avx512bw" in loader._cpu_flags()
pytestmark = pytest.mark.skipif(not HAVE, reason="oracle/_ref/libref_kswv.so not built or the host has no AVX512BW")


def _kswv(cases, xtras, a=1, b=4, gaps=(6, 1, 6, 1)):
    L = C.CDLL(PATH)
    assert L.ref_kswv_available() == 1
    n = len(cases)
    order = [i for i in range(n) if xtras[i] & KSW_XBYTE] + [i for i in range(n) if not (xtras[i] & KSW_XBYTE)]
    n8 = sum(1 for i in range(n) if xtras[i] & KSW_XBYTE)
    pairs = np.zeros(n, dtype=loader.SEQPAIR_DTYPE)
    ro = qo = 0
    refs, qers = [], []
    for k, i in enumerate(order):
        q, t = cases[i]
        pairs[k]["idr"], pairs[k]["idq"], pairs[k]["len1"], pairs[k]["len2"] = ro, qo, len(t), len(q)
        pairs[k]["h0"], pairs[k]["regid"], pairs[k]["id"] = xtras[i], i, k
        refs.append(t); qers.append(q)
        ro += len(t); qo += len(q)
    ref = np.concatenate(refs + [np.zeros(4096, np.uint8)])
    qer = np.concatenate(qers + [np.zeros(4096, np.uint8)])
    aln = np.zeros((n, 7), dtype=np.int32)
    L.ref_kswv_batch(gaps[0], gaps[1], gaps[2], gaps[3], a, b, pairs.ctypes.data_as(C.c_void_p), n8, n - n8,
                     ref.ctypes.data_as(C.c_void_p), qer.ctypes.data_as(C.c_void_p), 2048, 512, aln.ctypes.data_as(C.c_void_p))
    return aln


def test_kswv_equals_ksw_align2_and_the_restatement():
    cases = make_local_cases(700, seed=41)
    # what mem_matesw passes (bwamem_pair.cpp:214): the byte kernel only while l_ms * a < 250
    xtras = [KSW_XSUBO | KSW_XSTART | (KSW_XBYTE if len(q) < 250 else 0) | 19 for q, _ in cases]
    for i in range(0, len(cases), 5):                                   # a share through the 16-bit kernel as well
        xtras[i] &= ~KSW_XBYTE
    got = _kswv(cases, xtras)
    REF = loader.ref_lib()
    diff_ref, diff_orc = [], []
    for i, (q, t) in enumerate(cases):
        want = loader.ksw_align2(q, t, xtras[i])
        if tuple(int(x) for x in got[i]) != tuple(want):
            diff_orc.append((i, tuple(got[i]), want))
        if REF is not None and tuple(int(x) for x in got[i]) != tuple(loader.ref_ksw_align2(REF, q, t, xtras[i])):
            diff_ref.append(i)
    assert not diff_orc, diff_orc[:5]
    assert not diff_ref, diff_ref[:5]
