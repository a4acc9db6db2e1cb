"""Pin oracle/ksw_oracle.c to the reference's ksw_align2 (oracle/_ref, ksw.cpp compiled from
/root/reference): byte and 16-bit kernels, second-best tracking, start-point pass."""
import numpy as np
import pytest

from oracle import loader
from oracle.loader import KSW_XBYTE, KSW_XSTART, KSW_XSTOP, KSW_XSUBO
from util import make_local_cases

REF = loader.ref_lib()
needs_ref = pytest.mark.skipif(REF is None, reason="oracle/_ref not built (reference tree absent)")


@needs_ref
@pytest.mark.parametrize("flags", [
    KSW_XSUBO | KSW_XSTART | KSW_XBYTE | 19,      # what mem_matesw passes for 150-bp mates (bwamem_pair.cpp:214)
    KSW_XSUBO | KSW_XSTART | 19,                  # 16-bit kernel
    KSW_XSUBO | 30, KSW_XBYTE, 0, KSW_XSTART | KSW_XBYTE,
    KSW_XSTOP | KSW_XBYTE | 60,
])
def test_restatement_equals_reference(flags):
    cases = make_local_cases(250, seed=flags % 1000)
    bad = []
    for i, (q, t) in enumerate(cases):
        a = loader.ksw_align2(q, t, flags)
        b = loader.ref_ksw_align2(REF, q, t, flags)
        if a != b:
            bad.append((i, a, b, len(q), len(t)))
    assert not bad, bad[:5]


@needs_ref
def test_other_scoring_and_byte_overflow():
    opt = loader.default_sw_opt(5, 2, 3)
    opt.o_del, opt.e_del, opt.o_ins, opt.e_ins = 5, 2, 4, 1
    for q, t in make_local_cases(120, seed=77):
        for flags in (KSW_XSUBO | KSW_XSTART | 25, KSW_XSUBO | KSW_XSTART | KSW_XBYTE | 25):
            assert loader.ksw_align2(q, t, flags, opt) == loader.ref_ksw_align2(REF, q, t, flags, opt)
    # byte kernel overflow: a 300-base perfect match scores 300 > 250 -> the byte kernel reports 255
    q = np.random.default_rng(1).integers(0, 4, size=300, dtype=np.uint8)
    t = np.concatenate([np.zeros(20, np.uint8), q, np.ones(30, np.uint8)])
    a = loader.ksw_align2(q, t, KSW_XBYTE | KSW_XSUBO | 19)
    b = loader.ref_ksw_align2(REF, q, t, KSW_XBYTE | KSW_XSUBO | 19)
    assert a == b and a[0] == 255


def test_known_answers_without_reference():
    q = np.array([0, 1, 2, 3, 0, 1, 2, 3, 0, 1, 2, 3, 0, 1, 2, 3, 2, 2, 1, 0], np.uint8)
    t = np.concatenate([np.full(15, 3, np.uint8), q, np.full(10, 0, np.uint8)])
    sc, te, qe, sc2, te2, tb, qb = loader.ksw_align2(q, t, KSW_XSUBO | KSW_XSTART | KSW_XBYTE | 10)
    assert (sc, te, qe, tb, qb) == (20, 34, 19, 15, 0)
