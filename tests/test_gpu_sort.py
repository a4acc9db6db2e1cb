"""The region sorts of mem_sort_dedup_patch as the wave tier runs them (bwams_debug_sort): ksort.h's introsort is not
stable, so on tied keys only the same sequence of comparisons and swaps gives the reference's order.  Checked against
the reference's own ks_introsort (oracle/_ref/libref_chain.so when present, else the restatement pinned to it) on inputs
with heavy ties, including every small size (the six-record case of profiles/r01_notes.md item 20 among them), sorted /
reversed / organ-pipe inputs that reach the comb-sort depth fallback, and sizes up to the LDS limit."""
import numpy as np
import pytest

from bwams import capi
from oracle import loader
from util import toy

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ix():
    _, idx = toy(3000)
    h = capi.Index.from_host(idx, 0)
    yield h
    h.close()


def _want(which, k, s, q, L):
    if which == 0:
        return loader.ars_sort(0, k, L=L)
    return loader.ars_sort(1, s, k, q, L=L)


def _cases():
    rng = np.random.default_rng(7)
    out = []
    for n in list(range(0, 40)) + [63, 64, 65, 100, 129, 300, 777, 1024]:
        for spread in (1, 2, 3, 8, 1 << 20):                   # 1: all keys equal ... 2^20: hardly any tie
            k = rng.integers(0, spread, size=n)
            s = rng.integers(0, min(spread, 4), size=n)
            q = rng.integers(0, 2, size=n)
            out.append((k, s, q))
    for n in (17, 64, 200, 1000):                               # shapes that drive the quicksort to its depth limit
        a = np.arange(n)
        for k in (a, a[::-1], np.minimum(a, n - 1 - a), np.where(a % 2 == 0, a, n - a), np.zeros(n, dtype=np.int64)):
            out.append((k.astype(np.int64), (k % 3).astype(np.int64), (k % 2).astype(np.int64)))
    return out


def test_wave_tier_sorts_equal_ksort(ix):
    L = loader.ref_chain_lib()
    n_tied = 0
    for k, s, q in _cases():
        for which in (0, 1):
            want = _want(which, k, s, q, L)
            for mode in (0, 1, 2):
                got = ix.debug_sort(k, s, q, which, mode)
                assert np.array_equal(got, want), (len(k), which, mode, k[:12], got[:12], want[:12])
            n_tied += len(k) > len(np.unique(k))
    assert n_tied > 100
