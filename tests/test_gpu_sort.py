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


def _ks_introsort_trace(n, lt):
    """ksort.h's introsort (src/ksort.h: ks_introsort) over the indices 0..n-1 with a caller-supplied `lt(i, j)` on ELEMENT ids;
    returns True when the depth limit sent a range to the comb-sort fallback.  Used only to BUILD an adversarial input."""
    a = list(range(n))
    hit = [False]
    if n < 1:
        return False
    if n == 2:
        return False
    d = 2
    while (1 << d) < n:
        d += 1
    d <<= 1
    stack, s, t = [], 0, n - 1
    while True:
        if s < t:
            d -= 1
            if d == 0:
                hit[0] = True
                t = s
                continue
            i, j = s, t
            k = i + ((j - i) >> 1) + 1
            if lt(a[k], a[i]):
                if lt(a[k], a[j]):
                    k = j
            else:
                k = i if lt(a[j], a[i]) else j
            rp = a[k]
            if k != t:
                a[k], a[t] = a[t], a[k]
            while True:
                i += 1
                while lt(a[i], rp):
                    i += 1
                j -= 1
                while i <= j and lt(rp, a[j]):
                    j -= 1
                if j <= i:
                    break
                a[i], a[j] = a[j], a[i]
            a[i], a[t] = a[t], a[i]
            if i - s > t - i:
                if i - s > 16:
                    stack.append((s, i - 1, d))
                s = i + 1 if t - i > 16 else t
            else:
                if t - i > 16:
                    stack.append((i + 1, t, d))
                t = i - 1 if i - s > 16 else s
        else:
            if not stack:
                return hit[0]
            s, t, d = stack.pop()


def _antiquicksort(n):
    """McIlroy's adversary ("A killer adversary for quicksort", 1999) against the introsort above: values are fixed only when
    a comparison needs them, so that every pivot turns out to be among the smallest of its range.  -> keys (a permutation)."""
    GAS = n
    val = [GAS] * n
    state = {"nsolid": 0, "cand": 0}

    def lt(x, y):
        if val[x] == GAS and val[y] == GAS:
            if x == state["cand"]:
                val[x] = state["nsolid"]
            else:
                val[y] = state["nsolid"]
            state["nsolid"] += 1
        if val[x] == GAS:
            state["cand"] = x
        elif val[y] == GAS:
            state["cand"] = y
        return val[x] < val[y]

    hit = _ks_introsort_trace(n, lt)
    rest = state["nsolid"]
    for i in range(n):
        if val[i] == GAS:
            val[i] = rest
            rest += 1
    return np.array(val, np.int64), hit


def test_depth_limit_fallback(ix):
    """The comb-sort fallback of ks_introsort's depth limit in the wave tiers (wave_combsort: the whole wavefront, no lane
    sorting alone on LDS): (a) an adversarial permutation on which ksort.h itself exhausts its depth budget — the order must be
    the pinned ks_introsort's; (b) a depth budget of 2 on inputs of every kind — the wave form must equal the sequential form
    operation for operation (ties included)."""
    L = loader.ref_chain_lib()
    for n in (200, 700, 1024):
        k, hit = _antiquicksort(n)
        assert hit, n                                              # the construction really drives ksort to its fallback
        s = (k % 5).astype(np.int64)
        q = (k % 2).astype(np.int64)
        for which in (0, 1):
            want = _want(which, k, s, q, L)
            for mode in (0, 1, 2):
                assert np.array_equal(ix.debug_sort(k, s, q, which, mode), want), (n, which, mode)
    rng = np.random.default_rng(11)
    for n in (3, 17, 18, 40, 65, 130, 500, 1024):
        for spread in (1, 2, 5, 1 << 20):
            k = rng.integers(0, spread, size=n)
            s = rng.integers(0, min(spread, 4), size=n)
            q = rng.integers(0, 2, size=n)
            for which in (0, 1):
                a, b = ix.debug_sort(k, s, q, which, 3), ix.debug_sort(k, s, q, which, 4)
                assert np.array_equal(a, b), (n, spread, which)
                keys = k[a] if which == 0 else np.stack([-s[a], k[a], q[a]], 1)
                srt = np.all(np.diff(keys) >= 0) if which == 0 else all(tuple(keys[i]) <= tuple(keys[i + 1]) for i in range(n - 1))
                assert srt
