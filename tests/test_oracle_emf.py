"""EMF: the restated primitives are pinned to the reference's perfect.h (oracle/_ref/libref_emf.so);
the table builder and the probe loop are checked against brute force."""
import ctypes as C

import numpy as np
import pytest

from bwams import emf, fmindex, simulate
from oracle import loader
import os

HERE = os.path.dirname(os.path.abspath(loader.__file__))
_path = os.path.join(HERE, "_ref", "libref_emf.so")
REF = C.CDLL(_path) if os.path.exists(_path) else None
needs_ref = pytest.mark.skipif(REF is None, reason="oracle/_ref/libref_emf.so not built (reference tree absent)")


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


@needs_ref
def test_primitives_equal_reference_header():
    L = loader.lib()
    for fn in (L.orc_emf_hash, REF.ref_emf_hash_fw, REF.ref_emf_hash_rc):
        fn.restype = C.c_int64
    L.orc_emf_hash.argtypes = [C.c_uint32, C.c_void_p, C.c_int, C.c_int]
    REF.ref_emf_hash_fw.argtypes = REF.ref_emf_hash_rc.argtypes = [C.c_uint32, C.c_void_p, C.c_int]
    assert REF.ref_emf_sizeof_table_header() == 64 and REF.ref_emf_sizeof_seed_entry() == 16
    rng = np.random.default_rng(5)
    for length in (150, 151, 100, 64, 33, 31, 8, 7, 250, 19):
        for _ in range(40):
            s = rng.integers(0, 4, size=length, dtype=np.uint8)
            n = int(rng.integers(1000, 4_000_000_000))
            assert L.orc_emf_hash(n, _p(s), length, 1) == REF.ref_emf_hash_fw(n, _p(s), length)
            assert L.orc_emf_hash(n, _p(s), length, 0) == REF.ref_emf_hash_rc(n, _p(s), length)
            if rng.random() < 0.3:                       # near-palindromes exercise the tie rule
                half = (length + 1) // 2
                s[length - half:] = (3 - s[:half])[::-1]
                if rng.random() < 0.5:
                    s[rng.integers(0, length)] ^= 1
            assert L.orc_emf_compare_fw_rc(_p(s), length) == REF.ref_emf_compare_fw_rc(_p(s), length)
            t = s.copy()
            if rng.random() < 0.7:
                t[rng.integers(0, length)] = rng.integers(0, 4)
            for afl in (0, 1):
                for bfl in (0, 1):
                    assert L.orc_emf_seedcmp(_p(s), afl, _p(t), bfl, length) == REF.ref_emf_seedcmp(_p(s), afl, _p(t), bfl, length)
    # tail match for reads longer than the table's L
    g = rng.integers(0, 4, size=3000, dtype=np.uint8)
    ref = np.concatenate([g, (3 - g[::-1]).astype(np.uint8)])
    t = loader.OrcEmf(50, 0, 16, len(g), None, None, ref.ctypes.data)
    L.orc_emf_match_further.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_int, C.c_int]
    REF.ref_emf_match_further.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_uint32, C.c_void_p, C.c_int, C.c_int]
    for _ in range(300):
        ln = int(rng.integers(51, 90))
        loc = int(rng.integers(0, len(g) - 50))
        is_rev = int(rng.integers(0, 2))
        if is_rev:
            lo = loc + 50 - ln
            rd = (3 - g[max(lo, 0): loc + 50][::-1]).astype(np.uint8) if lo >= 0 else rng.integers(0, 4, ln, dtype=np.uint8)
        else:
            rd = g[loc: loc + ln].copy()
        if len(rd) != ln:
            rd = rng.integers(0, 4, ln, dtype=np.uint8)
        if rng.random() < 0.4:
            rd[rng.integers(50, ln)] ^= 2
        a = L.orc_emf_match_further(C.byref(t), loc, _p(rd), is_rev, ln)
        b = REF.ref_emf_match_further(_p(ref), len(g), 50, loc, _p(rd), is_rev, ln)
        assert (a != 0) == (b != 0), (loc, is_rev, ln)


def _brute_locations(g, read):
    """forward-strand start positions where read or its reverse complement occurs exactly"""
    gb = bytes(g + 1)
    out = []
    for strand, s in ((0, read), (1, (3 - read[::-1]).astype(np.uint8))):
        pat = bytes(s + 1)
        i = gb.find(pat)
        while i >= 0:
            out.append((i, strand))
            i = gb.find(pat, i + 1)
    return out


@pytest.mark.parametrize("L", [50, 33])
def test_builder_and_probe_against_brute_force(L, tmp_path):
    g = simulate.make_genome(6000, seed=71, repeat_frac=0.3, repeat_len=120, n_families=2, repeat_div=0.01)
    g[100:160] = (3 - g[100:160][::-1])                      # a reverse-complement palindrome region
    tab = emf.build_emf(g, L)
    emf.write_emf(str(tmp_path / "t.perfect"), tab)
    tab2 = emf.read_emf(str(tmp_path / "t.perfect"))
    assert np.array_equal(tab2.seed_table, tab.seed_table) and np.array_equal(tab2.loc_table, tab.loc_table)
    ref = np.concatenate([g, (3 - g[::-1]).astype(np.uint8)])
    o = loader.OracleEMF(tab2, ref)
    rng = np.random.default_rng(3)
    n_hit = n_multi = 0
    for it in range(600):
        ln = L if it % 3 else int(rng.integers(L + 1, L + 40))
        st = int(rng.integers(0, len(g) - ln))
        rd = g[st: st + ln].copy()
        kind = it % 5
        if kind == 1:
            rd = (3 - rd[::-1]).astype(np.uint8)
        elif kind == 2:
            rd[rng.integers(0, ln)] ^= 1
        elif kind == 3:
            rd[rng.integers(0, ln)] = 4
        code, flags, loc = o.probe(rd)
        if kind == 3:
            assert code == 1
            continue
        truth = _brute_locations(g, rd)
        if not truth:
            assert code in (2, 5), (it, code)
            continue
        assert code in (3, 4), (it, code, truth[:3])
        # the reported location is a true occurrence on the reported strand
        is_rc = 1 if (flags & 2) else 0
        assert flags & 1
        start = loc if not is_rc else loc - (ln - L)         # init_mem_aln_perfect (perfect_map.cpp:670-672)
        assert (start, is_rc) in truth, (it, loc, is_rc, truth[:4])
        n_hit += 1
        if flags >> 2:
            n_multi += 1
    assert n_hit > 200 and n_multi > 10


def test_torch_builder_against_brute_force():
    """The bench-scale builder (torch; here on the CPU device) yields a valid table: every probe result
    agrees with exhaustive search, and its hit/miss decisions equal the numpy builder's."""
    L = 70
    g = simulate.make_genome(9000, seed=81, repeat_frac=0.3, repeat_len=160, n_families=2, repeat_div=0.005)
    g[300:420] = (3 - g[300:420][::-1])
    t = emf.build_emf_torch(g, L, "cpu")
    tab = emf.EmfTable(t.seed_len, t.seq_len, t.loc_table.numpy().view(np.uint32), t.seed_table.numpy().view(np.uint32),
                       t.num_seed_used, t.num_seed_key)
    ref_tab = emf.build_emf(g, L)
    assert tab.num_seed_used == ref_tab.num_seed_used and tab.num_seed_key == ref_tab.num_seed_key
    ref = np.concatenate([g, (3 - g[::-1]).astype(np.uint8)])
    o, o2 = loader.OracleEMF(tab, ref), loader.OracleEMF(ref_tab, ref)
    rng = np.random.default_rng(4)
    hits = 0
    for it in range(500):
        ln = L if it % 3 else int(rng.integers(L + 1, L + 30))
        st = int(rng.integers(0, len(g) - ln))
        rd = g[st: st + ln].copy()
        if it % 4 == 1:
            rd = (3 - rd[::-1]).astype(np.uint8)
        elif it % 4 == 2:
            rd[rng.integers(0, ln)] ^= 1
        code, flags, loc = o.probe(rd)
        code2, _, _ = o2.probe(rd)
        assert (code in (3, 4)) == (code2 in (3, 4)), (it, code, code2)
        truth = _brute_locations(g, rd)
        if code in (3, 4):
            is_rc = 1 if flags & 2 else 0
            assert (loc if not is_rc else loc - (ln - L), is_rc) in truth
            hits += 1
        else:
            assert not truth
    assert hits > 200
