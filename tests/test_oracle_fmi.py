"""The FM-index restatement (oracle/fmi_oracle.c) checked from first principles.

The reference's FMI_search.cpp cannot be compiled here (un-vendored safestringlib)
and ships no vectors, so seeding parity is formally unpinned; these tests pin the
restatement to the *definitions* instead: Occ by counting, bi-intervals by naive
suffix sorting, SMEMs by exhaustive search, SA values by the full suffix array.
"""
import numpy as np
import pytest

from bwams import fmindex, simulate
from oracle import loader
from util import naive_sa, toy, toy_reads


def test_suffix_array_matches_naive():
    g = simulate.make_genome(700, seed=3, repeat_frac=0.3, repeat_len=60, n_families=2)
    text = fmindex.fw_rc_text(g)
    sa = fmindex._suffix_array_numpy(text)
    assert np.array_equal(sa, naive_sa(text))
    assert sa[0] == len(text)


def test_index_layout_and_occ_by_counting():
    g = simulate.make_genome(1500, seed=5, repeat_frac=0.2, repeat_len=80, n_families=2)
    idx = fmindex.build_fmindex(g)
    text = fmindex.fw_rc_text(g)
    sa = naive_sa(text)
    L = len(text) + 1
    assert idx.ref_seq_len == L and idx.cp_occ.shape == ((L >> 6) + 1, 8)
    bwt = np.where(sa > 0, text[np.maximum(sa - 1, 0)], 4)
    assert idx.sentinel_index == int(np.flatnonzero(sa == 0)[0])
    # count[] carries the loader's +1
    assert list(idx.count) == [1 + int((text < c).sum()) for c in range(4)] + [L]
    o = loader.OracleFMI(idx)
    for pos in list(range(0, 200)) + [L - 1, L, 777, 1024, 1025]:
        for c in range(4):
            assert o.occ(pos, c) == int((bwt[:pos] == c).sum()), (pos, c)
    # sampled SA
    samp = (idx.sa_ms_byte.astype(np.int64) << 32) + idx.sa_ls_word.astype(np.int64)
    assert np.array_equal(samp[: len(sa[::8])], sa[::8])


def _interval(text, sa, pat):
    """(k, s): rows of sa whose suffix starts with pat."""
    b = bytes((text + 1).astype(np.uint8))
    p = bytes((np.asarray(pat) + 1).astype(np.uint8))
    rows = [i for i, st in enumerate(sa) if b[st:st + len(p)] == p]
    return (rows[0], len(rows)) if rows else (None, 0)


def test_backward_ext_gives_true_biintervals():
    g = simulate.make_genome(900, seed=9, repeat_frac=0.3, repeat_len=50, n_families=2)
    idx = fmindex.build_fmindex(g)
    text = fmindex.fw_rc_text(g)
    sa = naive_sa(text)
    o = loader.OracleFMI(idx)
    rng = np.random.default_rng(1)
    for _ in range(40):
        st = int(rng.integers(0, len(g) - 30))
        pat = list(g[st:st + 12])
        a = pat[-1]
        k, l, s = int(idx.count[a]), int(idx.count[3 - a]), int(idx.count[a + 1] - idx.count[a])
        for j in range(len(pat) - 2, -1, -1):           # extend to the left
            k, l, s = o.backward_ext(k, l, s, int(pat[j]))
            sub = pat[j:]
            kk, ss = _interval(text, sa, sub)
            assert s == ss and (ss == 0 or k == kk)
            rc = [3 - x for x in sub[::-1]]
            ll, ss2 = _interval(text, sa, rc)
            assert ss2 == ss and (ss == 0 or l == ll)


def _brute_smems(read, text_bytes, min_len):
    """All maximal exact matches of read vs text that are not contained in a longer one."""
    L = len(read)
    rb = bytes((np.asarray(read) + 1).astype(np.uint8))
    ext = []                                  # longest match length starting at each i
    for i in range(L):
        n = 0
        while i + n < L and read[i + n] < 4 and rb[i:i + n + 1] in text_bytes:
            n += 1
        ext.append(n)
    out = []
    for i in range(L):
        if ext[i] == 0:
            continue
        if i > 0 and ext[i - 1] >= ext[i] + 1:    # contained in the match starting at i-1
            continue
        if ext[i] >= min_len:
            out.append((i, i + ext[i] - 1))
    return out


def test_round1_smems_are_the_supermaximal_matches():
    g, idx = toy(6000)
    reads, _, _ = toy_reads(6000, 60)
    text = fmindex.fw_rc_text(g)
    tb = bytes((text + 1).astype(np.uint8))
    o = loader.OracleFMI(idx)
    opt = loader.default_seed_opt()
    opt.max_mem_intv = 0            # round 3 off
    opt.split_width = -1            # round 2 off (no SMEM has s <= -1)
    enc, cum = simulate.flatten_reads(reads)
    sm = o.collect_smem(enc, cum, opt)
    for r in range(len(reads)):
        got = sorted({(int(x["m"]), int(x["n"])) for x in sm[sm["rid"] == r]})
        want = _brute_smems(reads[r], tb, opt.min_seed_len)
        assert got == want, (r, got, want)
    # interval sizes = number of occurrences (overlap-aware)
    for x in sm[:50]:
        pat = bytes((reads[x["rid"]][x["m"]: x["n"] + 1] + 1).astype(np.uint8))
        occ = sum(1 for i in range(len(tb) - len(pat) + 1) if tb[i:i + len(pat)] == pat)
        assert occ == int(x["s"])


def test_collect_smem_order_rounds_and_sa():
    g, idx = toy()
    reads, _, _ = toy_reads()
    o = loader.OracleFMI(idx)
    enc, cum = simulate.flatten_reads(reads)
    ctr = loader.Counters()
    sm = o.collect_smem(enc, cum, counters=ctr)
    assert len(sm) == sum(ctr.n_smem) and ctr.n_smem[2] > 0 and ctr.n_ext > 0
    key = (sm["rid"].astype(np.int64) << 32) | (sm["m"].astype(np.int64) << 16) | sm["n"]
    assert np.all(np.diff(key) >= 0)
    # every SMEM really occurs s times, k..k+s-1 are its rows
    text = fmindex.fw_rc_text(g)
    coord, off = o.sa_lookup(sm, 500)
    for i in np.random.default_rng(0).choice(len(sm), size=min(200, len(sm)), replace=False):
        x = sm[i]
        pat = reads[x["rid"]][x["m"]: x["n"] + 1]
        c = coord[off[i]: off[i + 1]]
        assert len(c) == min(int(x["s"]), 500)
        for p in c:
            assert np.array_equal(text[p: p + len(pat)], pat)
    # skip[] removes a read from every round
    skip = np.zeros(len(reads), np.uint8)
    skip[::3] = 1
    sm2 = o.collect_smem(enc, cum, skip=skip)
    assert np.array_equal(sm2, sm[skip[sm["rid"]] == 0])


def test_sa_entry_against_full_suffix_array_and_sentinel_quirk():
    g = simulate.make_genome(1200, seed=21, repeat_frac=0.0)
    idx = fmindex.build_fmindex(g)
    sa = naive_sa(fmindex.fw_rc_text(g))
    o = loader.OracleFMI(idx)
    sent = idx.sentinel_index
    # rows whose LF-walk crosses the sentinel row before reaching a sample return 0
    for pos in range(len(sa)):
        v = o.sa_entry(pos)
        walk, p, hit = 0, pos, False
        while p % 8 != 0:
            if p == sent:
                hit = True
                break
            p = int(np.flatnonzero(sa == sa[p] - 1)[0])
            walk += 1
        assert v == (0 if hit else int(sa[pos])), pos


def test_file_roundtrip(tmp_path):
    g, idx = toy(3000)
    fmindex.write_index(str(tmp_path / "toy"), idx)
    r = fmindex.read_index(str(tmp_path / "toy"))
    assert r.ref_seq_len == idx.ref_seq_len and r.sentinel_index == idx.sentinel_index
    for a, b in ((r.count, idx.count), (r.cp_occ, idx.cp_occ), (r.sa_ms_byte, idx.sa_ms_byte),
                 (r.sa_ls_word, idx.sa_ls_word), (r.ref_0123, idx.ref_0123)):
        assert np.array_equal(a, b)
    import os
    L = idx.ref_seq_len
    assert os.path.getsize(tmp_path / "toy.bwt.2bit.64") == 8 + 40 + ((L >> 6) + 1) * 64 + ((L >> 3) + 1) * 5 + 8


def test_fma_tables_are_pure_accelerators():
    """all_smem / last_smem replace the first forward steps with table lookups: with the
    reference's parameter relation (min_seed_len + 1 > last_bp) and reads without N the SMEM
    output is identical and only the number of extensions drops (FMA on vs off
    gave byte-identical SAM)."""
    g, idx = toy(30000, seed=17)
    reads, _, _ = simulate.make_reads(g, 400, seed=23)
    reads = reads[(reads < 4).all(axis=1)]
    enc, cum = simulate.flatten_reads(reads)
    o = loader.OracleFMI(idx)
    c0 = loader.Counters()
    plain = o.collect_smem(enc, cum, counters=c0)
    all_tab, last_tab = o.build_fma(7, 8)
    c1 = loader.Counters()
    fma = o.collect_smem(enc, cum, counters=c1)
    assert np.array_equal(plain, fma)
    assert c1.n_ext < c0.n_ext and list(c1.n_smem) == list(c0.n_smem)
    # table entries against direct extension: entry for a k-mer that occurs holds its interval
    mer = reads[0][:8]
    i8 = int(sum(int(b) << (2 * (7 - t)) for t, b in enumerate(mer)))
    ent = last_tab[i8]
    bp = int(ent[0] & 0xff)
    k, l, s = int(idx.count[mer[0]]), int(idx.count[3 - mer[0]]), int(idx.count[mer[0] + 1] - idx.count[mer[0]])
    for b in mer[1:bp]:
        l2, k2, s2 = o.backward_ext(l, k, s, 3 - int(b))      # forward = backward on the other strand
        k, l, s = k2, l2, s2
    assert (int(ent[1]), int(ent[2]), int(ent[3])) == (k & 0xffffffff, l & 0xffffffff, s & 0xffffffff)
    # reads with N: the reference's with_N quirk may shorten a match; rounds still agree in count of reads seeded
    reads_n, _, _ = simulate.make_reads(g, 300, seed=29)
    encn, cumn = simulate.flatten_reads(reads_n)
    a = o.collect_smem(encn, cumn)
    o.drop_fma()
    b = o.collect_smem(encn, cumn)
    clean = (reads_n < 4).all(axis=1)
    assert np.array_equal(a[clean[a["rid"]]], b[clean[b["rid"]]])
