"""The ERT restatement (oracle/ert_oracle.c) checked from first principles and against the FM-index restatement.

ertindex.cpp / ertseeding.cpp cannot be compiled here and the reference ships no ERT fixtures (parity unpinned), so
  * the decoder over the written index is pinned to the definitions: longest match with >= m occurrences and the
    occurrence list in suffix order, both by brute force over the text;
  * seeding over the ERT is pinned to seeding over the FM-index (orc_collect_smem + orc_sa_lookup), which the
    reference's ERT mode reproduces by design (src/ertseeding.cpp:2891 "Extra work for equivalency with BWA-MEM").
"""
import numpy as np
import pytest

from bwams import fmindex, simulate
from oracle import loader


def _setup(n_bases, seed, kmer, xmer, thr, repeat_frac=0.3, read_len=151):
    g = simulate.make_genome(n_bases, seed=seed, repeat_frac=repeat_frac, repeat_len=180, n_families=3)
    idx = fmindex.build_fmindex(g)
    text = fmindex.fw_rc_text(g)
    o = loader.OracleFMI(idx)
    e = loader.OracleERT(o, text, kmer=kmer, xmer=xmer, read_len=read_len, hit_threshold=thr)
    return g, idx, text, o, e


def _occ(tb, pat):
    """start positions of pat in the text, in suffix order of the text"""
    out, st = [], tb.find(pat)
    while st >= 0:
        out.append(st)
        st = tb.find(pat, st + 1)
    return sorted(out, key=lambda p: tb[p:] + b"\x00")


def test_entry_fields():
    g, idx, text, o, e = _setup(3000, 3, kmer=5, xmer=2, thr=6)
    tb = bytes((text + 1).astype(np.uint8))
    codes = e.kmer_table & 3
    assert set(np.unique(codes)) == {0, 1, 2, 3}            # all four entry kinds occur at this size
    ptr = (e.kmer_table >> 24).astype(np.int64)
    assert np.all(np.diff(ptr) >= 0) and ptr[-1] <= len(e.mlt)
    for key in range(4 ** 5):
        pat = bytes(((key >> (2 * j)) & 3) + 1 for j in range(5))
        n = len(_occ(tb, pat))
        ent = int(e.kmer_table[key])
        assert (ent & 3) == (0 if n == 0 else 1 if n == 1 else 2 if n <= 6 else 3), key
        assert ((ent >> 17) & 31) == (n if n < 20 else 0)
        if n == 1:
            assert e.mlt[ptr[key]] == 0
            assert int.from_bytes(bytes(e.mlt[ptr[key] + 1: ptr[key] + 6]), "little") >> 1 == _occ(tb, pat)[0]


@pytest.mark.parametrize("kmer,xmer,thr,seed", [(6, 2, 8, 1), (4, 1, 3, 2), (8, 3, 256, 3)])
def test_decoder_matches_brute_force(kmer, xmer, thr, seed):
    g, idx, text, o, e = _setup(6000, seed, kmer, xmer, thr)
    tb = bytes((text + 1).astype(np.uint8))
    reads, _, _ = simulate.make_reads(g, 40, seed=seed + 10, read_len=120)
    rng = np.random.default_rng(seed)
    for r in reads[:25]:
        r = np.array(r, dtype=np.uint8)
        if rng.random() < 0.3:
            r[int(rng.integers(0, len(r)))] = 4
        rb = bytes((r + 1).astype(np.uint8))
        for i in range(0, len(r), 7):
            L = e.profile(r, i, 20)
            # brute force: longest prefix of r[i:] with >= m occurrences (no N), by doubling the prefix
            best = [0] * 21
            n = 0
            while i + n < len(r) and r[i + n] < 4:
                c = len(_occ(tb, rb[i:i + n + 1])) if n + 1 <= 40 or best[1] == n else 0
                if c == 0:
                    break
                n += 1
                for m in range(1, min(c, 20) + 1):
                    best[m] = n
            for m in range(1, 21):
                if L[m - 1] == 0:
                    assert best[m] < kmer + xmer, (i, m, best[m])      # matches the table lookups do not resolve are reported as 0
                else:
                    assert L[m - 1] == best[m], (i, m, L[m - 1], best[m])
            if L[0] >= kmer + xmer:
                for mlen in {int(L[0]), kmer + xmer, (int(L[0]) + kmer + xmer) // 2}:
                    hits, cnt = e.hits(r, i, mlen)
                    want = _occ(tb, rb[i:i + mlen])
                    assert cnt == len(want) and list(hits) == want, (i, mlen)


@pytest.mark.parametrize("kmer,xmer,thr,n_bases,seed", [(6, 2, 8, 20000, 5), (8, 2, 16, 60000, 6), (5, 2, 4, 4000, 7)])
def test_seeding_over_ert_equals_seeding_over_fm(kmer, xmer, thr, n_bases, seed):
    g, idx, text, o, e = _setup(n_bases, seed, kmer, xmer, thr)
    reads, _, _ = simulate.make_reads(g, 300, seed=seed, read_len=150)
    rng = np.random.default_rng(seed)
    reads = [np.array(r, dtype=np.uint8) for r in reads]
    for r in reads[::9]:
        r[int(rng.integers(0, len(r)))] = 4              # reads with N
    reads.append(rng.integers(0, 4, size=150).astype(np.uint8))        # nothing to find
    reads.append(np.array(g[:150], dtype=np.uint8))                   # the start of the text
    reads.append(np.array(g[-150:], dtype=np.uint8))                  # the strand boundary
    enc = np.concatenate(reads)
    cum = np.concatenate([[0], np.cumsum([len(r) for r in reads])]).astype(np.int64)
    for opt in (loader.default_seed_opt(), loader.SeedOpt(19, 1.2, 12, 15, 7), loader.SeedOpt(25, 1.5, 10, 20, 500)):
        fm = o.collect_smem(enc, cum, opt)
        fm_coord, fm_off = o.sa_lookup(fm, opt.max_occ)
        em, e_coord, e_off = e.collect(enc, cum, opt)
        assert len(fm) == len(em)
        for f in ("rid", "m", "n", "s"):
            assert np.array_equal(fm[f], em[f]), f
        assert np.array_equal(fm_off, e_off)
        # the FM-index lookup returns 0 for a row whose LF walk meets the sentinel before a sampled row (a few
        # positions at the very start of the text, FMI_search.cpp:2234-2237); the ERT stores the true position
        same = fm_coord == e_coord
        assert np.all(same | ((fm_coord == 0) & (e_coord < 128)))
