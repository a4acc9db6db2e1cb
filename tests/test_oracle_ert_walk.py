"""The reference's OWN ERT walk, restated function by function (oracle/ert_walk_oracle.c = src/ertseeding.cpp's
get_seeds[_prefix] / reseed[_prefix] / last with every tree-walk variant, and the per-read driver of
mem_kernel1_core_ert), against the two other statements of ERT-mode seeding:

  * the match-profile formulation (oracle/ert_oracle.c: orc_ert_collect), which is what the HIP kernels implement;
  * FM-index seeding (orc_collect_smem + orc_sa_lookup).

What is compared is what mem_kernel1_core_ert hands mem_chain_new (bwamem.cpp:993-1006): the MEMs and, per MEM, the
sampled hit coordinates after the forward / fetch_leaves / end_correction mapping — in particular for MEMs found by
the BACKWARD walk, with 1 < hits <= max_occ (insertion order into the chain B-tree) and with hits > max_occ (which
subset the stride picks).  The reference re-gathers the hits of every multi-hit backward MEM by a forward traversal
("to report hits in the same order as BWA-MEM", ertseeding.cpp:644-648; rightExtend_fetch_leaves*, :1854-2140), so
only single-hit MEMs keep the reverse-complemented coordinate, where order cannot matter; the tests below show both
facts on the restated code: every class occurs, and the three statements agree.

The one place where they do not agree — reads whose placement at a hit would cross the junction between the two strands
of the text, where get_seq (ertseeding.cpp:455-472) hands back nothing and leaf expansion stops short — is pinned too
(test_strand_junction_is_the_only_difference).
"""
import collections

import numpy as np
import pytest

from bwams import fmindex, simulate
from oracle import loader


def _make(g, kmer, xmer, thr, read_len=151):
    idx = fmindex.build_fmindex(g)
    text = fmindex.fw_rc_text(g)
    o = loader.OracleFMI(idx)
    e = loader.OracleERT(o, text, kmer=kmer, xmer=xmer, read_len=read_len, hit_threshold=thr)
    return idx, text, o, e


def _reads(g, n, seed, margin=200):
    """simulated reads (a seventh of them with an N) that stay `margin` bases clear of both ends of the strand"""
    reads, _, _ = simulate.make_reads(g[margin:len(g) - margin], n, seed=seed)
    reads = [np.array(r, dtype=np.uint8) for r in reads]
    rng = np.random.default_rng(seed)
    for r in reads[::7]:
        r[int(rng.integers(0, len(r)))] = 4
    return reads


def _opt(**kw):
    oo = loader.default_seed_opt()
    for k, v in kw.items():
        setattr(oo, k, v)
    return oo


def _same(a, b):
    return len(a) == len(b) and all(np.array_equal(a[f], b[f]) for f in ("rid", "m", "n", "s"))


OPTS = ({}, {"split_factor": 1.2, "split_width": 12, "max_mem_intv": 15, "max_occ": 7}, {"min_seed_len": 25},
        {"max_mem_intv": 0}, {"max_occ": 3, "split_width": 18})


@pytest.mark.parametrize("kmer,xmer,thr,n_bases,seed", [(8, 2, 16, 100000, 1), (6, 2, 6, 20000, 2), (10, 4, 256, 300000, 3)])
def test_walk_equals_profile_formulation_and_fm(kmer, xmer, thr, n_bases, seed):
    g = simulate.make_genome(n_bases, seed=seed, repeat_frac=0.2, repeat_len=200, n_families=3)
    idx, text, o, e = _make(g, kmer, xmer, thr)
    enc, cum = simulate.flatten_reads(_reads(g, 1200, seed))
    seen = np.zeros(8, np.int64)
    for kw in OPTS:
        oo = _opt(**kw)
        got, coord, off, cls, flags = e.walk_collect(enc, cum, oo)
        assert flags == 0                                            # no assert of the reference trips, no stack underflow
        want, wcoord, woff = e.collect(enc, cum, oo)
        assert _same(got, want) and np.array_equal(off, woff) and np.array_equal(coord, wcoord), kw
        fm = o.collect_smem(enc, cum, oo)
        fcoord, foff = o.sa_lookup(fm, oo.max_occ)
        assert _same(got, fm) and np.array_equal(off, foff), kw
        # get_sa_entries' sentinel quirk (0 for the rows whose LF walk meets the sentinel) is the FM path's alone
        assert np.all((coord == fcoord) | ((fcoord == 0) & (coord < 128))), kw
        seen += np.bincount(cls, minlength=8)
    # every class of MEM the walk produces took part: backward single-hit (0), forward (1), backward re-gathered by a
    # forward traversal (2), backward single-hit whose leaf expansion ran past the search start (4: end_correction)
    assert seen[0] > 100 and seen[1] > 1000 and seen[2] > 100 and seen[4] > 100


def test_walk_equals_fm_on_the_harder_genome_shape():
    """The same three-way agreement on simulate.make_genome(profile="grch38_like") — a tandem array of higher-order repeats,
    microsatellites, poly-A runs, exact segmental duplications — with reads drawn from all of it."""
    g = simulate.make_genome(300000, seed=12, profile="grch38_like")
    idx, text, o, e = _make(g, 8, 2, 16)
    enc, cum = simulate.flatten_reads(_reads(g, 1500, 21, margin=400))
    for kw in ({}, {"max_occ": 7, "split_width": 15}):
        oo = _opt(**kw)
        got, coord, off, cls, flags = e.walk_collect(enc, cum, oo)
        assert flags == 0
        want, wcoord, woff = e.collect(enc, cum, oo)
        assert _same(got, want) and np.array_equal(off, woff) and np.array_equal(coord, wcoord), kw
        fm = o.collect_smem(enc, cum, oo)
        fcoord, foff = o.sa_lookup(fm, oo.max_occ)
        assert _same(got, fm) and np.array_equal(off, foff), kw
        assert np.all((coord == fcoord) | ((fcoord == 0) & (coord < 128))), kw
        assert int(got["s"].max()) >= 10 and (cls & 2).any()           # seeds in the repeats, backward MEMs re-gathered forward


def _heavy_genome(n_bases, seed):
    """a family of 900 near-identical 300-bp copies (seeds with more than 500 hits), a 171-bp tandem array, a poly-A run"""
    rng = np.random.default_rng(seed)
    g = rng.integers(0, 4, size=n_bases).astype(np.uint8)
    fam = rng.integers(0, 4, size=300).astype(np.uint8)
    for _ in range(900):
        p = int(rng.integers(1000, n_bases - 1300))
        cp = fam.copy()
        for m in rng.integers(0, 300, size=int(rng.integers(0, 4))):
            cp[m] = rng.integers(0, 4)
        g[p:p + 300] = cp
    sat = rng.integers(0, 4, size=171).astype(np.uint8)
    p = n_bases // 2
    for c in range(60):
        g[p + c * 171:p + (c + 1) * 171] = sat
    g[5000:5060] = 0
    return g


def test_hits_beyond_max_occ_and_multi_hit_backward_mems():
    """The case VERDICT r2 raised: MEMs found by the backward walk with 1 < hits <= max_occ and with hits > max_occ.  In
    the restated walk they all carry fetch_leaves (hits re-gathered in forward order); sampled coordinates == FM rows."""
    g = _heavy_genome(400000, 3)
    idx, text, o, e = _make(g, 8, 2, 16)
    enc, cum = simulate.flatten_reads(_reads(g, 2500, 11, margin=400))
    for kw in ({}, {"max_occ": 50}, {"split_width": 19, "max_mem_intv": 20, "split_factor": 1.0}):
        oo = _opt(**kw)
        got, coord, off, cls, flags = e.walk_collect(enc, cum, oo)
        assert flags == 0
        want, wcoord, woff = e.collect(enc, cum, oo)
        assert _same(got, want) and np.array_equal(off, woff) and np.array_equal(coord, wcoord), kw
        fm = o.collect_smem(enc, cum, oo)
        fcoord, foff = o.sa_lookup(fm, oo.max_occ)
        assert _same(got, fm) and np.array_equal(off, foff)
        assert np.all((coord == fcoord) | ((fcoord == 0) & (coord < 128)))
        big = got["s"] > oo.max_occ
        multi = (got["s"] > 1) & ~big
        backward = (cls & 1) == 0
        assert (big & backward).sum() > 300 and (multi & backward).sum() > 300
        # a backward-found MEM with more than one hit always had its hits re-gathered by rightExtend_fetch_leaves*
        assert np.all((cls[backward & (got["s"] > 1)] & 2) == 2)
        # ... so the only MEMs mem_chain_new maps back from the reverse-complemented coordinate have exactly one hit
        assert np.all(got["s"][backward & ((cls & 2) == 0)] == 1)
        # the stride really drops hits here: sampled < all
        assert (off[1:] - off[:-1])[big].max() <= oo.max_occ and got["s"][big].max() > oo.max_occ


def test_walk_output_through_the_ert_tail():
    """mem_kernel1_core_ert's tail (introsort, mem_chain_new, mem_chain_flt) on the walk's REAL records and hit arrays
    gives the chains mem_chain_seeds builds from the FM-index seeds."""
    g = _heavy_genome(300000, 5)
    idx, text, o, e = _make(g, 8, 2, 16)
    enc, cum = simulate.flatten_reads(_reads(g, 1500, 4, margin=400))
    oo = _opt()
    mems, mem_off, hits, hit_off, flags = e.walk(enc, cum, oo)
    assert flags == 0 and len(mems) > 5000
    assert {0, 1} <= set(np.unique(mems["forward"])) and mems["fetch_leaves"].any() and mems["end_correction"].any()
    ch_e, sd_e, off_e = loader.chain_new_ert(mems, mem_off, hits, hit_off, cum, len(g), ref_string=text, enc=enc)
    fm = o.collect_smem(enc, cum, oo)
    fcoord, foff = o.sa_lookup(fm, oo.max_occ)
    ch_f, sd_f, off_f = loader.chain_seeds(fm, fcoord, foff, cum, len(g), ref_string=text, enc=enc)
    assert np.array_equal(off_e, off_f) and len(ch_e) == len(ch_f) and len(sd_e) == len(sd_f)
    for f in ("seqid", "n", "first", "rid", "w_kept_alt", "frac_rep", "pos"):
        assert np.array_equal(ch_e[f], ch_f[f]), f
    for f in ("rbeg", "qbeg", "len", "score"):
        assert np.array_equal(sd_e[f], sd_f[f]), f


def test_strand_junction_is_the_only_difference():
    """get_seq (ertseeding.cpp:455-472) returns nothing for a window that bridges the forward / reverse-complement
    junction of the text, so the reference's leaf expansion stops short for a read whose placement AT A HIT would cross
    l_pac (whether or not the match itself gets there), and check_and_add_smem_prefix then emits matches that are not
    maximal.  FM-index seeding (and the HIP path, which follows it: include/bwams.h, bwams_seed_run_ert) finds the true
    SMEMs there.  Reads that stay clear of the junction by a read length agree in all three statements."""
    g = simulate.make_genome(100000, seed=1, repeat_frac=0.2, repeat_len=200, n_families=3)
    idx, text, o, e = _make(g, 8, 2, 16)
    L = len(g)
    rng = np.random.default_rng(5)
    near, clear = [], []
    for k in range(0, 140, 7):
        near.append(np.array(text[L - 75 + k:L + 75 + k], dtype=np.uint8))      # across the junction
        near.append(np.array(g[L - 150 + k // 2:L - k // 2], dtype=np.uint8))   # within a read length of it, forward strand
        clear.append(np.array(g[:150 - k], dtype=np.uint8))                     # start of the text
        clear.append(np.array(text[2 * L - 150 + k:], dtype=np.uint8))          # end of the text
    for ln in (1, 5, 18, 19, 20, 21, 151):
        clear.append(rng.integers(0, 4, size=ln).astype(np.uint8))
        clear.append(np.array(g[2000:2000 + ln], dtype=np.uint8))
    clear.append(np.full(80, 4, np.uint8))
    for at in ([0], [149], [10, 11, 12, 13], [20, 40, 60, 100]):
        r = np.array(g[5000:5150], dtype=np.uint8)
        r[at] = 4
        clear.append(r)
    oo = _opt()

    def per_read(sm, co, of):
        d = collections.defaultdict(list)
        for t, r in enumerate(sm):
            d[int(r["rid"])].append((int(r["m"]), int(r["n"]), int(r["s"]), tuple(co[of[t]:of[t + 1]])))
        return d

    enc, cum = simulate.flatten_reads(clear)
    got, coord, off, cls, flags = e.walk_collect(enc, cum, oo)
    want, wcoord, woff = e.collect(enc, cum, oo)
    assert flags == 0 and _same(got, want) and np.array_equal(coord, wcoord) and len(got) > 40

    enc, cum = simulate.flatten_reads(near)
    got, coord, off, cls, flags = e.walk_collect(enc, cum, oo)
    want, wcoord, woff = e.collect(enc, cum, oo)
    assert flags == 0
    A, B = per_read(got, coord, off), per_read(want, wcoord, woff)
    differing = [r for r in range(len(near)) if A.get(r) != B.get(r)]
    assert differing                                                 # the difference is real ...
    for r in differing:                                              # ... and confined to placements that cross l_pac
        placements = [p - m for (m, n, s, ps) in B[r] + A[r] for p in ps]
        ln = int(cum[r + 1] - cum[r])
        assert any(p < L < p + ln for p in placements), r
