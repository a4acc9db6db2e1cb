"""GPU parity for chaining, chain filtering and chain-to-alignment: the HIP path through the
C-ABI against the CPU oracle on the same seeded inputs.  Integer work (plus frac_rep, one float
division): every comparison is bit-exact."""
import numpy as np
import pytest

from bwams import capi, fmindex, simulate
from oracle import loader
from util import toy

pytestmark = pytest.mark.gpu

CHAIN_FIELDS = ("seqid", "n", "m", "first", "rid", "w_kept_alt", "frac_rep", "pos", "seed_off")
SEED_FIELDS = ("rbeg", "qbeg", "len", "score")
REG_FIELDS = ("rb", "re", "qb", "qe", "rid", "chain", "score", "truesc", "sub", "alt_sc", "csub", "sub_n", "w",
              "seedcov", "secondary", "secondary_all", "seedlen0", "n_comp_is_alt", "frac_rep", "hash", "flg")
PAIR_IN = ("idr", "idq", "len1", "len2", "h0", "seqid", "regid")


def _mem_opts(**kw):
    o, g = loader.default_mem_opt(), capi.default_mem_opt()
    for k, v in kw.items():
        setattr(o, k, v)
        setattr(g, k, v)
    if "a" in kw:
        for i, v in enumerate(loader.fill_scmat(kw["a"], 4)):
            o.mat[i] = v
            g.mat[i] = v
    return o, g


@pytest.fixture(scope="module")
def rep_toy():
    """A genome with several repeat families so that reads carry many chains (> 9: deep B-trees)."""
    capi.lib()
    g = simulate.make_genome(120000, seed=31, repeat_frac=0.45, repeat_len=260, n_families=4)
    idx = fmindex.build_fmindex(g)
    ix = capi.Index.from_host(idx, 0)
    yield g, idx, ix
    ix.close()


def _reads(g, n, seed, extra=()):
    reads, _, _ = simulate.make_reads(g, n, seed=seed)
    reads = list(reads) + list(extra)
    return reads


def _run(idx, ix, g, reads, contigs=None, seed_kw=None, **mem_kw):
    enc, cum = simulate.flatten_reads(reads)
    oopt, gopt = _mem_opts(**mem_kw)
    so, sg = loader.default_seed_opt(), capi.default_seed_opt()
    so.max_occ = sg.max_occ = oopt.max_occ
    so.min_seed_len = sg.min_seed_len = oopt.min_seed_len
    for k, v in (seed_kw or {}).items():
        setattr(so, k, v)
        setattr(sg, k, v)
    o = loader.OracleFMI(idx)
    sm = o.collect_smem(enc, cum, so)
    coord, off = o.sa_lookup(sm, so.max_occ)
    l_pac = len(g)
    ref = np.concatenate([g, (3 - g[::-1]).astype(np.uint8)])
    want = {}
    want["chains"], want["seeds"], want["chain_off"] = loader.chain_seeds(sm, coord, off, cum, l_pac, contigs=contigs, opt=oopt,
                                                                          ref_string=ref, enc=enc)
    b = capi.Batch(ix, max(len(cum) - 1, 1), max(int(cum[-1]), 1), max_smem=len(sm) + 4096, max_sa=len(coord) + 4096)
    if contigs is not None:
        ix.set_contigs(contigs)
    else:
        c = np.zeros(1, capi.CONTIG_DTYPE)
        c["len"] = l_pac
        ix.set_contigs(c)
    b.seed_upload(enc, cum)
    b.seed_run(sg, with_sa=True)
    b.chain_run(gopt)
    got = {}
    got["chains"], got["seeds"], got["chain_off"] = b.chain_fetch()
    ctx = dict(enc=enc, cum=cum, ref=ref, l_pac=l_pac, oopt=oopt, gopt=gopt, contigs=contigs, sm=sm, coord=coord, off=off)
    return b, want, got, ctx


def _assert_regs(regs, wregs, exact_dead: bool):
    """Kept regions and the purge decisions are always identical; the contents of purged regions (which
    the reference computes and then drops, bwamem.cpp:1446-1456) only with extend_all = 1."""
    assert len(regs) == len(wregs)
    purged = (wregs["qb"] == -1) & (wregs["qe"] == -1)
    assert np.array_equal((regs["qb"] == -1) & (regs["qe"] == -1), purged)
    keep = slice(None) if exact_dead else ~purged
    for f in REG_FIELDS:
        assert np.array_equal(regs[f][keep], wregs[f][keep]), f


def _assert_chains(want, got):
    assert np.array_equal(got["chain_off"], want["chain_off"])
    assert len(got["chains"]) == len(want["chains"]) and len(got["seeds"]) == len(want["seeds"])
    for f in CHAIN_FIELDS:
        assert np.array_equal(got["chains"][f], want["chains"][f]), f
    for f in SEED_FIELDS:
        assert np.array_equal(got["seeds"][f], want["seeds"][f]), f


@pytest.mark.parametrize("batch", ["1", "0"])
def test_chains_match_oracle(rep_toy, monkeypatch, batch):
    """batch = 1 (default): the wave tier settles 64 seeds per pass (chain.hip: chain_seeds_batch); 0: one seed at a time."""
    monkeypatch.setenv("BWAMS_CHAIN_BATCH", batch)
    capi.debug_reload()
    g, idx, ix = rep_toy
    b, want, got, ctx = _run(idx, ix, g, _reads(g, 3000, 7))
    _assert_chains(want, got)
    per_read = np.diff(want["chain_off"])
    assert per_read.max() > 9 and len(want["chains"]) > 3000          # deep trees and the filter both exercised
    kept = (want["chains"]["w_kept_alt"] >> 29) & 3
    assert set(np.unique(kept)) >= {1, 2, 3}
    b.close()


@pytest.mark.parametrize("kw", [
    dict(w=20, max_chain_gap=80), dict(mask_level=0.2, drop_ratio=0.9), dict(max_occ=4), dict(min_chain_weight=40),
    dict(max_chain_extend=2), dict(min_seed_len=12), dict(min_chain_weight=1000),
])
def test_chain_options(rep_toy, kw):
    g, idx, ix = rep_toy
    b, want, got, ctx = _run(idx, ix, g, _reads(g, 1200, 9), **kw)
    _assert_chains(want, got)
    b.close()


def test_chain_duplicate_positions_and_ties(rep_toy):
    """Reads built from tandem copies of one segment: the same reference position is hit from
    query offsets further apart than the band, which makes chains with EQUAL positions (B-tree
    duplicate keys) and equal weights (introsort tie order)."""
    g, idx, ix = rep_toy
    rng = np.random.default_rng(5)
    extra = []
    for _ in range(300):
        st = int(rng.integers(0, len(g) - 400))
        unit = g[st:st + int(rng.integers(25, 60))]
        gap = rng.integers(0, 4, size=int(rng.integers(101, 160)), dtype=np.uint8)
        r = np.concatenate([unit, gap, unit, gap[:int(rng.integers(0, 50))], unit])[:400]
        extra.append(simulate.revcomp(r) if rng.random() < 0.5 else r)
    b, want, got, ctx = _run(idx, ix, g, _reads(g, 200, 3, extra), w=10)
    _assert_chains(want, got)
    # the case did occur: some read has two chains with the same position
    dup = 0
    for r in range(len(want["chain_off"]) - 1):
        p = want["chains"]["pos"][want["chain_off"][r]:want["chain_off"][r + 1]]
        dup += len(p) != len(np.unique(p))
    assert dup > 0
    # such reads leave the wave tiers' ordered array for the exact B-tree (reads with few seeds never use the array)
    assert 0 < b.stats().n_chain_redo <= dup
    b.close()


def test_reads_with_thousands_of_seeds(monkeypatch):
    """A family of 5000 nearly exact copies and max_occ above that: reads carry 4000 .. 20000 seeds and thousands of chains — the wave tier's
    classes beyond 4096 seeds (ordered array of 13000 positions in LDS; beyond that the B-tree in HBM), many passes of 64 hits per SMEM with
    new chains and extensions mixed, the many-chain filter's largest class and its sequential form."""
    capi.lib()
    rng = np.random.default_rng(77)
    unit = rng.integers(0, 4, size=150, dtype=np.uint8)
    parts = []
    for _ in range(5000):
        cp = unit.copy()
        mut = rng.random(150) < 0.003
        cp[mut] = (cp[mut] + rng.integers(1, 4, size=int(mut.sum()), dtype=np.uint8)) & 3
        parts += [cp, rng.integers(0, 4, size=40, dtype=np.uint8)]
    g = np.concatenate(parts)
    idx = fmindex.build_fmindex(g)
    ix = capi.Index.from_host(idx, 0)
    try:
        reads = _reads(g, 40, 3)
        b, want, got, ctx = _run(idx, ix, g, reads, max_occ=6000)
        monkeypatch.setenv("BWAMS_CHAIN_BATCH", "0")          # ... and the same reads one seed at a time
        capi.debug_reload()
        b0, want0, got0, _ = _run(idx, ix, g, reads, max_occ=6000)
        _assert_chains(want0, got0)
        b0.close()
        seeds_per_read = np.diff(ctx["off"][np.searchsorted(ctx["sm"]["rid"], np.arange(len(reads) + 1))])
        assert seeds_per_read.max() > 13000 and ((seeds_per_read > 4096) & (seeds_per_read <= 13000)).any(), seeds_per_read
        assert np.diff(want["chain_off"]).max() > 3840
        _assert_chains(want, got)
        b.close()
    finally:
        ix.close()


def test_chain_contigs_and_alt(rep_toy):
    g, idx, ix = rep_toy
    l_pac = len(g)
    contigs = np.zeros(4, capi.CONTIG_DTYPE)
    contigs["offset"] = [0, 30000, 30150, 90000]
    contigs["len"] = [30000, 150, 59850, l_pac - 90000]
    contigs["is_alt"] = [0, 0, 1, 0]
    b, want, got, ctx = _run(idx, ix, g, _reads(g, 2500, 17), contigs=contigs)
    _assert_chains(want, got)
    assert (want["chains"]["w_kept_alt"] >> 31).sum() > 0
    # and the extension windows are clipped to the contig of the chain
    b.extend_run(ctx["gopt"])
    regs, reg_off, aln = b.extend_fetch()
    wregs, wreg_off, wseeds = loader.chain2aln(want["chains"], want["seeds"], want["chain_off"], ctx["enc"], ctx["cum"],
                                               ctx["ref"], l_pac, contigs=contigs, opt=ctx["oopt"])
    assert np.array_equal(reg_off, wreg_off) and np.array_equal(aln, wseeds["aln"])
    _assert_regs(regs, wregs, False)
    b.close()
    c = np.zeros(1, capi.CONTIG_DTYPE)
    c["len"] = l_pac
    ix.set_contigs(c)


def test_single_smem_chunk_yields_no_chain(rep_toy):
    """mem_chain_seeds' loop guard skips a work item that holds exactly one SMEM (bwamem.cpp:819)."""
    g, idx, ix = rep_toy
    reads = [g[5000:5019].copy()]                 # one 19-mer -> at most one SMEM
    b, want, got, ctx = _run(idx, ix, g, reads)
    assert len(ctx["sm"]) == 1
    _assert_chains(want, got)
    assert len(got["chains"]) == 0
    b.close()


def test_extension_tasks_match_oracle(rep_toy):
    g, idx, ix = rep_toy
    b, want, got, ctx = _run(idx, ix, g, _reads(g, 2000, 23))
    nl, nr = b.extend_build(ctx["gopt"])
    wregs, wreg_off, wseeds, tasks = loader.chain2aln(want["chains"], want["seeds"], want["chain_off"], ctx["enc"],
                                                      ctx["cum"], ctx["ref"], ctx["l_pac"], opt=ctx["oopt"], build_only=True)
    assert nl == len(tasks["left"]) and nr == len(tasks["right"])
    regs, reg_off, aln = b.extend_fetch()
    assert np.array_equal(reg_off, wreg_off) and np.array_equal(aln, wseeds["aln"])
    for f in REG_FIELDS:
        assert np.array_equal(regs[f], wregs[f]), f
    for side, name in ((0, "left"), (1, "right")):
        pairs, ref, qer = b.extend_tasks_fetch(side)
        for f in PAIR_IN:
            assert np.array_equal(pairs[f], tasks[name][f]), (name, f)
        assert np.array_equal(ref, tasks[name + "_ref"]) and np.array_equal(qer, tasks[name + "_qer"])
    b.close()


@pytest.mark.parametrize("kw", [dict(), dict(extend_all=1), dict(w=12, zdrop=30), dict(w=12, zdrop=30, extend_all=1),
                                dict(pen_clip5=0, pen_clip3=11), dict(e_del=2, e_ins=3, o_del=4), dict(a=2), dict(a=2, extend_all=1)])
def test_regions_match_oracle(rep_toy, kw):
    g, idx, ix = rep_toy
    extra = []
    rng = np.random.default_rng(41)
    for _ in range(200):                          # reads with long indels: band retries and clipping decisions
        st = int(rng.integers(0, len(g) - 600))
        a = g[st:st + 70]
        gap = int(rng.integers(5, 60))
        if rng.random() < 0.5:
            r = np.concatenate([a, g[st + 70 + gap:st + 150 + gap]])
        else:
            r = np.concatenate([a, rng.integers(0, 4, size=gap, dtype=np.uint8), g[st + 70:st + 150]])
        extra.append(r)
    b, want, got, ctx = _run(idx, ix, g, _reads(g, 1500, 29, extra), **kw)
    n = b.extend_run(ctx["gopt"])
    regs, reg_off, aln = b.extend_fetch()
    wregs, wreg_off, wseeds, tasks_all = loader.chain2aln(want["chains"], want["seeds"], want["chain_off"], ctx["enc"], ctx["cum"],
                                                          ctx["ref"], ctx["l_pac"], opt=ctx["oopt"], want_tasks=True)
    assert n == len(wregs) and np.array_equal(reg_off, wreg_off) and np.array_equal(aln, wseeds["aln"])
    _assert_regs(regs, wregs, bool(kw.get("extend_all")))
    st = b.stats()
    assert st.n_chains == len(want["chains"]) and st.n_left + st.n_right > 0
    if kw.get("extend_all"):
        assert st.n_ext_rounds == 1 and st.n_left + st.n_right == len(tasks_all["left"]) + len(tasks_all["right"])
    else:                                            # dead extensions are skipped, in a handful of rounds
        assert 1 < st.n_ext_rounds <= 8 and st.n_left + st.n_right < len(tasks_all["left"]) + len(tasks_all["right"])
    if kw.get("w") == 12:
        assert st.n_retry_left + st.n_retry_right > 0 and (wregs["w"] == 24).sum() > 0     # the band-retry path ran
    if not kw:
        purged = (regs["qb"] == -1) & (regs["qe"] == -1)
        assert purged.sum() > 0 and (~purged).sum() > 0
    b.close()


def test_round_cap_extends_the_rest(rep_toy, monkeypatch):
    """After the round cap everything still undecided is extended at once; regions are the same."""
    g, idx, ix = rep_toy
    b, want, got, ctx = _run(idx, ix, g, _reads(g, 1500, 77), max_occ=60)
    wregs, wreg_off, wseeds = loader.chain2aln(want["chains"], want["seeds"], want["chain_off"], ctx["enc"], ctx["cum"],
                                               ctx["ref"], ctx["l_pac"], opt=ctx["oopt"])
    rounds = []
    for cap, every_round in (("6", True), ("6", False), ("1", False)):
        monkeypatch.setenv("BWAMS_EXT_MAX_ROUNDS", cap)
        if every_round:
            monkeypatch.setenv("BWAMS_EXT_ALL_ROUNDS", "1")       # never cut the rounds short
        else:
            monkeypatch.delenv("BWAMS_EXT_ALL_ROUNDS", raising=False)
        capi.debug_reload()                                       # the switches are read once: say that they changed
        b.extend_run(ctx["gopt"])
        regs, reg_off, aln = b.extend_fetch()
        assert np.array_equal(reg_off, wreg_off) and np.array_equal(aln, wseeds["aln"])
        _assert_regs(regs, wregs, False)
        rounds.append(b.stats().n_ext_rounds)
    # natural convergence; the default (the rest is extended at once when little is left); round 0 + "the rest"
    assert rounds[0] > 2 and 2 <= rounds[1] <= rounds[0] and rounds[2] == 2
    b.close()


def test_chain_upload_then_extend(rep_toy):
    """A caller that chains on the host hands chain_ar over and gets the same regions."""
    g, idx, ix = rep_toy
    b, want, got, ctx = _run(idx, ix, g, _reads(g, 800, 37))
    b.extend_run(ctx["gopt"])
    regs1, off1, aln1 = b.extend_fetch()
    b.close()
    enc, cum = ctx["enc"], ctx["cum"]
    b2 = capi.Batch(ix, len(cum) - 1, int(cum[-1]))
    b2.seed_upload(enc, cum)
    b2.chain_upload(want["chains"], want["seeds"], want["chain_off"])
    b2.extend_run(ctx["gopt"])
    regs2, off2, aln2 = b2.extend_fetch()
    assert np.array_equal(off1, off2) and np.array_equal(aln1, aln2)
    _assert_regs(regs2, regs1, False)
    bad = want["chains"].copy()
    if len(bad):
        bad["seed_off"][0] += 1
        with pytest.raises(capi.BwamsError):
            b2.chain_upload(bad, want["seeds"], want["chain_off"])
    b2.close()


def test_long_reads_rescore_short_seeds(rep_toy):
    """Reads of ~1100 bases and more: mem_flt_chained_seeds re-scores every seed shorter than 200 bases with a
    local SW in a +-50 window and drops the weak ones; chains, seeds (with their new scores) and regions
    equal the oracle's."""
    g, idx, ix = rep_toy
    rng = np.random.default_rng(91)
    reads = []
    for i in range(60):
        L = int(rng.integers(1110, 2200))
        st = int(rng.integers(0, len(g) - L - 1))
        r = g[st:st + L].copy()
        nmut = int(rng.integers(5, 60))
        pos = rng.integers(0, L, size=nmut)
        r[pos] = (r[pos] + rng.integers(1, 4, size=nmut)) & 3
        if i % 3 == 0:                                   # a chimeric tail: short seeds with poor surroundings
            r[L - 300:] = rng.integers(0, 4, size=300)
            k = int(rng.integers(L - 280, L - 60))
            r[k:k + 30] = g[st + k:st + k + 30]
        reads.append(simulate.revcomp(r) if i % 2 else r)
    reads += list(simulate.make_reads(g, 200, seed=5)[0])             # short reads in the same chunk are untouched
    b, want, got, ctx = _run(idx, ix, g, reads)
    _assert_chains(want, got)
    long_chain = np.diff(ctx["cum"])[want["chains"]["seqid"]] >= 1110
    sc, ln = want["seeds"]["score"], want["seeds"]["len"]
    assert (sc != ln).sum() > 0                                          # some seeds carry an SW score
    raw_c, raw_s, _ = loader.chain_seeds(ctx["sm"], ctx["coord"], ctx["off"], ctx["cum"], ctx["l_pac"], opt=ctx["oopt"], do_flt=False)
    assert long_chain.any() and len(want["seeds"]) < int(raw_s["len"].shape[0])
    b.extend_run(ctx["gopt"])
    regs, reg_off, aln = b.extend_fetch()
    wregs, wreg_off, wseeds = loader.chain2aln(want["chains"], want["seeds"], want["chain_off"], ctx["enc"], ctx["cum"],
                                               ctx["ref"], ctx["l_pac"], opt=ctx["oopt"])
    assert np.array_equal(reg_off, wreg_off) and np.array_equal(aln, wseeds["aln"])
    _assert_regs(regs, wregs, False)
    b.close()


FINAL_FIELDS = REG_FIELDS


def _assert_final(b, ctx, want, contigs=None):
    wregs, wreg_off, _ = loader.chain2aln(want["chains"], want["seeds"], want["chain_off"], ctx["enc"], ctx["cum"],
                                          ctx["ref"], ctx["l_pac"], contigs=contigs, opt=ctx["oopt"])
    wfin, wfin_off = loader.regs_finish(wregs, wreg_off, ctx["enc"], ctx["cum"], ctx["ref"], ctx["l_pac"], contigs=contigs,
                                        opt=ctx["oopt"])
    b.extend_run(ctx["gopt"])
    n = b.dedup_run(ctx["gopt"])
    fin, fin_off = b.dedup_fetch()
    assert n == len(wfin) and np.array_equal(fin_off, wfin_off)
    for f in FINAL_FIELDS:
        assert np.array_equal(fin[f], wfin[f]), f
    return wfin, wregs


def test_final_regions_match_oracle(rep_toy):
    """mem_sort_dedup_patch on the device: redundant and identical hits removed, order by (score, rb, qb)."""
    g, idx, ix = rep_toy
    b, want, got, ctx = _run(idx, ix, g, _reads(g, 3000, 61))
    wfin, wregs = _assert_final(b, ctx, want)
    alive = ~((wregs["qb"] == -1) & (wregs["qe"] == -1))
    assert 0 < len(wfin) <= alive.sum()
    assert b.stats().n_final_regs == len(wfin)
    # the extension's own regions are still there
    regs, reg_off, aln = b.extend_fetch()
    assert len(regs) == len(wregs)
    b.close()


@pytest.mark.parametrize("kw", [dict(mask_level_redun=0.5), dict(max_chain_gap=300), dict(w=30), dict(max_occ=50)])
def test_final_regions_options(rep_toy, kw):
    g, idx, ix = rep_toy
    b, want, got, ctx = _run(idx, ix, g, _reads(g, 1500, 63), **kw)
    _assert_final(b, ctx, want)
    b.close()


def test_final_regions_patch_split_alignments(rep_toy):
    """Long reads with a deletion wider than the chaining band but within twice the band: two chains, two regions,
    merged into one by mem_patch_reg's global alignment (score-only ksw_global2, both strands); ALT contigs marked."""
    g, idx, ix = rep_toy
    rng = np.random.default_rng(71)
    reads = []
    for i in range(40):
        st = int(rng.integers(0, len(g) - 6000))
        gap = int(rng.integers(120, 200))
        r = np.concatenate([g[st:st + 2500], g[st + 2500 + gap:st + 5200 + gap]])
        pos = rng.integers(0, len(r), size=20)
        r[pos] = (r[pos] + 1) & 3
        reads.append(simulate.revcomp(r) if i % 2 else r)
    reads += list(simulate.make_reads(g, 300, seed=6)[0])
    contigs = np.zeros(2, capi.CONTIG_DTYPE)
    contigs["offset"], contigs["len"], contigs["is_alt"] = [0, 70000], [70000, len(g) - 70000], [0, 1]
    b, want, got, ctx = _run(idx, ix, g, reads, contigs=contigs)
    wfin, wregs = _assert_final(b, ctx, want, contigs=contigs)
    merged = (wfin["n_comp_is_alt"] & 0x3fffffff) > 1
    assert merged.sum() >= 10 and (wfin["n_comp_is_alt"] >> 30).sum() > 0
    b.close()
    c = np.zeros(1, capi.CONTIG_DTYPE)
    c["len"] = len(g)
    ix.set_contigs(c)


def test_patch_alignments_of_reads_with_many_regions():
    """The wave tier's patch alignment (the (h, e) row in LDS, 64 columns per step): 1000-base reads (the longest it takes) from a tandem
    array of 50 copies of a 900-base unit, 0.5 % diverged — both halves align to many copies, dozens of regions per read reach
    de-duplication — with a block of 20 mismatches in the middle (and, in two thirds of the reads, a 12-base deletion or a 9-base
    insertion beside it) that the extension does not cross at zdrop = 20: every kept copy holds a pair of regions that mem_patch_reg
    aligns and merges."""
    rng = np.random.default_rng(72)
    g = rng.integers(0, 4, size=150000, dtype=np.uint8)
    unit = rng.integers(0, 4, size=900, dtype=np.uint8)
    for c in range(50):
        cp = unit.copy()
        m = rng.random(900) < 0.005
        cp[m] = (cp[m] + rng.integers(1, 4, size=int(m.sum()))) & 3
        g[40000 + 900 * c:40000 + 900 * (c + 1)] = cp
    idx = fmindex.build_fmindex(g)
    ix = capi.Index.from_host(idx, 0)
    reads = []
    for i in range(60):
        st = 40000 + 900 * int(rng.integers(0, 45)) + int(rng.integers(0, 150))
        r = g[st:st + 1000].copy()
        k = i % 3
        if k == 0:
            r[490:510] = (r[490:510] + 2) & 3
        elif k == 1:
            r = np.concatenate([r[:500], r[512:], g[st + 1000:st + 1012]])
            r[470:490] = (r[470:490] + 2) & 3
        else:
            r = np.concatenate([r[:500], rng.integers(0, 4, size=9, dtype=np.uint8), r[500:991]])
            r[470:490] = (r[470:490] + 2) & 3
        pos = rng.integers(0, len(r), size=3)
        r[pos] = (r[pos] + 1) & 3
        reads.append(simulate.revcomp(r) if i % 2 else r)
    b, want, got, ctx = _run(idx, ix, g, reads, zdrop=20)
    wfin, wregs = _assert_final(b, ctx, want)
    regs, reg_off, _ = b.extend_fetch()
    fin, fin_off = b.dedup_fetch()
    slots = np.diff(reg_off)
    merged = np.array([int(((fin["n_comp_is_alt"][fin_off[r]:fin_off[r + 1]] & 0x3fffffff) > 1).sum()) for r in range(len(reads))])
    assert ((merged >= 5) & (slots > 32)).sum() >= 20 and slots.max() > 100, (merged, slots)
    b.close(); ix.close()


def test_pestat_matches_oracle(rep_toy):
    """mem_pestat over the final regions of paired reads: per-pair selection and the sort on the device, the
    reference's percentile / mean / std arithmetic on the host — every field bit-equal to the oracle's."""
    g, idx, ix = rep_toy
    rng = np.random.default_rng(123)
    reads = []
    for i in range(3000):
        ins = int(max(160, rng.normal(420, 45)))
        st = int(rng.integers(0, len(g) - ins - 1))
        frag = g[st:st + ins]
        r1, r2 = frag[:150].copy(), simulate.revcomp(frag[-150:])
        for r in (r1, r2):
            pos = rng.integers(0, 150, size=int(rng.integers(0, 4)))
            r[pos] = (r[pos] + 1) & 3
        if i % 7 == 0:                                 # some pairs in the other orientations / unpaired
            r2 = simulate.revcomp(r2)
        if i % 2:
            r1, r2 = simulate.revcomp(r1), simulate.revcomp(r2)
        reads += [r1, r2]
    for kw in (dict(), dict(max_ins=430), dict(mask_level=0.9)):
        b, want, got, ctx = _run(idx, ix, g, reads, **kw)
        wfin, _ = _assert_final(b, ctx, want)
        wregs, wreg_off, _ = loader.chain2aln(want["chains"], want["seeds"], want["chain_off"], ctx["enc"], ctx["cum"],
                                              ctx["ref"], ctx["l_pac"], opt=ctx["oopt"])
        wfin, wfin_off = loader.regs_finish(wregs, wreg_off, ctx["enc"], ctx["cum"], ctx["ref"], ctx["l_pac"], opt=ctx["oopt"])
        wpes = loader.pestat(wfin, wfin_off, ctx["l_pac"], opt=ctx["oopt"])
        pes = b.pestat(ctx["gopt"])
        for f in ("low", "high", "failed", "avg", "std"):
            assert np.array_equal(pes[f], wpes[f]), (kw, f, pes, wpes)
        if not kw:
            assert wpes["failed"][1] == 0 and 380 < wpes["avg"][1] < 460 and (wpes["failed"] == 0).sum() >= 2
        b.close()


def test_degenerate_inputs_through_the_whole_path(rep_toy):
    """Empty chunk, reads without seeds (all N, too short, random), a single read, an odd read count for mem_pestat."""
    g, idx, ix = rep_toy
    rng = np.random.default_rng(8)
    cases = [
        [],
        [np.full(100, 4, np.uint8)],
        [g[100:110].copy()],
        [rng.integers(0, 4, size=150, dtype=np.uint8)],
        [g[2000:2150].copy()],
        [np.full(60, 4, np.uint8), g[3000:3150].copy(), g[10:25].copy(), rng.integers(0, 4, size=80, dtype=np.uint8),
         simulate.revcomp(g[7000:7200]), np.zeros(150, np.uint8), g[500:519].copy()],
    ]
    for reads in cases:
        if reads:
            enc, cum = simulate.flatten_reads(reads)
        else:
            enc, cum = np.zeros(0, np.uint8), np.zeros(1, np.int64)
        o = loader.OracleFMI(idx)
        sm = o.collect_smem(enc, cum) if reads else np.zeros(0, loader.SMEM_DTYPE)
        coord, off = o.sa_lookup(sm) if len(sm) else (np.zeros(0, np.int64), np.zeros(1, np.int64))
        l_pac = len(g)
        ref = np.concatenate([g, (3 - g[::-1]).astype(np.uint8)])
        oopt, gopt = _mem_opts()
        wch, wsd, wchoff = loader.chain_seeds(sm, coord, off, cum, l_pac, opt=oopt, ref_string=ref, enc=enc)
        wregs, wreg_off, wsd2 = loader.chain2aln(wch, wsd, wchoff, enc, cum, ref, l_pac, opt=oopt)
        wfin, wfin_off = loader.regs_finish(wregs, wreg_off, enc, cum, ref, l_pac, opt=oopt)
        b = capi.Batch(ix, max(len(reads), 1), max(int(cum[-1]), 1))
        b.seed_upload(enc, cum)
        b.seed_run(capi.default_seed_opt(), with_sa=True)
        nc, ns = b.chain_run(gopt)
        ch, sd, choff = b.chain_fetch()
        assert nc == len(wch) and ns == len(wsd) and np.array_equal(choff, wchoff)
        n = b.extend_run(gopt)
        regs, reg_off, aln = b.extend_fetch()
        assert n == len(wregs) and np.array_equal(reg_off, wreg_off)
        _assert_regs(regs, wregs, False)
        assert b.dedup_run(gopt) == len(wfin)
        fin, fin_off = b.dedup_fetch()
        assert np.array_equal(fin_off, wfin_off)
        for f in FINAL_FIELDS:
            assert np.array_equal(fin[f], wfin[f]), f
        pes, wpes = b.pestat(gopt), loader.pestat(wfin, wfin_off, l_pac, opt=oopt)
        for f in ("low", "high", "failed", "avg", "std"):
            assert np.array_equal(pes[f], wpes[f]), f
        b.close()


def test_ert_mode_chaining_matches_oracle(rep_toy):
    """bwams_chain_run_ert: the MEMs / hits an ERT walk would leave (FM-index SMEMs dressed up as such: every hit,
    some stored the way backward search stores them, MEMs shuffled and some duplicated) through the device sort,
    mem_chain_new, the chain filter and on to the final regions — against the oracle's ERT tail, and, the inputs being
    equivalent, against the FM-index path too."""
    from util import ert_mems_from_smems
    g, idx, ix = rep_toy
    reads = _reads(g, 2000, 23, extra=[g[100:118].copy(), np.full(90, 4, np.uint8)])
    enc, cum = simulate.flatten_reads(reads)
    o = loader.OracleFMI(idx)
    sm = o.collect_smem(enc, cum)
    all_coord, all_off = o.sa_lookup(sm, 1 << 30)
    l_pac = len(g)
    ref = np.concatenate([g, (3 - g[::-1]).astype(np.uint8)])
    for trial, (mo, dup) in enumerate(((500, 0.1), (5, 0.0))):
        oopt, gopt = _mem_opts(max_occ=mo)
        mems, mem_off, hits, hit_off = ert_mems_from_smems(sm, all_coord, all_off, len(reads), l_pac, seed=trial, dup_frac=dup)
        want = loader.chain_new_ert(mems, mem_off, hits, hit_off, cum, l_pac, opt=oopt)
        b = capi.Batch(ix, len(reads), int(cum[-1]))
        b.seed_upload(enc, cum)
        nc, ns = b.chain_run_ert(mems, mem_off, hits, hit_off, gopt)
        got = dict(zip(("chains", "seeds", "chain_off"), b.chain_fetch()))
        assert nc == len(want[0]) and ns == len(want[1])
        _assert_chains(dict(chains=want[0], seeds=want[1], chain_off=want[2]), got)
        if dup == 0.0:                                        # equivalent inputs: the FM-index path's chains
            coord, off = o.sa_lookup(sm, mo)
            fm = loader.chain_seeds(sm, coord, off, cum, l_pac, opt=oopt)
            assert np.array_equal(fm[2], want[2]) and np.array_equal(fm[0]["pos"], want[0]["pos"])
            assert (mems["hitcount"] > mo).sum() > 20
        # the rest of the path runs on these chains like on any others
        b.extend_run(gopt)
        regs, reg_off, aln = b.extend_fetch()
        wregs, wreg_off, wseeds = loader.chain2aln(want[0], want[1], want[2], enc, cum, ref, l_pac, opt=oopt)
        assert np.array_equal(reg_off, wreg_off) and np.array_equal(aln, wseeds["aln"])
        _assert_regs(regs, wregs, False)
        b.close()
    # malformed input is refused
    bad = mems.copy()
    bad["hitbeg"][0] = 1 << 20
    b = capi.Batch(ix, len(reads), int(cum[-1]))
    b.seed_upload(enc, cum)
    with pytest.raises(capi.BwamsError):
        b.chain_run_ert(bad, mem_off, hits, hit_off)
    b.close()
