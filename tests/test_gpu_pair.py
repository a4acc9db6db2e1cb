"""GPU parity for the paired-end tail (mate rescue, mem_mark_primary_se, mem_pair): the HIP path through the C-ABI
against the CPU oracle on the same simulated read pairs.  Integer outputs, bit-exact; the insert-size term of
mem_pair goes through log / erfc in double precision and is rounded to an int (tolerance: none, see DESIGN)."""
import numpy as np
import pytest

from bwams import capi, fmindex, simulate
from oracle import loader

pytestmark = pytest.mark.gpu

REG_FIELDS = ("rb", "re", "qb", "qe", "rid", "score", "truesc", "sub", "alt_sc", "csub", "sub_n", "w", "seedcov", "secondary",
              "secondary_all", "seedlen0", "n_comp_is_alt", "frac_rep", "hash", "flg")


@pytest.fixture(scope="module")
def pe_toy():
    capi.lib()
    g = simulate.make_genome(150000, seed=41, repeat_frac=0.30, repeat_len=300, n_families=3)
    idx = fmindex.build_fmindex(g)
    ix = capi.Index.from_host(idx, 0)
    yield g, idx, ix
    ix.close()


def _gpu_final(ix, reads, gopt, contigs=None):
    enc, cum = simulate.flatten_reads(reads)
    b = capi.Batch(ix, len(reads), int(cum[-1]))
    b.seed_upload(enc, cum)
    b.seed_run(capi.default_seed_opt(), with_sa=True)
    b.chain_run(gopt)
    b.extend_run(gopt)
    b.dedup_run(gopt)
    return b, enc, cum


def _compare(g, ix, reads, pes=None, contigs=None, id_base=0, no_rescue=False, use_ert=False, sflag=0, T=30, **kw):
    oopt, gopt = loader.default_mem_opt(), capi.default_mem_opt()
    for k, v in kw.items():
        setattr(oopt, k, v)
        setattr(gopt, k, v)
    l_pac = len(g)
    ref = np.concatenate([g, (3 - g[::-1]).astype(np.uint8)])
    if contigs is not None:
        ix.set_contigs(contigs)
    b, enc, cum = _gpu_final(ix, reads, gopt)
    fin, fin_off = b.dedup_fetch()
    if pes is None:
        pes = b.pestat(gopt)
        assert np.array_equal(pes, loader.pestat(fin, fin_off, l_pac, opt=oopt))
        keys = b.pestat_keys(gopt)                       # the two halves a sharded chunk uses
        assert np.array_equal(keys, np.sort(loader.pestat_keys(fin, fin_off, l_pac, opt=oopt)))
        assert np.array_equal(capi.pestat_from_keys(keys[::-1]), pes)
    want_regs, want_off, want_pairs = loader.pair_pe(fin, fin_off, enc, cum, ref, l_pac, pes, contigs=contigs, opt=oopt,
                                                     id_base=id_base, no_rescue=no_rescue or bool(sflag & 0x20), use_ert=use_ert,
                                                     no_pairing=bool(sflag & 0x4), primary5_T=T if sflag & 0x800 else -1)
    sopt = None
    if sflag:
        sopt = capi.default_sam_opt(sflag)
        sopt.T = T
    n, n_tasks = b.pair_run(pes, gopt, id_base=id_base, no_rescue=no_rescue, use_ert=use_ert, sopt=sopt)
    regs, off, pairs = b.pair_fetch()
    assert n == len(want_regs) and np.array_equal(off, want_off)
    for f in ("score", "sub", "n_sub", "z", "n_pri", "n_matesw"):
        assert np.array_equal(pairs[f], want_pairs[f]), f
    for f in REG_FIELDS:
        assert np.array_equal(regs[f], want_regs[f]), f
    st = b.stats()
    assert st.n_pair_tasks == n_tasks and st.n_pair_regs == n
    print(f"pairs {len(pairs)}: {n_tasks} rescue alignments, {st.n_pair_redone} reads redone, {st.ms_pair:.2f} ms")
    b.close()
    return dict(pairs=pairs, regs=regs, off=off, fin_off=fin_off, n_tasks=n_tasks, pes=pes, redone=st.n_pair_redone)


def test_pair_matches_oracle(pe_toy):
    g, idx, ix = pe_toy
    reads = simulate.make_read_pairs(g, 1500, seed=3)
    r = _compare(g, ix, reads)
    assert r["pes"]["failed"].tolist() == [1, 0, 1, 1]
    assert r["n_tasks"] > 50 and r["pairs"]["n_matesw"].sum() > 50
    assert (np.diff(r["off"]) > np.diff(r["fin_off"])).sum() > 20           # rescued regions were added
    assert (r["pairs"]["score"] > 0).mean() > 0.8


def test_pair_all_orientations_and_ids(pe_toy):
    """Caller-provided statistics with every orientation allowed (four windows per anchor) and a non-zero id base."""
    g, idx, ix = pe_toy
    reads = simulate.make_read_pairs(g, 600, seed=4, damaged_frac=0.4, discordant_frac=0.2)
    pes = np.zeros(4, capi.PESTAT_DTYPE)
    pes["low"], pes["high"], pes["avg"], pes["std"] = 150, 700, 400.0, 45.0
    r = _compare(g, ix, reads, pes=pes, id_base=123456)
    assert r["n_tasks"] > 400


def test_pair_no_rescue_and_few_anchors(pe_toy):
    g, idx, ix = pe_toy
    reads = simulate.make_read_pairs(g, 400, seed=5, damaged_frac=0.3)
    r = _compare(g, ix, reads, no_rescue=True)
    assert r["n_tasks"] == 0 and r["pairs"]["n_matesw"].sum() == 0
    _compare(g, ix, reads, max_matesw=1, pen_unpaired=3)


def _chimeric(g, reads, seed, every=3, left=55):
    """Every `every`-th read becomes a split read: `left` bases of one place, then the rest from another — the longer,
    higher-scoring part on the right, so the primary hit is not the leftmost (what mem_reorder_primary5 looks for)."""
    rng = np.random.default_rng(seed)
    reads = [np.array(r, copy=True) for r in reads]
    n_made = 0
    for i in range(0, len(reads), every):
        L = len(reads[i])
        a, c = int(rng.integers(0, len(g) - L)), int(rng.integers(0, len(g) - L))
        rd = np.concatenate([g[a:a + left], g[c:c + L - left]]).astype(np.uint8)
        reads[i] = (3 - rd[::-1]).astype(np.uint8) if rng.random() < 0.5 else rd
        n_made += 1
    return reads, n_made


def test_pair_primary5_nopairing_norescue_flags(pe_toy):
    """bwams_pair_run_sam: `mem -5` (mem_reorder_primary5 between the marking and mem_pair), `mem -P` (no mem_pair), `mem -S`."""
    g, idx, ix = pe_toy
    reads, n_made = _chimeric(g, simulate.make_read_pairs(g, 600, seed=31), seed=32)
    base = _compare(g, ix, reads)
    r5 = _compare(g, ix, reads, sflag=0x800)
    moved = (base["regs"]["rb"] != r5["regs"]["rb"]).sum()
    assert moved >= n_made // 2, (moved, n_made)          # a reverse-complemented split read has its long part on the left already
    assert np.array_equal(np.sort(base["regs"]["hash"]), np.sort(r5["regs"]["hash"]))       # the same regions, another order
    rp = _compare(g, ix, reads, pes=base["pes"], sflag=0x4)
    assert (rp["pairs"]["score"] == 0).all() and (rp["pairs"]["z"] == -1).all() and np.array_equal(rp["pairs"]["n_pri"], base["pairs"]["n_pri"])
    rs = _compare(g, ix, reads, pes=base["pes"], sflag=0x20 | 0x800, T=20)
    assert rs["n_tasks"] == 0
    _compare(g, ix, reads, pes=base["pes"], sflag=0x800, T=200)                              # nothing scores >= T: no reorder
    _compare(g, ix, reads, pes=base["pes"], sflag=0x800 | 0x4, use_ert=True)


def test_pair_degenerate(pe_toy):
    g, idx, ix = pe_toy
    rng = np.random.default_rng(1)
    # ends without any hit, an end of Ns, a pair whose ends map to the same place on the same strand
    reads = [rng.integers(0, 4, size=150, dtype=np.uint8), rng.integers(0, 4, size=150, dtype=np.uint8),
             np.full(100, 4, np.uint8), g[5000:5150].copy(),
             g[9000:9150].copy(), g[9000:9150].copy(),
             g[20000:20150].copy(), simulate.revcomp(g[20250:20400])]
    pes = np.zeros(4, capi.PESTAT_DTYPE)
    pes["failed"] = [1, 0, 1, 1]
    pes["low"], pes["high"], pes["avg"], pes["std"] = 100, 700, 400.0, 50.0
    r = _compare(g, ix, reads, pes=pes)
    assert r["pairs"]["score"][3] > 0 and r["pairs"]["score"][0] == 0


def test_pair_second_pass(pe_toy, monkeypatch):
    """The reference's _post aligns on the spot when _pre did not batch a window it turns out to need; here such a
    read is redone in a second pass with every orientation planned.  Forced for every read with a test knob."""
    g, idx, ix = pe_toy
    reads = simulate.make_read_pairs(g, 500, seed=6, damaged_frac=0.3, discordant_frac=0.1)
    monkeypatch.setenv("BWAMS_PAIR_DROP_PLAN", "1")
    capi.debug_reload()
    r = _compare(g, ix, reads)
    assert r["redone"] > 30 and r["n_tasks"] > 30


def test_pair_with_alt_contigs(pe_toy):
    """Three sequences, the middle one ALT: rescue windows are clipped to the anchor's sequence, ALT hits sort behind
    the primary assembly in mem_mark_primary_se (its second pass) and stay out of mem_pair."""
    g, idx, ix = pe_toy
    l_pac = len(g)
    contigs = np.zeros(3, capi.CONTIG_DTYPE)
    contigs["offset"] = [0, 50000, 100000]
    contigs["len"] = [50000, 50000, l_pac - 100000]
    contigs["is_alt"] = [0, 1, 0]
    reads = simulate.make_read_pairs(g, 900, seed=8, damaged_frac=0.3)
    # pairs straddling a sequence boundary
    for p in (49800, 49900, 99750, 99950):
        reads += [g[p:p + 150].copy(), simulate.revcomp(g[p + 250:p + 400])]
    try:
        r = _compare(g, ix, reads, contigs=contigs)
        n_fin = np.diff(r["off"])
        assert (r["pairs"]["n_pri"].ravel() < n_fin).sum() > 50          # reads with ALT hits
        assert ((r["pairs"]["n_pri"].ravel() > 0) & (r["pairs"]["n_pri"].ravel() < n_fin)).sum() > 0
    finally:
        c = np.zeros(1, capi.CONTIG_DTYPE)
        c["len"] = l_pac
        ix.set_contigs(c)


def test_pair_ert_variant(pe_toy):
    """useErt: mem_matesw_batch_post_ert (list kept sorted by end, mem_dedup_patch, closing sort) — inferred statistics,
    and caller-given ones with every orientation allowed (several alignments per anchor, end-position ties)."""
    g, idx, ix = pe_toy
    reads = simulate.make_read_pairs(g, 900, seed=12, damaged_frac=0.35, discordant_frac=0.15)
    r = _compare(g, ix, reads, use_ert=True)
    assert r["pairs"]["n_matesw"].sum() > 100
    pes = np.zeros(4, capi.PESTAT_DTYPE)
    pes["low"], pes["high"], pes["avg"], pes["std"] = 150, 700, 400.0, 45.0
    r = _compare(g, ix, reads[:800], pes=pes, use_ert=True, id_base=77)
    assert r["n_tasks"] > 500
    _compare(g, ix, reads[:400], use_ert=True, no_rescue=True)
