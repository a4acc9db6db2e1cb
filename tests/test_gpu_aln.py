"""bwams_reg2aln_run (mem_reg2aln -> bwa_gen_cigar2 -> ksw_global2 with traceback, on the device; mapping quality on the
host side of the library) against the oracle on the very regions the device holds: every record field, every CIGAR
operation and every MD byte."""
import numpy as np
import pytest

from bwams import capi, simulate
from oracle import loader
from util import toy

pytestmark = pytest.mark.gpu

FIELDS = ("pos", "rid", "flag", "is_rev", "is_alt", "mapq", "NM", "n_cigar", "md_len", "cigar_off", "md_off", "score", "sub", "alt_sc")


def _opts(**kw):
    o, g = loader.default_mem_opt(), capi.default_mem_opt()
    for k, v in kw.items():
        setattr(o, k, v); setattr(g, k, v)
    return o, g


def _check(got, want):
    ga, gc, gm = got
    wa, wc, wm = want
    assert len(ga) == len(wa)
    for f in FIELDS:
        assert np.array_equal(ga[f], wa[f]), (f, np.flatnonzero(ga[f] != wa[f])[:5])
    assert np.array_equal(gc, wc) and np.array_equal(gm, wm)


def _run(g, idx, reads, contigs=None, pairs=False, **kw):
    oopt, gopt = _opts(**kw)
    ix = capi.Index.from_host(idx, 0)
    if contigs is not None:
        ix.set_contigs(contigs)
    enc, cum = simulate.flatten_reads(reads)
    b = capi.Batch(ix, len(reads), int(cum[-1]))
    b.seed_upload(enc, cum)
    b.seed_run(capi.default_seed_opt(), with_sa=True)
    b.chain_run(gopt); b.extend_run(gopt); b.dedup_run(gopt)
    fin, fin_off = b.dedup_fetch()
    l_pac = len(g)
    got = b.reg2aln(gopt, 0)
    want = loader.reg2aln(fin, fin_off, enc, cum, idx.ref_0123, l_pac, contigs=contigs, opt=oopt)
    _check(got, want)
    gapped = sum(1 for c in got[1] if (int(c) & 0xf) in (1, 2))
    res = dict(n=len(fin), gapped=gapped, rev=int(got[0]["is_rev"].sum()))
    if pairs:
        pes = b.pestat(gopt)
        b.pair_run(pes, gopt)
        pregs, poff, _ = b.pair_fetch()
        _check(b.reg2aln(gopt, 1), loader.reg2aln(pregs, poff, enc, cum, idx.ref_0123, l_pac, contigs=contigs, opt=oopt))
        res["n_pe"] = len(pregs)
        res["secondary"] = int((pregs["secondary"] >= 0).sum())
    b.close(); ix.close()
    return res


def test_reg2aln_equals_oracle_on_simulated_reads():
    g, idx = toy()
    reads, _, _ = simulate.make_reads(g, 3000, seed=13)
    r = _run(g, idx, reads)
    assert r["n"] > 2500 and r["gapped"] > 100 and 0.3 < r["rev"] / r["n"] < 0.7


def test_reg2aln_other_scoring_and_contigs():
    g, idx = toy()
    reads, _, _ = simulate.make_reads(g, 1500, seed=14)
    contigs = np.zeros(3, capi.CONTIG_DTYPE)
    contigs["offset"], contigs["len"], contigs["is_alt"] = [0, 9000, 15000], [9000, 6000, len(g) - 15000], [0, 1, 0]
    r = _run(g, idx, reads, contigs=contigs, o_del=4, e_del=2, o_ins=5, e_ins=1, w=30)
    assert r["n"] > 1000 and r["gapped"] > 50


def test_reg2aln_after_mate_rescue():
    g, idx = toy()
    pr = simulate.make_read_pairs(g, 400, seed=6, damaged_frac=0.3)
    r = _run(g, idx, pr, pairs=True)
    assert r["n_pe"] >= r["n"] and r["secondary"] > 0


def test_reg2aln_wide_bands_and_long_reads():
    """Reads of 400 bases with several indels: bands beyond the 32-column ring (the 128-column ring and the HBM row)."""
    g, idx = toy()
    rng = np.random.default_rng(3)
    reads = []
    for _ in range(300):
        p = int(rng.integers(0, len(g) - 500))
        r = list(g[p:p + 400])
        for _k in range(int(rng.integers(1, 5))):
            at = int(rng.integers(30, len(r) - 30))
            ln = int(rng.integers(1, 30))
            if rng.random() < 0.5:
                del r[at:at + ln]
            else:
                r[at:at] = list(rng.integers(0, 4, size=ln))
        a = np.array(r, np.uint8)
        reads.append(simulate.revcomp(a) if rng.random() < 0.5 else a)
    r = _run(g, idx, reads)
    assert r["gapped"] > 200


def test_single_end_mark_primary_then_reg2aln():
    """The single-end SAM side on the device: mem_mark_primary_se(opt, n, a, n_processed + i) of every read (odd read count
    allowed), then mem_reg2aln with the mapping quality of the marked regions."""
    g, idx = toy()
    reads, _, _ = simulate.make_reads(g, 1201, seed=23)
    oopt, gopt = _opts()
    ix = capi.Index.from_host(idx, 0)
    enc, cum = simulate.flatten_reads(reads)
    b = capi.Batch(ix, len(reads), int(cum[-1]))
    b.seed_upload(enc, cum)
    b.seed_run(capi.default_seed_opt(), with_sa=True)
    b.chain_run(gopt); b.extend_run(gopt); b.dedup_run(gopt)
    fin, fin_off = b.dedup_fetch()
    ID0 = 5_000_003
    n = b.mark_primary_se(gopt, id_base=ID0)
    regs, off, _ = b.pair_fetch()
    assert n == len(fin) and np.array_equal(off, fin_off)
    want = fin.copy()
    for r in range(len(reads)):
        w, _n_pri = loader.mark_primary_se(fin[fin_off[r]:fin_off[r + 1]], ID0 + r, oopt)
        want[fin_off[r]:fin_off[r + 1]] = w
    for f in ("rb", "re", "qb", "qe", "score", "sub", "sub_n", "alt_sc", "secondary", "secondary_all", "hash", "n_comp_is_alt"):
        assert np.array_equal(regs[f], want[f]), f
    assert (regs["secondary"] >= 0).sum() > 20 and (regs["sub"] > 0).sum() > 50
    _check(b.reg2aln(gopt, 1), loader.reg2aln(want, fin_off, enc, cum, idx.ref_0123, len(g), opt=oopt))
    b.close(); ix.close()
