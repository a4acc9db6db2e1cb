"""Differential test over random configurations: genome shape, read lengths and error rates, scoring and heuristic
options — the whole path (seeding .. de-duplication, then the paired-end tail) on the GPU against the oracle, bit for
bit.  Small inputs, many configurations: what the fixed-option parity tests cannot reach."""
import os

import numpy as np
import pytest

from bwams import capi, fmindex, simulate
from oracle import loader

pytestmark = pytest.mark.gpu

REG_F = ("rb", "re", "qb", "qe", "rid", "score", "truesc", "sub", "alt_sc", "csub", "sub_n", "w", "seedcov", "secondary",
         "secondary_all", "seedlen0", "n_comp_is_alt", "frac_rep", "hash")


def _config(seed):
    rng = np.random.default_rng(seed)
    a = int(rng.choice([1, 1, 2, 3]))
    b = int(rng.choice([2, 4, 4, 6]))
    kw = dict(
        a=a, o_del=int(rng.choice([4, 6, 8])), e_del=int(rng.choice([1, 2])), o_ins=int(rng.choice([4, 6, 7])),
        e_ins=int(rng.choice([1, 2])), pen_clip5=int(rng.choice([0, 5, 9])), pen_clip3=int(rng.choice([0, 5, 7])),
        w=int(rng.choice([8, 30, 100])), zdrop=int(rng.choice([20, 100, 300])),
        min_seed_len=int(rng.choice([15, 19, 19, 25])), min_chain_weight=int(rng.choice([0, 0, 30])),
        max_occ=int(rng.choice([20, 100, 500])), max_chain_gap=int(rng.choice([200, 10000])),
        mask_level=float(rng.choice([0.3, 0.5, 0.7])), drop_ratio=float(rng.choice([0.3, 0.5, 0.8])),
        mask_level_redun=float(rng.choice([0.8, 0.95])), pen_unpaired=int(rng.choice([5, 17, 40])),
        max_matesw=int(rng.choice([2, 10, 50])), max_ins=int(rng.choice([600, 10000])),
    )
    shape = dict(genome=int(rng.choice([20000, 60000, 150000])), repeat_frac=float(rng.choice([0.05, 0.3, 0.5])),
                 repeat_len=int(rng.choice([60, 250, 700])), n_families=int(rng.choice([1, 3, 6])),
                 read_len=int(rng.choice([60, 100, 151, 250, 250, 1300])), n_pairs=int(rng.choice([150, 300])),
                 fma=(int(rng.integers(4, 9)), int(rng.integers(5, 10))) if rng.random() < 0.35 else None,
                 extend_all=int(rng.random() < 0.3), emf=bool(rng.random() < 0.4),
                 insert=float(rng.choice([260.0, 420.0])), damaged=float(rng.choice([0.1, 0.4])),
                 n_contigs=int(rng.choice([1, 1, 3])))
    return kw, b, shape


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("BWAMS_FUZZ_N", "48")))))
def test_random_configuration(seed):
    kw, bmis, sh = _config(1000 + seed)
    capi.lib()
    L = sh["read_len"]
    if seed % 6 == 5:       # every sixth configuration on the harder genome shape (tandem arrays, microsatellites, poly-A, exact duplications)
        g = simulate.make_genome(sh["genome"], seed=seed, profile="grch38_like")
    else:
        g = simulate.make_genome(sh["genome"], seed=seed, repeat_frac=sh["repeat_frac"], repeat_len=sh["repeat_len"],
                                 n_families=sh["n_families"])
    idx = fmindex.build_fmindex(g)
    ix = capi.Index.from_host(idx, 0)
    l_pac = len(g)
    contigs = None
    if sh["n_contigs"] == 3:
        contigs = np.zeros(3, capi.CONTIG_DTYPE)
        c1, c2 = l_pac // 3, 2 * l_pac // 3
        contigs["offset"] = [0, c1, c2]
        contigs["len"] = [c1, c2 - c1, l_pac - c2]
        contigs["is_alt"] = [0, 1, 0]
        ix.set_contigs(contigs)
    if L > 1000:
        sh["n_pairs"] = 40                                     # long reads: mem_flt_chained_seeds re-scores short seeds
    reads = simulate.make_read_pairs(g, sh["n_pairs"], seed=seed + 50, read_len=L, insert_mean=max(sh["insert"], L + 40.0),
                                     insert_sd=25.0, damaged_frac=sh["damaged"], discordant_frac=0.08)
    rng = np.random.default_rng(seed)
    for r in reads[::17]:                                      # a few Ns
        r[rng.integers(0, L)] = 4
    enc, cum = simulate.flatten_reads(reads)
    oopt, gopt = loader.default_mem_opt(kw["a"], bmis), capi.default_mem_opt(kw["a"], bmis)
    for k, v in kw.items():
        setattr(oopt, k, v)
        setattr(gopt, k, v)
    so, sg = loader.default_seed_opt(), capi.default_seed_opt()
    so.max_occ = sg.max_occ = kw["max_occ"]
    so.min_seed_len = sg.min_seed_len = kw["min_seed_len"]
    so.split_factor = sg.split_factor = float(rng.choice([1.2, 1.5, 2.0]))
    so.split_width = sg.split_width = int(rng.choice([2, 10, 40]))
    so.max_mem_intv = sg.max_mem_intv = int(rng.choice([0, 20, 50]))
    ref = np.concatenate([g, (3 - g[::-1]).astype(np.uint8)])
    oopt.extend_all = gopt.extend_all = sh["extend_all"]
    # exact-match filter first (when drawn): the reads it resolves are skipped by seeding and get their regions from
    # mem_perfect2reg instead; undamaged ends of a simulated pair carry errors, so a few exact copies are added
    skip = None
    b = capi.Batch(ix, len(reads) + 16, int(cum[-1]) + 16 * L)
    if sh["emf"] and L <= 250 and sh["genome"] <= 60000:
        from bwams import emf
        exact = [g[p:p + L].copy() for p in rng.integers(0, l_pac - L, size=8)] + \
                [simulate.revcomp(g[p:p + L]) for p in rng.integers(0, l_pac - L, size=8)]
        reads = reads[:-16] + exact if len(reads) > 16 else reads
        enc, cum = simulate.flatten_reads(reads)
        tab = emf.build_emf(g, L)
        e = capi.Emf(ix, table=tab)
        oe = loader.OracleEMF(tab, idx.ref_0123)
        b.seed_upload(enc, cum)
        b.emf_run(e)
        perfect, code = b.emf_fetch(len(reads))
        wantp = oe.probe_many(list(reads))
        assert np.array_equal(code, wantp[:, 0].astype(np.uint8))
        hit = (code == 3) | (code == 4)
        assert np.array_equal(perfect[hit], wantp[hit, 1:].astype(np.uint32)) and hit.sum() >= 16
        eregs, eoff, erev = b.emf_regs(e, gopt)
        for r_ in np.flatnonzero(hit):
            wr, wrev = oe.perfect2reg(reads[r_], int(perfect[r_, 0]), int(perfect[r_, 1]), l_pac, contigs=contigs, opt=oopt)
            ge = eregs[eoff[r_]:eoff[r_ + 1]]
            assert len(ge) == len(wr) and erev[r_] == wrev
            for f in ("rb", "re", "qb", "qe", "rid", "score", "truesc", "seedcov", "w", "n_comp_is_alt"):
                assert np.array_equal(ge[f], wr[f]), f
        assert eoff[-1] == sum(eoff[r_ + 1] - eoff[r_] for r_ in np.flatnonzero(hit))
        skip = hit.astype(np.uint8)
        e.close()
    # oracle
    o = loader.OracleFMI(idx)
    if sh["fma"]:                                              # FMA tables at random (shallow) depths, built on both sides
        o.build_fma(*sh["fma"])
        ix.build_fma(*sh["fma"])
    sm = o.collect_smem(enc, cum, so, skip=skip)
    coord, off = o.sa_lookup(sm, so.max_occ)
    wch, wsd, wchoff = loader.chain_seeds(sm, coord, off, cum, l_pac, contigs=contigs, opt=oopt, ref_string=ref, enc=enc)
    wregs, wreg_off, _ = loader.chain2aln(wch, wsd, wchoff, enc, cum, ref, l_pac, contigs=contigs, opt=oopt)
    wfin, wfin_off = loader.regs_finish(wregs, wreg_off, enc, cum, ref, l_pac, contigs=contigs, opt=oopt)
    wpes = loader.pestat(wfin, wfin_off, l_pac, opt=oopt)
    # device (an EMF run above left its skip flags in the batch)
    if skip is None:
        b.seed_upload(enc, cum)
    b.seed_run(sg, with_sa=True)
    gsm, gcoord, goff = b.seed_fetch()
    assert len(gsm) == len(sm) and np.array_equal(gcoord, coord) and np.array_equal(goff, off)
    b.chain_run(gopt)
    ch, sd, choff = b.chain_fetch()
    assert np.array_equal(choff, wchoff)
    for f in ("n", "rid", "w_kept_alt", "frac_rep", "pos", "first"):
        assert np.array_equal(ch[f], wch[f]), f
    b.extend_run(gopt)
    regs, reg_off, _ = b.extend_fetch()
    assert np.array_equal(reg_off, wreg_off)
    purged = (wregs["qb"] == -1) & (wregs["qe"] == -1)
    assert np.array_equal((regs["qb"] == -1) & (regs["qe"] == -1), purged)
    keep = slice(None) if sh["extend_all"] else ~purged
    for f in ("rb", "re", "qb", "qe", "score", "truesc", "w", "seedcov"):
        assert np.array_equal(regs[f][keep], wregs[f][keep]), f
    assert b.dedup_run(gopt) == len(wfin)
    fin, fin_off = b.dedup_fetch()
    assert np.array_equal(fin_off, wfin_off)
    for f in ("rb", "re", "qb", "qe", "rid", "score", "truesc", "w", "seedcov", "n_comp_is_alt", "sub", "csub"):
        assert np.array_equal(fin[f], wfin[f]), f
    pes = b.pestat(gopt)
    assert np.array_equal(pes, wpes)
    if L <= 512 and (seed % 3 == 1 or os.environ.get("BWAMS_FUZZ_SAM")):
        # the same chunk as single-end reads: mem_mark_primary_se, the restricted mem_reg2aln, the SAM text (exact-match records for the
        # reads the EMF resolved), random text options
        names_c = [b"ctg%d" % i for i in range(sh["n_contigs"])]
        ix.set_contig_names(names_c)
        sflag = int(rng.choice([0, 0, 0x8, 0x200, 0x10, 0x1000, 0x800, 0x1800]))
        sso, ssg = loader.default_sam_opt(sflag, b"grp" if seed % 2 else b""), capi.default_sam_opt(sflag, b"grp" if seed % 2 else b"")
        sso.T = ssg.T = int(rng.choice([30, 30, 10, 60]))
        sso.max_XA_hits = ssg.max_XA_hits = int(rng.choice([5, 1, 50]))
        quals = rng.integers(33, 74, size=len(enc), dtype=np.uint8)
        rnames = [b"s%d" % i for i in range(len(reads))]
        comments = [b"c:Z:%d" % i if i % 3 == 0 else None for i in range(len(reads))]
        b.mark_primary_se(gopt, id_base=seed * 100, sopt=ssg)
        sregs, soff, _ = b.pair_fetch()
        wmark = wfin.copy()
        for r_ in range(len(reads)):
            a_, e_ = int(wfin_off[r_]), int(wfin_off[r_ + 1])
            if e_ > a_:
                wmark[a_:e_] = loader.mark_primary_se(wfin[a_:e_], seed * 100 + r_, oopt, primary5_T=int(sso.T) if sflag & 0x800 else -1)[0]
        assert np.array_equal(soff, wfin_off)
        for f in ("rb", "qb", "score", "secondary", "secondary_all", "sub", "sub_n", "hash"):
            assert np.array_equal(sregs[f], wmark[f]), f
        b.reg2aln_sam(gopt, ssg, fetch=False)
        b.sam_upload(rnames, quals, comments)
        if skip is not None:
            e2 = capi.Emf(ix, table=tab)
            b.emf_run(e2); b.emf_regs(e2, gopt)
            b.sam_run_emf(e2, gopt, ssg)
        else:
            b.sam_run(gopt, ssg)
        text, roff, _ = b.sam_fetch()
        wtext = loader.reg2sam_se(wmark, wfin_off, enc, cum, ref, l_pac, rnames, quals=quals, comments=comments, contigs=contigs,
                                  contig_names=names_c, opt=oopt, sopt=sso)
        for r_, w_ in enumerate(wtext):
            if skip is not None and skip[r_]:
                wr, _ = oe.perfect2reg(reads[r_], int(perfect[r_, 0]), int(perfect[r_, 1]), l_pac, contigs=contigs, opt=oopt)
                w_ = loader.perfect2sam(wr, reads[r_], l_pac, L, rnames[r_], qual=bytes(quals[cum[r_]:cum[r_ + 1]]), comment=comments[r_],
                                        contigs=contigs, contig_names=names_c, opt=oopt, sopt=sso)
            assert text[roff[r_]:roff[r_ + 1]] == w_, (r_, sflag)
        if skip is not None:
            e2.close()
    if L <= 512 and all(p["failed"] or p["high"] - p["low"] + L <= 20000 for p in pes):
        for use_ert in (False, True):
            wout, wout_off, wpairs = loader.pair_pe(wfin, wfin_off, enc, cum, ref, l_pac, wpes, contigs=contigs, opt=oopt,
                                                    id_base=seed * 1000, use_ert=use_ert)
            n, _ = b.pair_run(pes, gopt, id_base=seed * 1000, use_ert=use_ert)
            out, out_off, pairs = b.pair_fetch()
            assert n == len(wout) and np.array_equal(out_off, wout_off), use_ert
            assert np.array_equal(pairs, wpairs), use_ert
            for f in REG_F:
                assert np.array_equal(out[f], wout[f]), (use_ert, f)
            if not use_ert and (seed % 3 == 0 or os.environ.get("BWAMS_FUZZ_SAM")):
                # the paired-end SAM text of these very regions, with random text options, after aligning only what it reads
                names_c = [b"ctg%d" % i for i in range(sh["n_contigs"])]
                ix.set_contig_names(names_c)
                sflag = int(rng.choice([0, 0, 0x8, 0x200, 0x10, 0x1000]))
                pflag = int(rng.choice([0, 0, 0x4, 0x800, 0x804, 0x1800, 0x20]))      # `mem -P`, `-5`, `-S`: they act in bwams_pair_run_sam
                sflag |= pflag
                sso, ssg = loader.default_sam_opt(sflag, b"rg" if seed % 2 else b""), capi.default_sam_opt(sflag, b"rg" if seed % 2 else b"")
                sso.T = ssg.T = int(rng.choice([30, 30, 10, 60]))
                sso.max_XA_hits = ssg.max_XA_hits = int(rng.choice([5, 1, 50]))
                if pflag:
                    wout, wout_off, wpairs = loader.pair_pe(wfin, wfin_off, enc, cum, ref, l_pac, wpes, contigs=contigs, opt=oopt, id_base=seed * 1000,
                                                            no_rescue=bool(pflag & 0x20), no_pairing=bool(pflag & 0x4),
                                                            primary5_T=int(sso.T) if pflag & 0x800 else -1)
                    b.pair_run(pes, gopt, id_base=seed * 1000, sopt=ssg)
                    out, out_off, pairs = b.pair_fetch()
                    assert np.array_equal(out_off, wout_off) and np.array_equal(pairs, wpairs), pflag
                    for f in REG_F:
                        assert np.array_equal(out[f], wout[f]), (pflag, f)
                quals = rng.integers(33, 74, size=len(enc), dtype=np.uint8)
                rnames = [b"p%d" % (i // 2) for i in range(len(reads))]
                b.reg2aln_sam(gopt, ssg, pes=pes, fetch=False)
                b.sam_upload(rnames, quals)
                b.sam_run_pe(pes, gopt, ssg)
                text, roff, _ = b.sam_fetch()
                wtext = loader.sam_pe(wout, wout_off, enc, cum, ref, l_pac, wpes, wpairs, rnames, quals=quals, contigs=contigs,
                                      contig_names=names_c, opt=oopt, sopt=sso)
                for r_, w_ in enumerate(wtext):
                    assert text[roff[r_]:roff[r_ + 1]] == w_, (r_, sflag)
    # ERT mode's way into chaining, fed the same seeds dressed up as an ERT walk's output
    from util import ert_mems_from_smems
    all_coord, all_off = o.sa_lookup(sm, 1 << 30)
    mems, mem_off, hits, hit_off = ert_mems_from_smems(sm, all_coord, all_off, len(reads), l_pac, seed=seed, dup_frac=0.05)
    wech = loader.chain_new_ert(mems, mem_off, hits, hit_off, cum, l_pac, contigs=contigs, opt=oopt, ref_string=ref, enc=enc)
    b.chain_run_ert(mems, mem_off, hits, hit_off, gopt)
    ech, esd, echoff = b.chain_fetch()
    assert np.array_equal(echoff, wech[2])
    for f in ("n", "rid", "w_kept_alt", "frac_rep", "pos", "first"):
        assert np.array_equal(ech[f], wech[0][f]), f
    for f in ("rbeg", "qbeg", "len", "score"):
        assert np.array_equal(esd[f], wech[1][f]), f
    b.close()
    ix.close()
