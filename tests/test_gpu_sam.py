"""bwams_sam_run — the single-end SAM text on the device (mem_reg2sam + mem_gen_alt + mem_aln2sam, mapping quality included) —
against the oracle's restatement on the very regions the device holds: byte for byte, read by read."""
import numpy as np
import pytest

from bwams import capi, simulate
from oracle import loader
from test_oracle_sam import repeat_genome

pytestmark = pytest.mark.gpu


def _pipeline(n_reads, seed, contigs=None, contig_names=None, read_len=None, split_every=0, mark_flag=0, **optkw):
    g, idx, starts = repeat_genome()
    oopt, gopt = loader.default_mem_opt(), capi.default_mem_opt()
    for k, v in optkw.items():
        setattr(oopt, k, v); setattr(gopt, k, v)
    reads, _, _ = simulate.make_reads(g, n_reads, seed=seed)
    rng = np.random.default_rng(seed + 100)
    for i in range(0, n_reads, 3):
        st = starts[int(rng.integers(0, len(starts)))] + int(rng.integers(0, 500 - reads.shape[1]))
        rd = g[st:st + reads.shape[1]].copy()
        reads[i] = (3 - rd[::-1]).astype(np.uint8) if rng.random() < 0.5 else rd
    reads = [r for r in reads]
    for i in (range(1, n_reads, split_every) if split_every else ()):  # split reads: 55 bases of one place, the rest from another
        a_, c_ = int(rng.integers(0, len(g) - 150)), int(rng.integers(0, len(g) - 150))
        rd = np.concatenate([g[a_:a_ + 55], g[c_:c_ + len(reads[i]) - 55]]).astype(np.uint8)
        reads[i] = (3 - rd[::-1]).astype(np.uint8) if rng.random() < 0.5 else rd
    reads[5] = rng.integers(0, 4, size=150, dtype=np.uint8)            # unalignable
    reads[6] = np.full(150, 4, np.uint8)                               # all N
    ix = capi.Index.from_host(idx, 0)
    if contigs is not None:
        ix.set_contigs(contigs)
    names_c = contig_names or [b"chr1"]
    ix.set_contig_names(names_c)
    enc, cum = simulate.flatten_reads(np.stack(reads))
    b = capi.Batch(ix, len(reads), int(cum[-1]))
    b.seed_upload(enc, cum)
    b.seed_run(capi.default_seed_opt(), with_sa=True)
    b.chain_run(gopt); b.extend_run(gopt); b.dedup_run(gopt)
    ID0 = 77_000
    b.mark_primary_se(gopt, id_base=ID0, sopt=capi.default_sam_opt(mark_flag) if mark_flag else None)
    regs, off, _ = b.pair_fetch()
    if mark_flag & 0x800:                                              # mem_reorder_primary5 after the marking (bwamem.cpp:1840)
        fin, fin_off = b.dedup_fetch()
        n_moved = 0
        for r in range(len(reads)):
            w, _ = loader.mark_primary_se(fin[fin_off[r]:fin_off[r + 1]], ID0 + r, oopt, primary5_T=30)
            w0, _ = loader.mark_primary_se(fin[fin_off[r]:fin_off[r + 1]], ID0 + r, oopt)
            n_moved += int(not np.array_equal(w, w0))
            assert np.array_equal(regs[off[r]:off[r + 1]], w), r
        assert n_moved >= n_reads // (split_every or n_reads) // 4, n_moved
    aln, cig, md = b.reg2aln(gopt, 1)
    quals = rng.integers(33, 74, size=len(enc), dtype=np.uint8)
    names = [b"r%d/x" % i for i in range(len(reads))]
    comments = [b"BC:Z:%d" % i if i % 4 == 0 else None for i in range(len(reads))]
    return dict(g=g, idx=idx, ix=ix, b=b, enc=enc, cum=cum, regs=regs, off=off, aln=aln, quals=quals, names=names, comments=comments,
                oopt=oopt, gopt=gopt, contigs=contigs, contig_names=names_c)


def _compare(c, flag=0, rg=b"", quals=True, comments=True, T=None, annos=None):
    import contextlib
    b = c["b"]
    if annos is not None:
        c["ix"].set_contig_annos(annos)
    so, sg = loader.default_sam_opt(flag, rg), capi.default_sam_opt(flag, rg)
    if T is not None:
        so.T = sg.T = T
    b.sam_upload(c["names"], c["quals"] if quals else None, c["comments"] if comments else None)
    nbytes = b.sam_run(c["gopt"], sg)
    text, roff, mq = b.sam_fetch(len(c["regs"]))
    with (loader.contig_annos(annos) if annos is not None else contextlib.nullcontext()):
        want = loader.reg2sam_se(c["regs"], c["off"], c["enc"], c["cum"], c["idx"].ref_0123, len(c["g"]), c["names"],
                                 quals=c["quals"] if quals else None, comments=c["comments"] if comments else None, contigs=c["contigs"],
                                 contig_names=c["contig_names"], opt=c["oopt"], sopt=so)
    assert nbytes == sum(len(w) for w in want) == roff[-1] and roff[0] == 0
    for r, w in enumerate(want):
        got = text[roff[r]:roff[r + 1]]
        assert got == w, (r, got, w)
    # the device's mem_approx_mapq_se equals the host's (bwams_reg2aln_fetch) on every region
    assert np.array_equal(mq, c["aln"]["mapq"])
    return text


def test_sam_text_equals_oracle():
    c = _pipeline(900, 5)
    text = _compare(c)
    assert text.count(b"\tXA:Z:") > 20 and text.count(b"\tSA:Z:") >= 0 and b"\t4\t*\t0\t0\t*\t" in text
    assert text.count(b"\n") >= 900
    for flag, rg in ((0x8, b"grp"), (0x200, b""), (0x10, b""), (0x1000 | 0x10 | 0x200, b"lane7")):
        _compare(c, flag, rg)
    _compare(c, quals=False, comments=False)
    _compare(c, T=10 ** 6)                       # nothing passes: every read unaligned
    _compare(c, T=0)
    c["b"].close(); c["ix"].close()


def test_sam_text_contigs_alt_and_scoring():
    g, _, _ = repeat_genome()
    l_pac = len(g)
    contigs = np.zeros(3, capi.CONTIG_DTYPE)
    cut1, cut2 = l_pac // 2, l_pac * 3 // 4
    contigs["offset"] = [0, cut1, cut2]
    contigs["len"] = [cut1, cut2 - cut1, l_pac - cut2]
    contigs["is_alt"] = [0, 0, 1]
    c = _pipeline(700, 9, contigs=contigs, contig_names=[b"chrA", b"chrB_longer_name", b"chrB_alt"])
    text = _compare(c)
    assert b"chrB_alt" in text and b"\tpa:f:" in text
    _compare(c, 0x8)
    with pytest.raises(capi.BwamsError, match="annotations"):
        c["b"].sam_run(c["gopt"], capi.default_sam_opt(0x100))        # `mem -V` before the annotations are known
    tv = _compare(c, 0x100, annos=[b"first half\tof the toy", b"", b"alternate locus"])          # MEM_F_REF_HDR
    assert tv.count(b"\tXR:Z:first half of the toy\n") > 100 and tv.count(b"\tXR:Z:alternate locus\n") > 10
    assert b"XR:Z:\n" not in tv and sum(b"XR:Z:" not in ln for ln in tv.split(b"\n")[:-1]) > 50      # chrB_longer_name has none
    assert b"XR:Z:" not in _compare(c, 0x8, annos=[b"a", b"b", b"c"])
    c["b"].close(); c["ix"].close()
    c = _pipeline(500, 13, a=2, b=5, o_del=7, e_del=2, mapq_coef_len=0)
    _compare(c)
    c["b"].close(); c["ix"].close()


def test_sam_text_primary5():
    """`mem -5` (MEM_F_PRIMARY5 | MEM_F_KEEP_SUPP_MAPQ): the leftmost primary hit of a split read leads its records."""
    c0 = _pipeline(600, 21, split_every=4)
    t0 = _compare(c0, 0x1000)
    c0["b"].close(); c0["ix"].close()
    c = _pipeline(600, 21, split_every=4, mark_flag=0x800)
    t5 = _compare(c, 0x800 | 0x1000)
    assert t5 != t0 and t5.count(b"\tSA:Z:") == t0.count(b"\tSA:Z:") > 100
    _compare(c, 0x800)
    _compare(c, 0x800 | 0x8 | 0x200)
    c["b"].close(); c["ix"].close()


def test_sam_text_call_order_and_unsupported_flags():
    c = _pipeline(64, 3)
    b = c["b"]
    b.sam_upload(c["names"], c["quals"])
    with pytest.raises(capi.BwamsError):
        b.sam_run(c["gopt"], capi.default_sam_opt(0x2000))           # MEM_F_XB
    with pytest.raises(capi.BwamsError):
        b.sam_run(c["gopt"], capi.default_sam_opt(0x2))              # MEM_F_PE is the caller's
    b.dedup_run(c["gopt"])
    b.reg2aln(c["gopt"], 0)                                          # regions without mem_mark_primary_se
    with pytest.raises(capi.BwamsError):
        b.sam_run(c["gopt"], capi.default_sam_opt())
    b.close(); c["ix"].close()


def _pe_pipeline(n_pairs, seed, pair_flag=0, **optkw):
    g, idx, starts = repeat_genome()
    oopt, gopt = loader.default_mem_opt(), capi.default_mem_opt()
    for k, v in optkw.items():
        setattr(oopt, k, v); setattr(gopt, k, v)
    pr = simulate.make_read_pairs_bulk(g, n_pairs, seed=seed)
    reads = np.asarray(pr).reshape(-1, np.asarray(pr).shape[-1]).copy()
    rng = np.random.default_rng(seed)
    for p in range(0, n_pairs, 4):                                   # pairs inside the repeat copies: XA, ties, rescue
        st = starts[int(rng.integers(0, len(starts)))]
        a = st + int(rng.integers(0, 100)); bpos = a + int(rng.integers(180, 330))
        L = reads.shape[1]
        if bpos + L <= len(g):
            reads[2 * p] = g[a:a + L]
            reads[2 * p + 1] = (3 - g[bpos:bpos + L][::-1]).astype(np.uint8)
    if pair_flag & 0x800:
        for i in range(2, len(reads), 14):                             # split reads for mem_reorder_primary5 (few: mem_pestat sees them)
            a_, c_ = int(rng.integers(0, len(g) - 150)), int(rng.integers(0, len(g) - 150))
            rd = np.concatenate([g[a_:a_ + 55], g[c_:c_ + reads.shape[1] - 55]]).astype(np.uint8)
            reads[i] = (3 - rd[::-1]).astype(np.uint8) if rng.random() < 0.5 else rd
    reads[8] = rng.integers(0, 4, size=reads.shape[1], dtype=np.uint8)          # one end unalignable
    reads[20] = rng.integers(0, 4, size=reads.shape[1], dtype=np.uint8); reads[21] = rng.integers(0, 4, size=reads.shape[1], dtype=np.uint8)
    ix = capi.Index.from_host(idx, 0)
    ix.set_contig_names([b"chr1"])
    enc, cum = simulate.flatten_reads(reads)
    b = capi.Batch(ix, len(reads), int(cum[-1]))
    b.seed_upload(enc, cum)
    b.seed_run(capi.default_seed_opt(), with_sa=True)
    b.chain_run(gopt); b.extend_run(gopt); b.dedup_run(gopt)
    pes = b.pestat(gopt)
    b.pair_run(pes, gopt, id_base=0, sopt=capi.default_sam_opt(pair_flag) if pair_flag else None)
    regs, off, pairs = b.pair_fetch()
    b.reg2aln(gopt, 1)
    quals = rng.integers(33, 74, size=len(enc), dtype=np.uint8)
    names = [b"pair%d" % (i // 2) for i in range(len(reads))]
    comments = [b"BC:Z:%d" % (i // 2) if (i // 2) % 4 == 0 else None for i in range(len(reads))]
    return dict(g=g, idx=idx, ix=ix, b=b, enc=enc, cum=cum, regs=regs, off=off, pairs=pairs, pes=pes, quals=quals, names=names,
                comments=comments, oopt=oopt, gopt=gopt)


def _compare_pe(c, flag=0, rg=b"", T=None, annos=None):
    import contextlib
    b = c["b"]
    if annos is not None:
        c["ix"].set_contig_annos(annos)
    so, sg = loader.default_sam_opt(flag, rg), capi.default_sam_opt(flag, rg)
    if T is not None:
        so.T = sg.T = T
    b.sam_upload(c["names"], c["quals"], c["comments"])
    nbytes = b.sam_run_pe(c["pes"], c["gopt"], sg)
    text, roff, _ = b.sam_fetch()
    with (loader.contig_annos(annos) if annos is not None else contextlib.nullcontext()):
        want = loader.sam_pe(c["regs"], c["off"], c["enc"], c["cum"], c["idx"].ref_0123, len(c["g"]), c["pes"], c["pairs"], c["names"],
                             quals=c["quals"], comments=c["comments"], contig_names=[b"chr1"], opt=c["oopt"], sopt=so)
    assert nbytes == sum(len(w) for w in want) == roff[-1]
    for r, w in enumerate(want):
        got = text[roff[r]:roff[r + 1]]
        assert got == w, (r, got, w)
    return text, want


def test_paired_end_sam_text_equals_oracle():
    c = _pe_pipeline(500, 17)
    text, want = _compare_pe(c)
    lines = [ln.split(b"\t") for ln in text.split(b"\n")[:-1]]
    flags = np.array([int(f[1]) for f in lines])
    assert (flags & 2).sum() > 600 and (flags & 0x8).sum() >= 1 and sum(any(t.startswith(b"MC:Z:") for t in f[11:]) for f in lines) > 600
    assert text.count(b"\tXA:Z:") > 20
    # both branches ran: paired records (0x2 with the pair's mapq) and mem_reg2sam records of the no_pairing branch
    pr = c["pairs"]
    assert (pr["score"] > 0).sum() > 300 and (pr["score"] == 0).sum() >= 1
    for flag, rg, T in ((0x8, b"rg1", None), (0x200 | 0x10, b"", None), (0, b"", 60)):
        _compare_pe(c, flag, rg, T)
    tv, _ = _compare_pe(c, 0x100, annos=[b"AC:1  LN:x\tM5:y"])     # `mem -V`: an unmapped end placed at its mate carries the tag, an unmapped pair not
    nl = tv.count(b"\n")
    assert nl - 10 < tv.count(b"\tXR:Z:AC:1  LN:x M5:y\n") < nl
    with pytest.raises(capi.BwamsError):
        c["b"].sam_run(c["gopt"], capi.default_sam_opt())              # the single-end form refuses a paired-end chunk
    bad_names = list(c["names"]); bad_names[7] = b"other"
    c["b"].sam_upload(bad_names, c["quals"])
    with pytest.raises(capi.BwamsError, match="different names"):
        c["b"].sam_run_pe(c["pes"], c["gopt"], capi.default_sam_opt())   # mem_sam_pe's fatal error, reported
    c["b"].close(); c["ix"].close()
    c = _pe_pipeline(300, 23, a=2, b=5, pen_unpaired=9, mapq_coef_len=0)
    _compare_pe(c)
    c["b"].close(); c["ix"].close()


def test_paired_end_sam_text_nopairing_and_primary5():
    """`mem -P`: no mem_pair, and mem_sam_pe's unpaired branch does not flag proper pairs (bwamem_pair.cpp:1066, :1176);
    `mem -5`: the reordered lists through the paired-end text."""
    c = _pe_pipeline(300, 29, pair_flag=0x4)
    assert (c["pairs"]["score"] == 0).all()
    text, _ = _compare_pe(c, 0x4)
    flags = np.array([int(ln.split(b"\t")[1]) for ln in text.split(b"\n")[:-1]])
    assert (flags & 2).sum() == 0 and (flags & 1).all()
    t2, _ = _compare_pe(c, 0)                                   # the same regions without the flag in the text: proper pairs are flagged
    assert (np.array([int(ln.split(b"\t")[1]) for ln in t2.split(b"\n")[:-1]]) & 2).sum() > 300
    c["b"].close(); c["ix"].close()
    c = _pe_pipeline(300, 31, pair_flag=0x800)
    text, _ = _compare_pe(c, 0x800 | 0x1000)
    assert text.count(b"\tSA:Z:") > 20
    c["b"].close(); c["ix"].close()


@pytest.mark.parametrize("L", [150, 101])
def test_exact_match_records_equal_oracle(L):
    """A single-end chunk behind the exact-match filter: resolved reads get mem_perfect2sam_cont's records (all flag settings,
    ALT-only reads, multi-location reads, reads longer than L on the reverse strand), the others mem_reg2sam's."""
    from bwams import emf, fmindex
    from util import toy
    g0, _ = toy(70000)
    g = g0[:60000].copy()
    g[5000:5400] = g[1000:1400]
    g[9000:9400] = (3 - g[1000:1400][::-1])
    g[30000:30400] = g[1000:1400]                            # a copy on the ALT sequence
    g[40000:40300] = g[26000:26300]                          # ALT-only copies
    g[20000:20700] = np.tile(np.array([0, 1, 0, 2, 1], np.uint8), 140)
    idx = fmindex.build_fmindex(g)
    ix = capi.Index.from_host(idx, 0)
    contigs = np.zeros(3, capi.CONTIG_DTYPE)
    contigs["offset"], contigs["len"], contigs["is_alt"] = [0, 8000, 25000], [8000, 17000, len(g) - 25000], [0, 0, 1]
    ix.set_contigs(contigs)
    cnames = [b"chrA", b"chrB", b"chrB_alt"]
    ix.set_contig_names(cnames)
    tab = emf.build_emf(g, L)
    e = capi.Emf(ix, table=tab)
    o = loader.OracleEMF(tab, idx.ref_0123)
    rng = np.random.default_rng(L + 5)
    reads = []
    for it in range(900):
        ln = L if it % 3 else int(rng.integers(L + 1, L + 40))
        where = it % 6
        st = (int(rng.integers(1000, 1400 - ln)) if where == 0 else int(rng.integers(20000, 20700 - ln)) if where == 1
              else int(rng.integers(26000, 26300 - ln)) if where == 2 else int(rng.integers(0, len(g) - ln)))
        rd = g[st:st + ln].copy()
        if it % 2:
            rd = simulate.revcomp(rd)
        if it % 7 == 0:
            rd[rng.integers(0, ln)] ^= 1                     # not exact: the normal path
        reads.append(rd)
    enc, cum = simulate.flatten_reads(reads)
    gopt, oopt = capi.default_mem_opt(), loader.default_mem_opt()
    b = capi.Batch(ix, len(reads), int(cum[-1]))
    b.seed_upload(enc, cum)
    b.emf_run(e)
    perfect, code = b.emf_fetch(len(reads))
    eregs, eoff, _ = b.emf_regs(e, gopt)
    b.seed_run(capi.default_seed_opt(), with_sa=True)        # the resolved reads are skipped
    b.chain_run(gopt); b.extend_run(gopt); b.dedup_run(gopt)
    b.mark_primary_se(gopt, id_base=0)
    regs, off, _ = b.pair_fetch()
    b.reg2aln(gopt, 1)
    names = [b"q%d" % i for i in range(len(reads))]
    quals = rng.integers(33, 74, size=len(enc), dtype=np.uint8)
    comments = [b"X:Z:%d" % i if i % 5 == 0 else None for i in range(len(reads))]
    b.sam_upload(names, quals, comments)
    n_res = 0
    annos = [b"", b"chrB's\tannotation", b"ALT"]
    ix.set_contig_annos(annos)
    for flag, rg in ((0, b""), (0x100, b"g"), (0x8, b"rg")):
        so, sg = loader.default_sam_opt(flag, rg), capi.default_sam_opt(flag, rg)
        b.sam_run_emf(e, gopt, sg)
        text, roff, _ = b.sam_fetch()
        assert (b"\tXR:Z:chrB's annotation\n" in text) == (flag == 0x100)
        with loader.contig_annos(annos):
            normal = loader.reg2sam_se(regs, off, enc, cum, idx.ref_0123, len(g), names, quals=quals, comments=comments, contigs=contigs,
                                       contig_names=cnames, opt=oopt, sopt=so)
        for r, rd in enumerate(reads):
            got = text[roff[r]:roff[r + 1]]
            if eoff[r + 1] > eoff[r]:
                want_regs, _ = o.perfect2reg(rd, int(perfect[r, 0]), int(perfect[r, 1]), len(g), contigs=contigs)
                with loader.contig_annos(annos):
                    want = loader.perfect2sam(want_regs, rd, len(g), L, names[r], qual=bytes(quals[cum[r]:cum[r + 1]]), comment=comments[r],
                                              contigs=contigs, contig_names=cnames, opt=oopt, sopt=so)
                n_res += 1
                f = got.split(b"\n")[0].split(b"\t")
                assert f[4] == b"60" and f[5] == b"%dM" % len(rd) and b"NM:i:0" in f
            else:
                want = normal[r]
            assert got == want, (r, got, want)
    assert n_res > 900
    joined = text
    assert b"chrB_alt\t" in joined and joined.count(b"\t256\t") + joined.count(b"\t272\t") > 50       # MEM_F_ALL: secondary exact-match records
    with pytest.raises(capi.BwamsError):
        b2 = capi.Batch(ix, 4, 600)
        b2.sam_run_emf(e, gopt, capi.default_sam_opt())
    b.close(); e.close(); ix.close()


def test_reg2aln_restricted_to_what_the_text_needs():
    """bwams_reg2aln_run_sam aligns only the regions the SAM text reads (as the reference calls mem_reg2aln): the text is the same, the
    records of the aligned regions are those of the full run, the others stay unmapped records."""
    # single-end
    c = _pipeline(900, 5)
    b = c["b"]
    full = c["aln"]
    for flag in (0, 0x8, 0x10):
        so, sg = loader.default_sam_opt(flag), capi.default_sam_opt(flag)
        b.reg2aln(c["gopt"], 1)
        b.sam_upload(c["names"], c["quals"], c["comments"])
        b.sam_run(c["gopt"], sg)
        want_text, _, _ = b.sam_fetch()
        aln, cig, md, n_need = b.reg2aln_sam(c["gopt"], sg)
        b.sam_run(c["gopt"], sg)
        text, _, _ = b.sam_fetch()
        assert text == want_text
        done = aln["rid"] >= 0
        assert n_need == int(done.sum()) + int(((full["rid"] < 0) & False).sum()) and 0 < n_need < len(full)
        for f in ("pos", "rid", "flag", "is_rev", "NM", "n_cigar", "md_len", "score", "sub", "mapq"):
            assert np.array_equal(aln[f][done], full[f][done]), f
        assert (aln["n_cigar"][~done] == 0).all()
        if flag == 0:
            assert n_need < len(full)                         # e.g. the secondaries below XA_drop_ratio are not aligned
    b.close(); c["ix"].close()
    # paired-end
    c = _pe_pipeline(500, 17)
    b = c["b"]
    for flag in (0, 0x8):
        so, sg = loader.default_sam_opt(flag), capi.default_sam_opt(flag)
        b.reg2aln(c["gopt"], 1)
        b.sam_upload(c["names"], c["quals"], c["comments"])
        b.sam_run_pe(c["pes"], c["gopt"], sg)
        want_text, _, _ = b.sam_fetch()
        aln, cig, md, n_need = b.reg2aln_sam(c["gopt"], sg, pes=c["pes"])
        b.sam_run_pe(c["pes"], c["gopt"], sg)
        text, _, _ = b.sam_fetch()
        assert text == want_text and 0 < n_need <= len(aln)
    with pytest.raises(capi.BwamsError):
        b.reg2aln_sam(c["gopt"], capi.default_sam_opt())          # a paired-end chunk needs its pes
    b.close(); c["ix"].close()
