"""bwams_fastq_decode — FASTQ text to the hot path's arrays on the device — against the oracle's restatement of kseq_read +
trim_readno + kseq2bseq1 + the base encoding, and the whole device chain FASTQ text -> SAM text against the oracle chain."""
import numpy as np
import pytest

from bwams import capi, simulate
from oracle import loader
from test_oracle_fastq import make_fastq
from test_oracle_sam import repeat_genome

pytestmark = pytest.mark.gpu


def _same(got, want):
    assert got["n"] == want["n"]
    assert got["names"] == want["names"] and got["comments"] == want["comments"]
    assert np.array_equal(got["cum"], want["cum"]) and np.array_equal(got["enc"], want["enc"]) and np.array_equal(got["quals"], want["quals"])


@pytest.mark.parametrize("seed,crlf,n", [(1, False, 3000), (2, True, 1500), (3, False, 1), (4, False, 257)])
def test_decode_equals_oracle(seed, crlf, n):
    text, _ = make_fastq(n, seed, crlf=crlf)
    text = text.replace(b"-", b"N")
    for t in (text, text[:-1] if not crlf else text):            # with and without the final newline
        f = capi.Fastq(t)
        want = loader.fastq_parse(t)
        assert want["status"] == 0 and want["has_qual"].all()
        _same(f.fetch(), want)
        f.close()


def test_decode_corners_and_refusals():
    f = capi.Fastq(b"")
    assert f.n_reads == 0 and f.fetch()["n"] == 0
    f.close()
    f = capi.Fastq(b"@x/9 c  d\nacgtn\n+x\n!!!!!\n@/1\n\n+\n\n")
    g = f.fetch()
    assert g["names"] == [b"x", b"/1"] and g["comments"] == [b"c  d", None] and list(g["enc"]) == [0, 1, 2, 3, 4] and list(g["cum"]) == [0, 5, 5]
    _same(g, loader.fastq_parse(b"@x/9 c  d\nacgtn\n+x\n!!!!!\n@/1\n\n+\n\n"))
    f.close()
    for bad in (b"@a\nACGT\nAC\n+\nIIIII\n",                       # multi-line record, quality one short
                b"@a\nACGT\n@b\nACGT\n+\nIIII\n",                   # a record without a '+' line among FASTQ records
                b"@a\nACGT\n+\nIIII\nstray\n@b\nAC\n+\nII\n",       # text between records (the serial reader scans it for '@')
                b">fa\nACGT\n@fq\nACGT\n+\nIIII\n",                 # FASTA and FASTQ records mixed
                b">fa\nACGT\n+\nIIII\n",                            # '>' header with a quality part
                b">fa\nAC-T\n",
                b"@a\nACGT\n+\nIII\n",                             # quality shorter than the sequence
                b"@a\nACGT\n+\nIIIII\n",                           # ... longer
                b"junk\n@a\nACGT\n+\nIIII\n",                      # text before the first header
                b"@a\nAC-T\n+\nIIII\n"):                           # '-' (nst_nt4_table: 5)
        with pytest.raises(capi.BwamsError):
            capi.Fastq(bad)


def _wrapped_fastq(n, seed, crlf=False):
    """FASTQ records as kseq_read admits them beyond the four-line shape: sequence and quality over several lines (of different widths),
    blank lines inside the sequence and between records, quality lines that BEGIN with '@' or '+' (so that a line index alone cannot
    tell records apart), comments, "/1" names, an empty record, '+name' repeated on the plus line."""
    rng = np.random.default_rng(seed)
    nl = b"\r\n" if crlf else b"\n"
    parts = []
    for i in range(n):
        ln = int(rng.integers(1, 320)) if (i % 19 != 7 or crlf) else 0      # (an empty record in CRLF text is the reader's -2: its quality is '\\r')
        seq = bytes(rng.choice(np.frombuffer(b"ACGTacgtNn", np.uint8), size=ln))
        qual = bytearray(rng.integers(33, 74, size=ln, dtype=np.uint8).tobytes())
        ws = int(rng.integers(1, 100)) if i % 3 else 1000
        wq = int(rng.integers(1, 100)) if i % 4 else 1000
        for k in range(0, ln, wq):                                   # quality lines starting with '@' / '+' (and a header-looking one)
            if rng.random() < 0.3:
                qual[k] = ord("@") if rng.random() < 0.6 else ord("+")
        hdr = b"@wr%d" % i + (b"/1" if i % 3 == 0 else b"") + (b" c=%d  y" % ln if i % 4 == 0 else b"")
        sl = [seq[k:k + ws] for k in range(0, ln, ws)]
        if i % 7 == 0 and sl:
            sl.insert(int(rng.integers(1 if crlf else 0, len(sl) + 1)), b"")   # a blank line before / inside / after the sequence lines
            #                                                          (a "\\r\\n" line in FRONT of the sequence is a base to the reader: its '\\r' stays)
        ql = [bytes(qual[k:k + wq]) for k in range(0, ln, wq)] or [b""]
        parts.append(hdr + nl + b"".join(x + nl for x in sl) + (b"+wr%d" % i if i % 5 == 0 else b"+") + nl + b"".join(x + nl for x in ql) +
                     (nl if i % 11 == 0 else b""))                   # a blank line between records
    return b"".join(parts)


@pytest.mark.parametrize("seed,crlf,n", [(1, False, 3000), (2, True, 800), (3, False, 1), (4, False, 64)])
def test_wrapped_fastq_records_equal_oracle(seed, crlf, n):
    """Records over any number of lines are DECODED on the device (round 3 refused them): every '@' line parsed as if a record began
    there, the real records found as the chain of successors from the first line (fastq.hip)."""
    text = _wrapped_fastq(n, seed, crlf)
    for t in (text, text.rstrip(b"\r\n")):
        want = loader.fastq_parse(t)
        assert want["status"] == 0 and want["n"] == n
        f = capi.Fastq(t)
        got = f.fetch()
        assert got["n"] == n and got["names"] == want["names"] and got["comments"] == want["comments"]
        assert np.array_equal(got["cum"], want["cum"]) and np.array_equal(got["enc"], want["enc"])
        hq = np.repeat(want["has_qual"].astype(bool), np.diff(want["cum"]))
        assert np.array_equal(got["quals"][hq], want["quals"][hq])
        f.close()
    # four lines per record but one record wrapped so that the line count stays a multiple of four
    t = b"@a\nAC\nGT\n+\nIIII\n@b\nA\n+\nI\n"
    f = capi.Fastq(t + b"@c\nACG\n+\n@II\n" * 0)
    want = loader.fastq_parse(t)
    got = f.fetch()
    assert got["names"] == want["names"] == [b"a", b"b"] and np.array_equal(got["enc"], want["enc"]) and np.array_equal(got["quals"], want["quals"])
    f.close()
    # a blank line behind the last record, blank lines in front of the first
    for t in (b"@a\nACGT\n+\nIIII\n\n", b"\n\n@a\nACGT\n+\nIIII\n"):
        f = capi.Fastq(t)
        want = loader.fastq_parse(t)
        got = f.fetch()
        assert got["n"] == want["n"] == 1 and np.array_equal(got["enc"], want["enc"]) and np.array_equal(got["quals"], want["quals"])
        f.close()


def _fasta_text(n, seed, crlf=False, width=60):
    """FASTA records as the serial reader admits them: sequences over several lines of `width`, blank lines, comments, "/1" names, lower
    case and IUPAC codes, an empty record, lines holding a lone '\\r'."""
    rng = np.random.default_rng(seed)
    nl = b"\r\n" if crlf else b"\n"
    parts = []
    for i in range(n):
        ln = int(rng.integers(1, 400)) if i % 17 else 0
        seq = bytes(rng.choice(np.frombuffer(b"ACGTacgtNnRYKM", np.uint8), size=ln))
        hdr = b">fa%d" % i + (b"/2" if i % 3 == 0 else b"") + (b"\tlen=%d  x" % ln if i % 4 == 0 else b" " if i % 4 == 1 else b"")
        w = width if i % 5 else int(rng.integers(1, 90))
        lines = [seq[k:k + w] for k in range(0, ln, w)]
        if i % 7 == 0 and lines:
            lines.insert(int(rng.integers(0, len(lines) + 1)), b"")              # a blank line inside / before / after the sequence
        parts.append(hdr + nl + b"".join(x + nl for x in lines))
    return b"".join(parts)


@pytest.mark.parametrize("seed,crlf,n", [(1, False, 2500), (2, True, 900), (3, False, 1)])
def test_fasta_decode_equals_oracle(seed, crlf, n):
    text = _fasta_text(n, seed, crlf)
    for t in (text, text.rstrip(b"\r\n")):
        f = capi.Fastq(t)
        want = loader.fastq_parse(t)
        assert want["status"] == 0 and not want["has_qual"].any() and want["n"] == n
        got = f.fetch()
        assert got["quals"] is None
        assert got["n"] == want["n"] and got["names"] == want["names"] and got["comments"] == want["comments"]
        assert np.array_equal(got["cum"], want["cum"]) and np.array_equal(got["enc"], want["enc"])
        f.close()
    # the '\r' rule looks at the whole sequence so far: a first line of a lone '\r' stays (one N), later ones go
    for t in (b">a\n\r\nAC\r\n\r\nGT\n>b c\n\n\n>c\nA\r", b">only_header", b">x\n\r\n\r\n"):
        f = capi.Fastq(t)
        want = loader.fastq_parse(t)
        got = f.fetch()
        assert got["names"] == want["names"] and got["comments"] == want["comments"]
        assert np.array_equal(got["cum"], want["cum"]) and np.array_equal(got["enc"], want["enc"]), t
        f.close()


def test_fastq_text_to_sam_text_on_the_device():
    """FASTQ bytes in, SAM bytes out, nothing but the two texts crossing the boundary; the oracle chain on the same bytes."""
    g, idx, starts = repeat_genome()
    reads, _, _ = simulate.make_reads(g, 600, seed=31)
    rng = np.random.default_rng(8)
    for i in range(0, 600, 3):
        st = starts[int(rng.integers(0, len(starts)))] + int(rng.integers(0, 500 - reads.shape[1]))
        reads[i] = g[st:st + reads.shape[1]]
    parts = []
    for i, r in enumerate(reads):
        seq = bytes(b"ACGTN"[b] for b in r)
        qual = bytes(rng.integers(33, 74, size=len(r), dtype=np.uint8))
        parts.append(b"@frag%d/1%s\n%s\n+\n%s\n" % (i, b" RG:Z:x%d" % i if i % 5 == 0 else b"", seq, qual))
    parts.insert(7, b"@empty_read with a comment\n\n+\n\n")            # no bases: an unaligned record with empty SEQ and '*' for QUAL
    text = b"".join(parts)
    ix = capi.Index.from_host(idx, 0)
    ix.set_contig_names([b"chrR"])
    f = capi.Fastq(text)
    b = capi.Batch(ix, f.n_reads, f.n_bases)
    f.to_batch(b)
    gopt, oopt = capi.default_mem_opt(), loader.default_mem_opt()
    b.seed_run(capi.default_seed_opt(), with_sa=True)
    b.chain_run(gopt); b.extend_run(gopt); b.dedup_run(gopt)
    b.mark_primary_se(gopt, id_base=0)
    b.reg2aln(gopt, 1)
    b.sam_run(gopt, capi.default_sam_opt())
    sam, roff, _ = b.sam_fetch()
    # the oracle chain from the same FASTQ bytes
    w = loader.fastq_parse(text)
    o = loader.OracleFMI(idx)
    sm = o.collect_smem(w["enc"], w["cum"])
    coord, off = o.sa_lookup(sm)
    l_pac = len(g)
    ch, sd, choff = loader.chain_seeds(sm, coord, off, w["cum"], l_pac)
    regs, reg_off, _ = loader.chain2aln(ch, sd, choff, w["enc"], w["cum"], idx.ref_0123, l_pac)
    fin, fin_off = loader.regs_finish(regs, reg_off, w["enc"], w["cum"], idx.ref_0123, l_pac)
    for r in range(len(fin_off) - 1):
        a, e = int(fin_off[r]), int(fin_off[r + 1])
        if e > a:
            fin[a:e] = loader.mark_primary_se(fin[a:e], r)[0]
    want = loader.reg2sam_se(fin, fin_off, w["enc"], w["cum"], idx.ref_0123, l_pac, w["names"], quals=w["quals"], comments=w["comments"],
                             contig_names=[b"chrR"], opt=oopt)
    assert sam == b"".join(want)
    assert sam.count(b"\n") >= 600 and b"frag0\t" in sam and b"/1" not in sam.split(b"\n")[0].split(b"\t")[0]
    assert b"empty_read\t4\t*\t0\t0\t*\t*\t0\t0\t\t*\tAS:i:0\tXS:i:0\twith a comment\n" in sam
    f.close(); b.close(); ix.close()


def _fastq_of(reads, names, quals, comments=None):
    parts = []
    o = 0
    for i, r in enumerate(reads):
        seq = bytes(b"ACGTN"[b] for b in r)
        hdr = names[i] + (b" " + comments[i] if comments is not None and comments[i] else b"")
        parts.append(b"@" + hdr + b"\n" + seq + b"\n+\n" + bytes(quals[o:o + len(r)]) + b"\n")
        o += len(r)
    return b"".join(parts)


def test_process_chunk_is_the_stage_sequence():
    """bwams_process_chunk (the outer boundary, text to text) against the oracle chain: single-end, single-end over the ERT,
    single-end behind the exact-match filter, paired-end with inferred and with given insert-size statistics."""
    from bwams import emf as emf_mod
    from util import oracle_pe_pipeline
    g, idx, starts = repeat_genome()
    l_pac = len(g)
    ix = capi.Index.from_host(idx, 0)
    ix.set_contig_names([b"chrR"])
    rng = np.random.default_rng(12)
    gopt, oopt = capi.default_mem_opt(), loader.default_mem_opt()
    # ---- single-end (+ ERT, + EMF)
    reads, _, _ = simulate.make_reads(g, 400, seed=41)
    for i in range(0, 400, 4):
        st = starts[int(rng.integers(0, len(starts)))] + int(rng.integers(0, 500 - reads.shape[1]))
        reads[i] = g[st:st + reads.shape[1]]
    for i in range(1, 400, 8):                               # exact copies: the EMF resolves them
        st = int(rng.integers(0, l_pac - reads.shape[1]))
        reads[i] = g[st:st + reads.shape[1]] if i % 16 == 1 else simulate.revcomp(g[st:st + reads.shape[1]])
    for i in range(2, 400, 16):                              # split reads, the longer part on the right (`mem -5` below)
        a_, c_ = int(rng.integers(0, l_pac - 150)), int(rng.integers(0, l_pac - 150))
        reads[i] = np.concatenate([g[a_:a_ + 55], g[c_:c_ + reads.shape[1] - 55]])
    enc, cum = simulate.flatten_reads(reads)
    quals = rng.integers(33, 74, size=len(enc), dtype=np.uint8)
    names = [b"rd%d" % i for i in range(len(reads))]
    comments = [b"x:i:%d" % i if i % 3 == 0 else None for i in range(len(reads))]
    text = _fastq_of(reads, names, quals, comments)
    b = capi.Batch(ix, len(reads), int(cum[-1]))
    sam, off = b.process_chunk(text, n_processed=1000, copy_comment=True)
    sam5, _ = b.process_chunk(text, n_processed=1000, sopt=capi.default_sam_opt(0x800 | 0x1000))       # `mem -5`
    sam_nc, _ = b.process_chunk(text, n_processed=1000)                                  # without `-C` the comments are dropped
    o = loader.OracleFMI(idx)
    sm = o.collect_smem(enc, cum)
    coord, soff = o.sa_lookup(sm)
    ch, sd, choff = loader.chain_seeds(sm, coord, soff, cum, l_pac)
    regs, reg_off, _ = loader.chain2aln(ch, sd, choff, enc, cum, idx.ref_0123, l_pac)
    fin, fin_off = loader.regs_finish(regs, reg_off, enc, cum, idx.ref_0123, l_pac)
    fin5 = fin.copy()
    for r in range(len(reads)):
        a, e = int(fin_off[r]), int(fin_off[r + 1])
        if e > a:
            fin5[a:e] = loader.mark_primary_se(fin[a:e], 1000 + r, primary5_T=30)[0]
            fin[a:e] = loader.mark_primary_se(fin[a:e], 1000 + r)[0]
    want = loader.reg2sam_se(fin, fin_off, enc, cum, idx.ref_0123, l_pac, names, quals=quals, comments=comments, contig_names=[b"chrR"])
    assert sam == b"".join(want) and off[-1] == len(sam)
    want5 = loader.reg2sam_se(fin5, fin_off, enc, cum, idx.ref_0123, l_pac, names, quals=quals, contig_names=[b"chrR"],
                              sopt=loader.default_sam_opt(0x800 | 0x1000))
    assert sam5 == b"".join(want5) and (fin5["rb"] != fin["rb"]).sum() >= 10
    want_nc = loader.reg2sam_se(fin, fin_off, enc, cum, idx.ref_0123, l_pac, names, quals=quals, contig_names=[b"chrR"])
    assert sam_nc == b"".join(want_nc) and sam_nc != sam
    ert = capi.Ert.build(ix, kmer=8, xmer=2, read_len=151, hit_threshold=16)
    sam_e, _ = b.process_chunk(text, ert=ert, n_processed=1000, copy_comment=True)
    assert sam_e == sam                                      # ERT seeding: the same seeds, the same text (single-end)
    ert.close()
    tab = emf_mod.build_emf(g, reads.shape[1])
    e = capi.Emf(ix, table=tab)
    sam_f, off_f = b.process_chunk(text, emf=e, n_processed=1000, copy_comment=True)
    n_exact = 0
    for r in range(len(reads)):
        got = sam_f[off_f[r]:off_f[r + 1]]
        if got != want[r]:                                   # a resolved read: the exact-match record instead
            f = got.split(b"\n")[0].split(b"\t")
            assert f[4] == b"60" and f[5] == b"%dM" % reads.shape[1] and b"NM:i:0" in f
            n_exact += 1
    assert n_exact >= 40
    e.close(); b.close()
    # ---- paired-end
    pr = simulate.make_read_pairs_bulk(g, 300, seed=9)
    preads = np.asarray(pr).reshape(-1, np.asarray(pr).shape[-1])
    penc, pcum = simulate.flatten_reads(preads)
    pquals = rng.integers(33, 74, size=len(penc), dtype=np.uint8)
    pnames = [b"pp%d" % (i // 2) for i in range(len(preads))]
    ptext = _fastq_of(preads, [n + (b"/1" if i % 2 == 0 else b"/2") for i, n in enumerate(pnames)], pquals)
    b = capi.Batch(ix, len(preads), int(pcum[-1]))
    psam, poff = b.process_chunk(ptext, paired=True, n_processed=0)
    c = oracle_pe_pipeline(g, idx, preads)
    wregs, woff, wpairs = loader.pair_pe(c["regs"], c["reg_off"], penc, pcum, c["ref"], l_pac, c["pes"])
    pwant = loader.sam_pe(wregs, woff, penc, pcum, c["ref"], l_pac, c["pes"], wpairs, pnames, quals=pquals, contig_names=[b"chrR"])
    assert psam == b"".join(pwant)
    psam_p, _ = b.process_chunk(ptext, paired=True, sopt=capi.default_sam_opt(0x4 | 0x800))            # `mem -P -5` (without -5's MAPQ flag)
    wregs_p, woff_p, wpairs_p = loader.pair_pe(c["regs"], c["reg_off"], penc, pcum, c["ref"], l_pac, c["pes"], no_pairing=True, primary5_T=30)
    assert psam_p == b"".join(loader.sam_pe(wregs_p, woff_p, penc, pcum, c["ref"], l_pac, c["pes"], wpairs_p, pnames, quals=pquals,
                                             contig_names=[b"chrR"], sopt=loader.default_sam_opt(0x4 | 0x800))) and psam_p != psam
    psam_s, _ = b.process_chunk(ptext, paired=True, sopt=capi.default_sam_opt(0x20))                   # `mem -S` through the options' flag
    wregs_s, woff_s, wpairs_s = loader.pair_pe(c["regs"], c["reg_off"], penc, pcum, c["ref"], l_pac, c["pes"], no_rescue=True)
    assert psam_s == b"".join(loader.sam_pe(wregs_s, woff_s, penc, pcum, c["ref"], l_pac, c["pes"], wpairs_s, pnames, quals=pquals,
                                             contig_names=[b"chrR"], sopt=loader.default_sam_opt(0x20)))
    # the same pairs as two files (bseq_read_orig with ks2): record k of each text
    t1 = _fastq_of(preads[0::2], [n + b"/1" for n in pnames[0::2]], np.concatenate([pquals[pcum[i]:pcum[i + 1]] for i in range(0, len(preads), 2)]))
    t2 = _fastq_of(preads[1::2], [n + b"/2" for n in pnames[1::2]], np.concatenate([pquals[pcum[i]:pcum[i + 1]] for i in range(1, len(preads), 2)]))
    psam2f, poff2f = b.process_chunk2(t1, t2)
    assert psam2f == psam and np.array_equal(poff2f, poff)
    with pytest.raises(capi.BwamsError, match="same number"):
        b.process_chunk2(t1, t2[: t2.index(b"@pp7/2")])
    pes2 = c["pes"].copy(); pes2["low"][1] += 3
    psam2, _ = b.process_chunk(ptext, paired=True, pes=pes2)
    wregs2, woff2, wpairs2 = loader.pair_pe(c["regs"], c["reg_off"], penc, pcum, c["ref"], l_pac, pes2)
    assert psam2 == b"".join(loader.sam_pe(wregs2, woff2, penc, pcum, c["ref"], l_pac, pes2, wpairs2, pnames, quals=pquals, contig_names=[b"chrR"]))
    ert = capi.Ert.build(ix, kmer=8, xmer=2, read_len=151, hit_threshold=16)
    psam_e, _ = b.process_chunk(ptext, paired=True, ert=ert)              # ERT mode: the same seeds, mate rescue in its useErt form
    wregs_e, woff_e, wpairs_e = loader.pair_pe(c["regs"], c["reg_off"], penc, pcum, c["ref"], l_pac, c["pes"], use_ert=True)
    assert psam_e == b"".join(loader.sam_pe(wregs_e, woff_e, penc, pcum, c["ref"], l_pac, c["pes"], wpairs_e, pnames, quals=pquals,
                                             contig_names=[b"chrR"]))
    ert.close()
    with pytest.raises(capi.BwamsError):
        b.process_chunk(ptext[: ptext.index(b"@pp1/2")], paired=True)      # an odd number of reads
    assert b.process_chunk(b"", fetch=False) == 0                          # an empty chunk
    with pytest.raises(capi.BwamsError):
        b.process_chunk(b"@fq\nACGT\nAC\n+\nIIIII\n")                      # a quality string one short (kseq_read's -2): refused
    # FASTA reads (sequence lines of 70 bases): the same chain without qualities
    fa = b"".join(b">%s some text\n" % names[i] + b"".join(bytes(b"ACGTN"[x] for x in reads[i][k:k + 70]) + b"\n" for k in range(0, len(reads[i]), 70))
                  for i in range(len(reads)))
    b.close()
    b = capi.Batch(ix, len(reads), int(cum[-1]))
    sam_fa, _ = b.process_chunk(fa, n_processed=1000, copy_comment=True)
    want_fa = loader.reg2sam_se(fin, fin_off, enc, cum, idx.ref_0123, l_pac, names, comments=[b"some text"] * len(reads), contig_names=[b"chrR"])
    assert sam_fa == b"".join(want_fa) and sam_fa.count(b"\t*\tNM:i:") > 300
    b.close(); ix.close()


def test_process_chunk_paired_end_behind_the_exact_match_filter():
    """worker_sam's paired-end branch with PERFECT_MATCH: mem_pestat on the regions of worker_aln, then mem_perfect2reg for the resolved
    ends, then mem_sam_pe (bwamem.cpp:1689-1716) — bwams_process_chunk(paired, emf) against the same order of the restated pieces."""
    from bwams import emf as emf_mod
    g, idx, starts = repeat_genome()
    l_pac = len(g)
    ref = idx.ref_0123
    ix = capi.Index.from_host(idx, 0)
    ix.set_contig_names([b"chrR"])
    rng = np.random.default_rng(3)
    pr = simulate.make_read_pairs_bulk(g, 300, seed=19)
    reads = np.asarray(pr).reshape(-1, np.asarray(pr).shape[-1]).copy()
    L = reads.shape[1]
    for p in range(0, 300, 3):                               # pairs whose ends are exact copies of the genome (FR, insert ~ 330)
        a = int(rng.integers(0, l_pac - 400 - L))
        reads[2 * p] = g[a:a + L]
        if p % 2 == 0:
            reads[2 * p + 1] = simulate.revcomp(g[a + 330 - L:a + 330])
    enc, cum = simulate.flatten_reads(reads)
    quals = rng.integers(33, 74, size=len(enc), dtype=np.uint8)
    names = [b"pq%d" % (i // 2) for i in range(len(reads))]
    text = _fastq_of(reads, [n + (b"/1" if i % 2 == 0 else b"/2") for i, n in enumerate(names)], quals)
    tab = emf_mod.build_emf(g, L)
    e = capi.Emf(ix, table=tab)
    b = capi.Batch(ix, len(reads), int(cum[-1]))
    sam, off = b.process_chunk(text, paired=True, emf=e)
    # the restated pieces in worker_sam's order
    oe = loader.OracleEMF(tab, ref)
    probe = oe.probe_many(list(reads))
    hit = (probe[:, 0] == 3) | (probe[:, 0] == 4)
    assert hit.sum() > 120
    o = loader.OracleFMI(idx)
    sm = o.collect_smem(enc, cum, skip=hit.astype(np.uint8))
    coord, soff = o.sa_lookup(sm)
    ch, sd, choff = loader.chain_seeds(sm, coord, soff, cum, l_pac, ref_string=ref, enc=enc)
    regs, reg_off, _ = loader.chain2aln(ch, sd, choff, enc, cum, ref, l_pac)
    fin, fin_off = loader.regs_finish(regs, reg_off, enc, cum, ref, l_pac)
    pes = loader.pestat(fin, fin_off, l_pac)
    merged, moff = [], [0]
    for r in range(len(reads)):
        part = [fin[fin_off[r]:fin_off[r + 1]]]
        if hit[r]:
            part.append(oe.perfect2reg(reads[r], int(probe[r, 1]), int(probe[r, 2]), l_pac)[0])
        merged.extend(part)
        moff.append(moff[-1] + sum(len(x) for x in part))
    merged = np.concatenate(merged)
    wregs, woff, wpairs = loader.pair_pe(merged, np.asarray(moff, np.int64), enc, cum, ref, l_pac, pes)
    want = loader.sam_pe(wregs, woff, enc, cum, ref, l_pac, pes, wpairs, names, quals=quals, contig_names=[b"chrR"])
    for r, w in enumerate(want):
        assert sam[off[r]:off[r + 1]] == w, (r, sam[off[r]:off[r + 1]], w)
    e.close(); b.close(); ix.close()


def _bseq_classify(names):
    """bseq_classify (src/bwa.cpp:346-362) on the list of names -> (indices of the single reads, indices of the paired ones)."""
    a = ([], [])
    has_last = 1
    n = len(names)
    i = 1
    while i < n:
        if has_last:
            if names[i] == names[i - 1]:
                a[1].extend((i - 1, i)); has_last = 0
            else:
                a[0].append(i - 1)
        else:
            has_last = 1
        i += 1
    if has_last and n:
        a[0].append(n - 1)
    return a


def test_smart_pairing_chunk():
    """`mem -p` (process()'s MEM_F_SMARTPE branch, fastmap.cpp:378-414): single reads and interleaved pairs in one chunk; the restated
    pieces run on the two sets bseq_classify makes, with the ids the reference gives them, and every text returns to its place."""
    from util import oracle_pe_pipeline
    g, idx, starts = repeat_genome()
    l_pac = len(g)
    ix = capi.Index.from_host(idx, 0)
    ix.set_contig_names([b"chrR"])
    rng = np.random.default_rng(77)
    pr = np.asarray(simulate.make_read_pairs_bulk(g, 260, seed=5))
    pr = pr.reshape(-1, pr.shape[-1])
    singles, _, _ = simulate.make_reads(g, 150, seed=6)
    reads, names = [], []
    p = s_ = 0
    while p < 260 or s_ < 150:                                  # a random mixture; now and then three reads of one name (a pair and a single)
        if p < 260 and (s_ >= 150 or rng.random() < 0.6):
            nm = b"frag%d" % p
            reads += [pr[2 * p], pr[2 * p + 1]]; names += [nm + b"/1", nm + b"/2"]
            if p % 37 == 0 and s_ < 150:
                reads.append(singles[s_]); names.append(nm + b"/3"); s_ += 1
            p += 1
        else:
            reads.append(singles[s_]); names.append(b"solo%d" % s_); s_ += 1
    enc, cum = simulate.flatten_reads(reads)
    quals = rng.integers(33, 74, size=len(enc), dtype=np.uint8)
    text = _fastq_of(reads, names, quals)
    trimmed = [nm[:-2] if nm[-2:-1] == b"/" else nm for nm in names]
    i0, i1 = _bseq_classify(trimmed)
    assert len(i0) == 150 and len(i1) == 520
    NP = 4000
    b = capi.Batch(ix, len(reads), int(cum[-1]))
    sam, off, n_single = b.process_chunk_smart(text, n_processed=NP)
    assert n_single == len(i0) and len(off) == len(reads) + 1 and off[-1] == len(sam)

    def sub(ids):
        rs = [reads[i] for i in ids]
        e_, c_ = simulate.flatten_reads(rs)
        q_ = np.concatenate([quals[cum[i]:cum[i + 1]] for i in ids])
        return rs, e_, c_, q_, [trimmed[i] for i in ids]
    # the single reads: mem_process_seqs(n_processed, single-end)
    rs, e0, c0, q0, nm0 = sub(i0)
    o = loader.OracleFMI(idx)
    sm = o.collect_smem(e0, c0)
    coord, soff = o.sa_lookup(sm)
    ch, sd, choff = loader.chain_seeds(sm, coord, soff, c0, l_pac)
    regs, reg_off, _ = loader.chain2aln(ch, sd, choff, e0, c0, idx.ref_0123, l_pac)
    fin, fin_off = loader.regs_finish(regs, reg_off, e0, c0, idx.ref_0123, l_pac)
    for r in range(len(rs)):
        a, e = int(fin_off[r]), int(fin_off[r + 1])
        if e > a:
            fin[a:e] = loader.mark_primary_se(fin[a:e], NP + r)[0]
    want0 = loader.reg2sam_se(fin, fin_off, e0, c0, idx.ref_0123, l_pac, nm0, quals=q0, contig_names=[b"chrR"])
    # the pairs: mem_process_seqs(n_processed + n_single, paired-end, pes0 = NULL)
    rs, e1, c1, q1, nm1 = sub(i1)
    c = oracle_pe_pipeline(g, idx, np.stack(rs))
    wregs, woff, wpairs = loader.pair_pe(c["regs"], c["reg_off"], e1, c1, c["ref"], l_pac, c["pes"], id_base=(NP + len(i0)) >> 1)
    want1 = loader.sam_pe(wregs, woff, e1, c1, c["ref"], l_pac, c["pes"], wpairs, nm1, quals=q1, contig_names=[b"chrR"])
    want = [None] * len(reads)
    for j, i in enumerate(i0):
        want[i] = want0[j]
    for j, i in enumerate(i1):
        want[i] = want1[j]
    for i in range(len(reads)):
        assert sam[off[i]:off[i + 1]] == want[i], (i, names[i])
    # only single reads / only pairs / nothing
    only0 = _fastq_of([reads[i] for i in i0], [names[i] for i in i0], np.concatenate([quals[cum[i]:cum[i + 1]] for i in i0]))
    s0, _, n0 = b.process_chunk_smart(only0, n_processed=NP)
    assert n0 == len(i0) and s0 == b"".join(want0)
    s_e, o_e, n_e = b.process_chunk_smart(b"")
    assert s_e == b"" and n_e == 0 and len(o_e) == 1
    b.close(); ix.close()
