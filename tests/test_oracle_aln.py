"""The SAM-side alignment step of the oracle (oracle/aln_oracle.c).  ksw_global2 with its traceback is PINNED to the
reference's ksw.cpp object; bwa_gen_cigar2 / mem_reg2aln (bwa.cpp / bwamem.cpp, not buildable here) are checked through
properties: the CIGAR consumes exactly the query and reference spans, NM and MD are recomputed independently."""
import numpy as np
import pytest

from bwams import simulate
from oracle import loader
from util import toy

REF = loader.ref_lib()
needs_ref = pytest.mark.skipif(REF is None, reason="oracle/_ref not built (reference tree absent)")


def _pairs(n, seed):
    rng = np.random.default_rng(seed)
    for _ in range(n):
        ql = int(rng.integers(1, 160))
        q = rng.integers(0, 4, size=ql, dtype=np.uint8)
        t = list(q)
        rate = float(rng.choice([0.0, 0.02, 0.1, 0.3]))
        o = []
        for b in t:
            u = rng.random()
            if u < rate * 0.5:
                o.append((b + rng.integers(1, 4)) & 3)
            elif u < rate * 0.75:
                continue
            elif u < rate:
                o.extend([b, rng.integers(0, 4)])
            else:
                o.append(b)
        if not o:
            o = [0]
        t = np.array(o, np.uint8)
        if rng.random() < 0.1:
            q[rng.integers(0, ql)] = 4
        w = int(rng.choice([0, 1, 3, 8, 20, 100, 400]))
        yield q, t, max(w, abs(len(t) - ql))            # the band must reach the last cell, as bwa_gen_cigar2's min_w ensures


@needs_ref
def test_global_alignment_with_traceback_equals_reference():
    for opt in (loader.default_sw_opt(), loader.default_sw_opt(5, 2, 3)):
        if opt.mat[0] == 2:
            opt.o_del, opt.e_del, opt.o_ins, opt.e_ins = 5, 2, 4, 1
        for q, t, w in _pairs(600, 3):
            a = loader.ksw_global2_cigar(q, t, w, opt)
            b = loader.ksw_global2_cigar(q, t, w, opt, L=REF)
            assert a[0] == b[0] and np.array_equal(a[1], b[1]), (len(q), len(t), w)
            assert a[0] == loader.ksw_global2_score(q, t, w, opt)


def _walk(cig, q, r):
    """(query consumed, reference consumed, mismatches + gap bases, MD) from a CIGAR without clips."""
    x = y = nm = u = 0
    md = ""
    for k, c in enumerate(cig):
        op, ln = int(c) & 0xf, int(c) >> 4
        if op == 0:
            for i in range(ln):
                if q[x + i] != r[y + i]:
                    md += str(u) + "ACGTN"[r[y + i]]; nm += 1; u = 0
                else:
                    u += 1
            x += ln; y += ln
        elif op == 2:
            if 0 < k < len(cig) - 1:
                md += str(u) + "^" + "".join("ACGTN"[b] for b in r[y:y + ln]); u = 0; nm += ln
            y += ln
        elif op == 1:
            x += ln; nm += ln
    return x, y, nm, md + str(u)


def test_reg2aln_properties_on_toy_regions():
    g, idx = toy()
    l_pac = len(g)
    ref = idx.ref_0123
    reads, _, _ = simulate.make_reads(g, 400, seed=5)
    enc, cum = simulate.flatten_reads(reads)
    o = loader.OracleFMI(idx)
    sm = o.collect_smem(enc, cum)
    coord, off = o.sa_lookup(sm)
    ch, sd, choff = loader.chain_seeds(sm, coord, off, cum, l_pac)
    regs, reg_off, _ = loader.chain2aln(ch, sd, choff, enc, cum, ref, l_pac)
    fin, fin_off = loader.regs_finish(regs, reg_off, enc, cum, ref, l_pac)
    assert len(fin) > 300
    aln, cig, md = loader.reg2aln(fin, fin_off, enc, cum, ref, l_pac)
    n_gapped = 0
    for r in range(len(fin_off) - 1):
        q = enc[cum[r]:cum[r + 1]]
        for k in range(fin_off[r], fin_off[r + 1]):
            a, ar = aln[k], fin[k]
            c = cig[a["cigar_off"]:a["cigar_off"] + a["n_cigar"]]
            m = bytes(md[a["md_off"]:a["md_off"] + a["md_len"]])
            assert m.endswith(b"\0") and a["rid"] == 0 and a["flag"] in (0, 0x100)
            is_rev = ar["rb"] >= l_pac
            assert a["is_rev"] == int(is_rev)
            core = [x for x in c if (int(x) & 0xf) != 3]
            clip5 = int(c[0]) >> 4 if (int(c[0]) & 0xf) == 3 else 0
            clip3 = int(c[-1]) >> 4 if len(c) > 1 and (int(c[-1]) & 0xf) == 3 else 0
            assert clip5 == (len(q) - ar["qe"] if is_rev else ar["qb"]) and clip3 == (ar["qb"] if is_rev else len(q) - ar["qe"])
            # the alignment in forward-strand terms: query segment (reverse-complemented on the reverse strand) vs forward text
            qs = q[ar["qb"]:ar["qe"]]
            rs = ref[ar["rb"]:ar["re"]]
            if is_rev:                                  # bwa_gen_cigar2 reverses both; the CIGAR then reads along the forward strand
                qs, rs = qs[::-1], rs[::-1]
                rs_f = np.where(rs < 4, 3 - rs, rs); qs_f = np.where(qs < 4, 3 - qs, qs)
            else:
                rs_f, qs_f = rs, qs
            # leading / trailing deletions were squeezed out of the CIGAR: put the reference bases they skip back
            lead = int(a["pos"]) - int(l_pac * 2 - 1 - (ar["re"] - 1) if is_rev else ar["rb"])
            x, y, nm, want_md = _walk(core, qs_f, rs_f[lead:])
            assert x == len(qs) and lead + y <= len(rs) and lead >= 0
            assert nm == a["NM"] or lead + y < len(rs) or lead > 0
            if lead == 0 and y == len(rs):
                assert want_md.encode() + b"\0" == m, (r, k)
            n_gapped += any((int(v) & 0xf) in (1, 2) for v in core)
            assert 0 <= a["mapq"] <= 60 and a["score"] == ar["score"]
    assert n_gapped > 10
