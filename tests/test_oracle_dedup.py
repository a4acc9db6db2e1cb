"""The tail of mem_kernel2_core (purged regions dropped, mem_sort_dedup_patch + mem_patch_reg, ALT mark): the global
alignment and the two sorts it relies on are pinned to the reference's own ksw.cpp object / ksort.h; the driver logic
is checked on constructed cases and through invariants."""
import numpy as np
import pytest

from bwams import simulate
from oracle import loader
from tests import util

REF = loader.ref_lib()
CREF = loader.ref_chain_lib()
needs_ref = pytest.mark.skipif(REF is None or CREF is None, reason="oracle/_ref not built (reference tree absent)")


@needs_ref
def test_ksw_global2_score_equals_reference_object():
    rng = np.random.default_rng(8)
    for trial in range(600):
        ql = int(rng.integers(1, 260))
        q = rng.integers(0, 4, size=ql, dtype=np.uint8)
        t = list(q)
        for _ in range(int(rng.integers(0, 6))):                   # indels and substitutions
            c = int(rng.integers(0, max(1, len(t))))
            k = int(rng.integers(0, 3))
            if k == 0 and t:
                t[c] = (t[c] + 1) & 3
            elif k == 1:
                t[c:c] = list(rng.integers(0, 4, size=int(rng.integers(1, 30))))
            elif t:
                del t[c:c + int(rng.integers(1, 30))]
        t = np.array(t if t else [0], dtype=np.uint8)
        if rng.random() < 0.1:
            t[rng.integers(0, len(t))] = 4
        w = int(rng.choice([1, 3, 10, 50, 100, 400]))
        opt = loader.default_sw_opt()
        if trial % 3 == 1:
            opt.o_del, opt.e_del, opt.o_ins, opt.e_ins = 4, 2, 7, 1
        assert loader.ksw_global2_score(q, t, w, opt) == loader.ksw_global2_score(q, t, w, opt, REF), (trial, ql, len(t), w)


@needs_ref
def test_region_sorts_equal_reference_ksort():
    rng = np.random.default_rng(9)
    for trial in range(300):
        n = int(rng.integers(0, 50 if trial < 150 else 2500))
        hi = int(rng.choice([3, 20, 1 << 30]))
        re = rng.integers(0, hi, size=n)
        assert np.array_equal(loader.ars_sort(0, re), loader.ars_sort(0, re, L=CREF))
        sc, rb, qb = rng.integers(0, min(hi, 200), size=n), rng.integers(0, hi, size=n), rng.integers(0, 4, size=n)
        assert np.array_equal(loader.ars_sort(1, sc, rb, qb), loader.ars_sort(1, sc, rb, qb, L=CREF))


def _reg(rb, re, qb, qe, score, rid=0, w=100, seedcov=0):
    r = np.zeros(1, loader.ALNREG_DTYPE)
    r["rb"], r["re"], r["qb"], r["qe"], r["score"], r["truesc"], r["rid"], r["w"], r["seedcov"] = rb, re, qb, qe, score, score, rid, w, seedcov
    return r


def test_constructed_cases():
    g, idx = util.toy(30000)
    l_pac = len(g)
    ref = np.concatenate([g, (3 - g[::-1]).astype(np.uint8)])
    # read = genome[1000:1070] + genome[1073:1153]: a 3-base deletion; two regions, colinear, close -> merged
    # (global score 150 - 9 = 141 >= 0.9 x the 153 predicted from the reference span; a 12-base gap: relative
    # bandwidth 12/162 > 5 % -> refused before the alignment)
    read = np.concatenate([g[1000:1070], g[1073:1153]])
    enc, cum = simulate.flatten_reads([read])
    regs = np.concatenate([_reg(1073, 1153, 70, 150, 80, seedcov=40), _reg(1000, 1070, 0, 70, 70, seedcov=70)])
    out, off = loader.regs_finish(regs, np.array([0, 2]), enc, cum, ref, l_pac)
    assert len(out) == 1 and off[-1] == 1
    o = out[0]
    assert (o["rb"], o["re"], o["qb"], o["qe"]) == (1000, 1153, 0, 150)
    assert o["score"] == 150 - (6 + 3) == o["truesc"] and (o["n_comp_is_alt"] & 0x3fffffff) == 3 and o["seedcov"] == 70
    assert o["w"] == 3 + 100 + 100
    regs12 = np.concatenate([_reg(1082, 1162, 70, 150, 80), _reg(1000, 1070, 0, 70, 70)])
    read12 = np.concatenate([g[1000:1070], g[1082:1162]])
    e12, c12 = simulate.flatten_reads([read12])
    assert len(loader.regs_finish(regs12, np.array([0, 2]), e12, c12, ref, l_pac)[0]) == 2
    # redundant hit: a region inside a better one is dropped, the better one stays; purged input regions are ignored
    regs = np.concatenate([_reg(5000, 5150, 0, 150, 150), _reg(5010, 5140, 10, 140, 90), _reg(-1, -1, -1, -1, 7)])
    regs[2]["qb"] = regs[2]["qe"] = -1
    out, off = loader.regs_finish(regs, np.array([0, 3]), enc, cum, ref, l_pac)
    assert len(out) == 1 and out[0]["score"] == 150
    # identical hits (same score, rb, qb) collapse; different contigs never interact; ALT mark
    contigs = np.zeros(2, loader.CONTIG_DTYPE)
    contigs["offset"], contigs["len"], contigs["is_alt"] = [0, 20000], [20000, l_pac - 20000], [0, 1]
    regs = np.concatenate([_reg(100, 200, 0, 100, 100), _reg(100, 230, 0, 130, 100), _reg(25000, 25100, 0, 100, 95, rid=1)])
    out, off = loader.regs_finish(regs, np.array([0, 3]), enc, cum, ref, l_pac, contigs=contigs)
    assert [int(x) for x in out["score"]] == [100, 95] and (out["n_comp_is_alt"] >> 30).tolist() == [0, 1]


def test_whole_path_invariants():
    g, idx = util.toy(30000)
    reads, _, _ = util.toy_reads(30000, 400, 21)
    rng = np.random.default_rng(3)
    extra = []
    for _ in range(12):        # long reads with a deletion wider than the chaining band: two chains, two regions, patched into one
        st = int(rng.integers(0, len(g) - 6000))
        gap = int(rng.integers(120, 200))
        r = np.concatenate([g[st:st + 2500], g[st + 2500 + gap:st + 5200 + gap]])
        extra.append(simulate.revcomp(r) if rng.random() < 0.5 else r)
    enc, cum = simulate.flatten_reads(list(reads) + extra)
    o = loader.OracleFMI(idx)
    sm = o.collect_smem(enc, cum)
    coord, off = o.sa_lookup(sm)
    l_pac = len(g)
    ref = np.concatenate([g, (3 - g[::-1]).astype(np.uint8)])
    ch, sd, choff = loader.chain_seeds(sm, coord, off, cum, l_pac, ref_string=ref, enc=enc)
    regs, reg_off, _ = loader.chain2aln(ch, sd, choff, enc, cum, ref, l_pac)
    out, out_off = loader.regs_finish(regs, reg_off, enc, cum, ref, l_pac)
    alive = ~((regs["qb"] == -1) & (regs["qe"] == -1))
    assert 0 < len(out) <= alive.sum() and out_off[-1] == len(out)
    merged = (out["n_comp_is_alt"] & 0x3fffffff) > 1
    assert merged.sum() > 0                                # mem_patch_reg did merge split alignments
    for r in range(len(cum) - 1):
        a = out[out_off[r]:out_off[r + 1]]
        if len(a) > 1:                                     # sorted by score desc, then rb, qb; no identical (score, rb, qb)
            key = list(zip(-a["score"], a["rb"], a["qb"]))
            assert key == sorted(key) and len(set(key)) == len(key)
        assert np.all(a["qe"] > a["qb"])


def test_pestat_from_definition():
    """mem_pestat: orientation / insert size per pair (mem_infer_dir), the 'unique pair' selection, percentile bounds,
    mean and std — recomputed in numpy for constructed final regions."""
    l_pac = 1_000_000
    rng = np.random.default_rng(2)
    regs, off = [], [0]
    ins_fr = []
    for i in range(400):
        p = int(rng.integers(1000, l_pac - 2000))
        ins = int(rng.normal(350, 30))
        a = _reg(p, p + 150, 0, 150, 150)                                         # read 1 forward at p
        rb2 = 2 * l_pac - 1 - (p + ins)                                           # read 2 on the reverse strand, 5' end at p + ins
        b = _reg(rb2, rb2 + 150, 0, 150, 150)
        regs += [a, b]
        off += [off[-1] + 1, off[-1] + 2]
        ins_fr.append(ins)
        if i % 50 == 0:                                                           # a pair with a strong second hit: not unique
            regs.insert(len(regs) - 1, _reg(p + 5000, p + 5150, 0, 150, 140))
            off[-2] += 1; off[-1] += 1
            ins_fr.pop()
    regs = np.concatenate(regs)
    pes = loader.pestat(regs, np.array(off), l_pac)
    assert list(pes["failed"]) == [1, 0, 1, 1]
    q = np.sort(np.array(ins_fr, dtype=np.uint64))
    n = len(q)
    p25, p75 = int(q[int(.25 * n + .499)]), int(q[int(.75 * n + .499)])
    low, high = max(1, int(p25 - 2.0 * (p75 - p25) + .499)), int(p75 + 2.0 * (p75 - p25) + .499)
    sel = q[(q >= low) & (q <= high)].astype(np.float64)
    avg = 0.0
    for v in sel:
        avg += v
    avg /= len(sel)
    sd = 0.0
    for v in sel:
        sd += (v - avg) * (v - avg)
    sd = (sd / len(sel)) ** 0.5
    assert pes["avg"][1] == avg and pes["std"][1] == sd
    lo2, hi2 = int(p25 - 3.0 * (p75 - p25) + .499), int(p75 + 3.0 * (p75 - p25) + .499)
    if lo2 > avg - 4.0 * sd:
        lo2 = int(avg - 4.0 * sd + .499)
    if hi2 < avg + 4.0 * sd:
        hi2 = int(avg + 4.0 * sd + .499)
    assert (pes["low"][1], pes["high"][1]) == (max(1, lo2), hi2)
