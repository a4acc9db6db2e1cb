"""Shared fixtures: small seeded genomes/indexes/reads, cached per process."""
from __future__ import annotations

import functools

import numpy as np

from bwams import fmindex, simulate


@functools.lru_cache(maxsize=None)
def toy(n_bases: int = 20000, seed: int = 7, repeat_frac: float = 0.15):
    g = simulate.make_genome(n_bases, seed=seed, repeat_frac=repeat_frac, repeat_len=200, n_families=3)
    idx = fmindex.build_fmindex(g)
    return g, idx


@functools.lru_cache(maxsize=None)
def toy_reads(n_bases: int = 20000, n_reads: int = 300, seed: int = 11):
    g, idx = toy(n_bases)
    reads, pos, rev = simulate.make_reads(g, n_reads, seed=seed)
    return reads, pos, rev


def naive_sa(text: np.ndarray) -> np.ndarray:
    """Suffix array of text+'$' by direct comparison (small inputs only)."""
    n = len(text)
    b = bytes((text + 1).astype(np.uint8)) + b"\x00"
    return np.array(sorted(range(n + 1), key=lambda i: b[i:]), dtype=np.int64)


def make_pairs(n: int, seed: int = 5, max_q: int = 140, max_extra_t: int = 120, n_frac: float = 0.02,
               h0_max: int = 150):
    """Random extension tasks in the reference's SeqPair layout.

    Targets are mutated copies of the query (substitutions, indels, or unrelated
    tails) so that every exit path of the DP is exercised: z-drop, band shrink,
    row maximum reaching zero, reaching the query end (gscore)."""
    from oracle.loader import SEQPAIR_DTYPE
    rng = np.random.default_rng(seed)
    pairs = np.zeros(n, dtype=SEQPAIR_DTYPE)
    refs, qers = [], []
    ro = qo = 0
    for i in range(n):
        ql = int(rng.integers(1, max_q + 1))
        q = rng.integers(0, 4, size=ql, dtype=np.uint8)
        mode = rng.integers(0, 5)
        t = list(q)
        if mode >= 1:
            rate = [0.0, 0.02, 0.08, 0.2, 0.5][mode]
            out = []
            for b in t:
                u = rng.random()
                if u < rate * 0.6:
                    out.append((b + rng.integers(1, 4)) & 3)
                elif u < rate * 0.8:
                    continue
                elif u < rate:
                    out.extend([b, rng.integers(0, 4)])
                else:
                    out.append(b)
            t = out
        if rng.random() < 0.3:                     # a long gap somewhere
            cut = int(rng.integers(0, len(t) + 1))
            gap = int(rng.integers(1, 40))
            if rng.random() < 0.5:
                t = t[:cut] + list(rng.integers(0, 4, size=gap)) + t[cut:]
            else:
                t = t[:cut] + t[cut + gap:]
        t = t + list(rng.integers(0, 4, size=int(rng.integers(0, max_extra_t))))
        if len(t) == 0:
            t = [0]
        t = np.array(t, dtype=np.uint8)
        nmask = rng.random(len(t)) < n_frac
        t[nmask] = 4
        q = q.copy()
        q[rng.random(ql) < n_frac] = 4
        pairs[i]["idr"], pairs[i]["idq"], pairs[i]["id"] = ro, qo, i
        pairs[i]["len1"], pairs[i]["len2"] = len(t), ql
        pairs[i]["h0"] = int(rng.integers(1, h0_max + 1))
        pairs[i]["seqid"], pairs[i]["regid"] = i // 3, i % 3
        refs.append(t); qers.append(q)
        ro += len(t); qo += ql
    return pairs, np.concatenate(refs), np.concatenate(qers)


OUT_FIELDS = ("score", "tle", "gtle", "qle", "gscore", "max_off")


def assert_pairs_equal(a, b, what=""):
    for f in OUT_FIELDS:
        bad = np.flatnonzero(a[f] != b[f])
        assert bad.size == 0, f"{what}: field {f} differs at {bad[:5]}: {a[f][bad[:5]]} vs {b[f][bad[:5]]}; pair={a[bad[0]]}"


def make_local_cases(n: int, seed: int = 9, qmax: int = 151, tmax: int = 900):
    """(query, target) pairs shaped like mate rescue: a (mutated, possibly truncated) copy of the
    query somewhere in a longer random window, sometimes twice (second-best hit), sometimes not at all."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        ql = int(rng.integers(20, qmax + 1))
        q = rng.integers(0, 4, size=ql, dtype=np.uint8)
        tl = int(rng.integers(ql, tmax))
        t = rng.integers(0, 4, size=tl, dtype=np.uint8)

        def mutated():
            rate = float(rng.choice([0.0, 0.02, 0.06, 0.15]))
            o = []
            for b in q:
                u = rng.random()
                if u < rate * 0.6:
                    o.append((b + rng.integers(1, 4)) & 3)
                elif u < rate * 0.8:
                    continue
                elif u < rate:
                    o.extend([b, rng.integers(0, 4)])
                else:
                    o.append(b)
            a = int(rng.integers(0, max(1, len(o) // 3)))
            z = int(rng.integers(0, max(1, len(o) // 3)))
            return np.array(o[a: len(o) - z], dtype=np.uint8)
        k = int(rng.choice([0, 1, 1, 1, 2, 3]))
        for _c in range(k):
            c = mutated()
            if len(c) and len(c) < tl:
                st = int(rng.integers(0, tl - len(c) + 1))
                t[st: st + len(c)] = c
        if rng.random() < 0.1:
            t[rng.integers(0, tl, size=3)] = 4
        if rng.random() < 0.1:
            q[rng.integers(0, ql, size=2)] = 4
        out.append((q, t))
    return out


def oracle_pe_pipeline(g, idx, reads, contigs=None, opt=None):
    """Run the CPU oracle from reads to final regions (+ insert-size statistics) for a chunk of read pairs.
    Returns a dict with enc, cum, ref, regs, reg_off, pes."""
    from bwams import simulate
    from oracle import loader
    enc, cum = simulate.flatten_reads(reads)
    opt = opt or loader.default_mem_opt()
    o = loader.OracleFMI(idx)
    sm = o.collect_smem(enc, cum)
    coord, off = o.sa_lookup(sm)
    l_pac = len(g)
    ref = np.concatenate([g, (3 - g[::-1]).astype(np.uint8)])
    ch, sd, choff = loader.chain_seeds(sm, coord, off, cum, l_pac, contigs=contigs, opt=opt, ref_string=ref, enc=enc)
    regs, reg_off, _ = loader.chain2aln(ch, sd, choff, enc, cum, ref, l_pac, contigs=contigs, opt=opt)
    fin, fin_off = loader.regs_finish(regs, reg_off, enc, cum, ref, l_pac, contigs=contigs, opt=opt)
    pes = loader.pestat(fin, fin_off, l_pac, opt=opt)
    return dict(enc=enc, cum=cum, ref=ref, l_pac=l_pac, regs=fin, reg_off=fin_off, pes=pes, opt=opt, sm=sm, coord=coord, off=off)



def ert_mems_from_smems(sm, all_coord, all_off, nseq, l_pac, seed=0, shuffle=True, backward_frac=0.4, dup_frac=0.05):
    """Dress FM-index SMEMs up as the output of the reference's ERT walk (mem_t records + per-read hit arrays).

    all_coord / all_off hold EVERY occurrence of each SMEM (SA lookup with max_occ = infinity), in BWT-row order, so
    that mem_chain_new's strided pick (hits[hitbeg + k], k = 0, step, ..) sees the rows get_sa_entries would.  A
    backward_frac of the MEMs are marked "found by backward search": their hits are stored as the walk stores them
    (position of the reverse-complemented match, off by end_correction) and mem_chain_new maps them back.  MEMs are
    shuffled within a read (the reference sorts them) and a few are duplicated (ties for the unstable sort)."""
    from oracle import loader
    rng = np.random.default_rng(seed)
    mems, hits, mem_off, hit_off = [], [], [0], [0]
    order = np.argsort(sm["rid"], kind="stable")
    by_read = [[] for _ in range(nseq)]
    for i in order:
        by_read[int(sm["rid"][i])].append(int(i))
    for r in range(nseq):
        idxs = list(by_read[r])
        idxs += [i for i in idxs if rng.random() < dup_frac]
        if shuffle:
            rng.shuffle(idxs)
        hb = 0
        for i in idxs:
            pos = all_coord[all_off[i]:all_off[i + 1]].astype(np.int64)
            m = np.zeros(1, loader.ERT_MEM_DTYPE)[0]
            m["start"], m["end"] = int(sm["m"][i]), int(sm["n"][i]) + 1
            slen = int(m["end"]) - int(m["start"])
            m["hitbeg"], m["hitcount"] = hb, len(pos)
            kind = rng.random()
            if kind < backward_frac:
                m["forward"], m["fetch_leaves"] = 0, 0
                m["end_correction"] = int(rng.integers(0, 4))
                stored = 2 * l_pac - pos - slen + int(m["end_correction"])
            elif kind < backward_frac + 0.2:
                m["forward"], m["fetch_leaves"] = 0, 1
                stored = pos
            else:
                m["forward"] = 1
                stored = pos
            hits.append(stored.astype(np.uint64))
            hb += len(pos)
            mems.append(m)
        mem_off.append(len(mems))
        hit_off.append(hit_off[-1] + hb)
    mems = np.array(mems, dtype=loader.ERT_MEM_DTYPE) if mems else np.zeros(0, loader.ERT_MEM_DTYPE)
    hits = np.concatenate(hits) if hits else np.zeros(0, np.uint64)
    return mems, np.array(mem_off, np.int64), hits, np.array(hit_off, np.int64)


def seeds_equal_but_junction(got, gcoord, goff, want, wcoord, woff, cum, l_pac):
    """Seeding results (SMEM records in (rid, m, n) order + sampled coordinates) of two statements of ERT-mode seeding
    must be identical read by read, EXCEPT for reads with a hit whose placement crosses the junction between the two
    strands of the text: there the reference's get_seq (ertseeding.cpp:455-472) hands leaf expansion nothing and its walk
    emits non-maximal matches, which FM-index seeding and the HIP path do not reproduce
    (tests/test_oracle_ert_walk.py::test_strand_junction_is_the_only_difference).  Returns the differing reads."""
    import collections

    def per_read(sm, co, of):
        d = collections.defaultdict(list)
        rid, m, n, s = sm["rid"], sm["m"], sm["n"], sm["s"]
        for t in range(len(sm)):
            d[int(rid[t])].append((int(m[t]), int(n[t]), int(s[t]), tuple(int(x) for x in co[of[t]:of[t + 1]])))
        return d

    if (len(got) == len(want) and all(np.array_equal(got[f], want[f]) for f in ("rid", "m", "n", "s"))
            and np.array_equal(goff, woff) and np.array_equal(gcoord, wcoord)):
        return []
    A, B = per_read(got, gcoord, goff), per_read(want, wcoord, woff)
    bad = sorted(r for r in set(A) | set(B) if A.get(r) != B.get(r))
    for r in bad:
        ln = int(cum[r + 1] - cum[r])
        placements = [p - m for (m, n, s, ps) in A.get(r, []) + B.get(r, []) for p in ps]
        assert any(p < l_pac < p + ln for p in placements), f"read {r}: {A.get(r)} != {B.get(r)}"
    return bad
