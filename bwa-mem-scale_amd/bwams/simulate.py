"""Synthetic genomes and reads (SURVEY.md §8d "Synthetic inputs").

The real GRCh38 / E. coli FASTA files are not available offline, so every
configuration runs on a seeded synthetic genome of a stated size.  Reads follow
the recipe of SURVEY.md §8d: 150 bp, both strands, per-read error profile drawn
from {0, 0, 0.5 %, 2 %, 5 %} substitutions with indel rate = sub/5, 1 % of reads
carry one N, 2 % are random sequence.

Base codes follow the reference (src/bwa.cpp nst_nt4_table): A=0 C=1 G=2 T=3 N=4.
"""
from __future__ import annotations

import numpy as np

READ_LEN = 150
_ERR_PROFILE = np.array([0.0, 0.0, 0.005, 0.02, 0.05])


def make_genome(n_bases: int, seed: int = 2024, repeat_frac: float = 0.10,
                repeat_len: int = 300, n_families: int = 8,
                repeat_div: float = 0.08, profile: str | None = None, return_holes: bool = False):
    """Random genome (codes 0..3) with interspersed repeat families.

    A fraction ``repeat_frac`` of the genome is overwritten with diverged copies
    (substitution rate ``repeat_div``) of ``n_families`` consensus elements, so
    SMEM intervals with s > 1 and multi-seed chains occur as they do on a real
    genome (a uniformly random genome has essentially none).

    ``profile="grch38_like"`` adds what makes the human reference hard for a seed-and-extend aligner (see
    _grch38_like); ``return_holes`` then also returns the N holes as (offset, len) records.
    """
    if profile is not None:
        if profile != "grch38_like":
            raise ValueError(f"unknown genome profile {profile!r}")
        g, holes = _grch38_like(n_bases, seed)
        return (g, holes) if return_holes else g
    rng = np.random.default_rng(seed)
    g = rng.integers(0, 4, size=n_bases, dtype=np.uint8)
    if repeat_frac > 0 and n_bases >= 4 * repeat_len:
        fams = rng.integers(0, 4, size=(n_families, repeat_len), dtype=np.uint8)
        n_copies = int(n_bases * repeat_frac / repeat_len)
        starts = rng.integers(0, n_bases - repeat_len, size=n_copies)
        fam_id = rng.integers(0, n_families, size=n_copies)
        # process in slabs to bound memory
        slab = 1 << 16
        for a in range(0, n_copies, slab):
            st = starts[a:a + slab]
            cp = fams[fam_id[a:a + slab]].copy()
            mut = rng.random(cp.shape) < repeat_div
            cp[mut] = (cp[mut] + rng.integers(1, 4, size=int(mut.sum()), dtype=np.uint8)) & 3
            idx = st[:, None] + np.arange(repeat_len)[None, :]
            g[idx.ravel()] = cp.ravel()
    return g


_MICROSAT = ("A", "CA", "GA", "TA", "AAT", "CAG", "GATA", "GGAA", "TTAGGG", "AAAG")
HOLE_DTYPE = np.dtype([("offset", "<i8"), ("len", "<i8")])


def _interspersed(g, rng, frac, unit_len, n_families, div, truncate=False):
    """overwrite a fraction of g with diverged copies of consensus elements (truncate: keep a random-length 3' part, as L1 copies do)"""
    n_bases = len(g)
    if n_bases < 4 * unit_len:
        return
    fams = rng.integers(0, 4, size=(n_families, unit_len), dtype=np.uint8)
    mean_len = unit_len * (0.55 if truncate else 1.0)
    n_copies = int(n_bases * frac / mean_len)
    slab = max(1, (1 << 24) // unit_len)
    for a in range(0, n_copies, slab):
        k = min(slab, n_copies - a)
        st = rng.integers(0, n_bases - unit_len, size=k)
        cp = fams[rng.integers(0, n_families, size=k)].copy()
        mut = rng.random(cp.shape) < div
        cp[mut] = (cp[mut] + rng.integers(1, 4, size=int(mut.sum()), dtype=np.uint8)) & 3
        idx = st[:, None] + np.arange(unit_len)[None, :]
        if truncate:
            keep = np.arange(unit_len)[None, :] >= rng.integers(0, unit_len - unit_len // 10, size=k)[:, None]
            g[idx[keep]] = cp[keep]
        else:
            g[idx.ravel()] = cp.ravel()


def _grch38_like(n_bases: int, seed: int):
    """A synthetic genome with the structures of GRCh38 that stress seeding, chaining and index construction:

      * interspersed repeats: an Alu-like family set (300 bp, ~10 % of the bases, 12 % diverged) and an L1-like one (6 kb
        consensus, 5'-truncated copies, ~15 %, 5 % diverged);
      * satellite arrays: tandem arrays of 171-bp monomers organised as 12-monomer higher-order repeats (monomers of one
        HOR 20 % apart, HOR copies 0.5 % apart, the first 20 HOR copies of an array exact), 10^4 monomers per array at
        GRCh38 size (fewer on small genomes), about one array per 130 Mbp: seeds with 10^3..10^4 hits;
      * microsatellites (exact (A)n, (CA)n, (GATA)n, (TTAGGG)n ... runs of 20-400 bp, mean 50, one per 20 kb) and poly-A /
        poly-T runs of 15-45 bp (one per 30 kb): suffixes that agree for hundreds of bases, k-mers with 10^6 hits.  Their
        number and length stay within what the ERT format can hold, as GRCh38 does: a leaf's hits are counted in 16 bits
        (src/ertindex.cpp:336-352: no 151-base string 65536 times) and a k-mer's tree is below 64 MiB (:452, :607: A^15 with
        more than some 10^7 hits has no tree) — beyond either the reference's writer does not finish and bwams_ert_build
        refuses the text;
      * segmental duplications: four exact copies of >= 100 kb (on genomes of 4 Mbp and more), one of them inverted: the
        longest common prefixes a suffix sort meets, and reads with two perfect placements;
      * N holes: short ones (1-50 bases) and one centromere-sized one beside every satellite array, filled with random
        bases the way bns_fasta2bntseq does (src/bntseq.cpp:275-279: `c = lrand48() & 3` for a base code >= 4; the holes
        go to the .amb records, which the MEM path does not read).

    Returns (codes 0..3, holes as HOLE_DTYPE records sorted by offset)."""
    rng = np.random.default_rng(seed)
    g = rng.integers(0, 4, size=n_bases, dtype=np.uint8)
    _interspersed(g, rng, 0.10, 300, 8, 0.12)
    _interspersed(g, rng, 0.15, 6000, 3, 0.05, truncate=True)
    holes = []
    # satellite arrays, each with a large hole beside it
    n_arr = max(1, n_bases // 130_000_000)
    n_mono = int(min(10_000, max(48, n_bases // (171 * 60))))
    n_hor = n_mono // 12
    arr_len = n_hor * 12 * 171
    big_hole = int(min(3_000_000, n_bases // 200))
    cons = rng.integers(0, 4, size=171, dtype=np.uint8)
    if arr_len + big_hole < n_bases // (2 * n_arr):
        slot = n_bases // n_arr
        for a in range(n_arr):
            hor = np.tile(cons, 12)
            mut = rng.random(hor.shape) < 0.20
            hor[mut] = (hor[mut] + rng.integers(1, 4, size=int(mut.sum()), dtype=np.uint8)) & 3
            arr = np.tile(hor, n_hor)
            mut = rng.random(arr.shape) < 0.005
            mut[:min(20, n_hor) * len(hor)] = False
            arr[mut] = (arr[mut] + rng.integers(1, 4, size=int(mut.sum()), dtype=np.uint8)) & 3
            p = a * slot + int(rng.integers(slot // 8, slot - arr_len - big_hole - slot // 8))
            g[p:p + arr_len] = arr
            holes.append((p + arr_len, big_hole))
    # microsatellites and poly-A / poly-T runs
    n_ms = n_bases // 20_000
    ms_pos = rng.integers(0, max(1, n_bases - 400), size=n_ms)
    ms_len = np.minimum(400, 20 + rng.geometric(1.0 / 30.0, size=n_ms))      # one in eighty reaches a read length
    ms_unit = rng.integers(0, len(_MICROSAT), size=n_ms)
    units = [np.array(["ACGT".index(ch) for ch in u], dtype=np.uint8) for u in _MICROSAT]
    tiles = [np.tile(u, 400 // len(u) + 1) for u in units]
    for p, ln, u in zip(ms_pos.tolist(), ms_len.tolist(), ms_unit.tolist()):
        if p + ln <= n_bases:
            g[p:p + ln] = tiles[u][:ln]
    n_pa = n_bases // 30_000
    pa_pos = rng.integers(0, max(1, n_bases - 60), size=n_pa)
    pa_len = rng.integers(15, 46, size=n_pa)
    pa_base = rng.integers(0, 2, size=n_pa) * 3                    # A or T
    if n_pa:
        # vectorised: a run is a slice of ones in a difference array
        for ln in range(15, 46):
            sel = pa_len == ln
            if sel.any():
                idx = pa_pos[sel][:, None] + np.arange(ln)[None, :]
                ok = idx < n_bases
                g[idx[ok]] = np.broadcast_to(pa_base[sel][:, None], idx.shape)[ok].astype(np.uint8)
    # exact segmental duplications, the last one inverted
    seg = int(min(150_000, n_bases // 40))
    if seg >= 1000:
        for k in range(4):
            src = int(rng.integers(0, n_bases - seg))
            dst = int(rng.integers(0, n_bases - seg))
            if abs(src - dst) < seg:
                dst = (src + n_bases // 2) % (n_bases - seg)
            piece = g[src:src + seg].copy()
            g[dst:dst + seg] = piece if k < 3 else (3 - piece[::-1])
    # short holes
    n_sh = max(2, n_bases // 10_000_000)
    sh_pos = rng.integers(0, max(1, n_bases - 50), size=n_sh)
    sh_len = rng.integers(1, 51, size=n_sh)
    holes += [(int(p), int(l)) for p, l in zip(sh_pos, sh_len) if p + l <= n_bases]
    holes.sort()
    for p, l in holes:                                              # bns_fasta2bntseq: a random base for every N
        g[p:p + l] = rng.integers(0, 4, size=l, dtype=np.uint8)
    return g, np.array(holes, dtype=HOLE_DTYPE)


def chromosomes(n_bases: int, n_seq: int = 24) -> np.ndarray:
    """Cut a synthetic genome into `n_seq` sequences whose lengths fall off like the human chromosomes' (longest about
    five times the shortest): bntann1_t's `len` is 32 bits (src/bntseq.h), so a genome beyond 2^31 bases must be several
    sequences, as GRCh38 is.  Returns records with the fields of bwams_contig_t (offset, len, is_alt)."""
    w = np.linspace(5.0, 1.0, n_seq)
    lens = np.floor(w / w.sum() * n_bases).astype(np.int64)
    lens[0] += n_bases - int(lens.sum())
    assert lens.max() < 2 ** 31
    out = np.zeros(n_seq, dtype=np.dtype([("offset", "<i8"), ("len", "<i4"), ("is_alt", "<i4")]))
    out["offset"][1:] = np.cumsum(lens)[:-1]
    out["len"] = lens
    return out


def contig_bounds(contigs) -> np.ndarray:
    """Ascending start offsets plus the total length, the form make_reads / make_read_pairs_bulk take."""
    return np.concatenate([contigs["offset"], [contigs["offset"][-1] + contigs["len"][-1]]]).astype(np.int64)


def revcomp(x: np.ndarray) -> np.ndarray:
    """Reverse complement of a code array; N (4) stays N."""
    r = x[..., ::-1]
    return np.where(r < 4, 3 - r, r).astype(np.uint8)


def make_reads(genome: np.ndarray, n_reads: int, seed: int = 12345,
               read_len: int = READ_LEN, contig_bounds: np.ndarray | None = None):
    """Return (reads[n_reads, read_len] uint8 codes, truth_pos int64, truth_rev bool).

    ``contig_bounds`` (ascending start offsets plus the total length) keeps reads
    from straddling contigs; None treats the genome as one contig.
    """
    rng = np.random.default_rng(seed)
    n = genome.shape[0]
    win = read_len + 24                      # slack for deletions
    if contig_bounds is None:
        pos = rng.integers(0, n - win, size=n_reads)
    else:
        cb = np.asarray(contig_bounds, dtype=np.int64)
        lens = np.diff(cb)
        ok = lens > win
        w = np.where(ok, lens - win, 0).astype(np.float64)
        c = rng.choice(len(lens), size=n_reads, p=w / w.sum())
        pos = cb[c] + (rng.random(n_reads) * (lens[c] - win)).astype(np.int64)

    reads = np.empty((n_reads, read_len), dtype=np.uint8)
    prof = _ERR_PROFILE[rng.integers(0, len(_ERR_PROFILE), size=n_reads)]
    is_rev = rng.random(n_reads) < 0.5
    slab = 1 << 17
    col = np.arange(win)[None, :]
    for a in range(0, n_reads, slab):
        b = min(a + slab, n_reads)
        m = b - a
        w_ = genome[(pos[a:b, None] + col).ravel()].reshape(m, win)
        sub = prof[a:b, None]
        ind = sub / 5.0
        u = rng.random((m, read_len))
        is_del = u < ind / 2
        is_ins = (u >= ind / 2) & (u < ind)
        shift = np.cumsum(is_del.astype(np.int32) - is_ins.astype(np.int32), axis=1)
        src = np.clip(np.arange(read_len)[None, :] + shift, 0, win - 1)
        r = np.take_along_axis(w_, src, axis=1)
        rnd = rng.integers(0, 4, size=(m, read_len), dtype=np.uint8)
        r = np.where(is_ins, rnd, r)
        is_sub = rng.random((m, read_len)) < sub
        r = np.where(is_sub, (r + 1 + (rnd % 3)) & 3, r).astype(np.uint8)
        rv = is_rev[a:b]
        r[rv] = revcomp(r[rv])
        reads[a:b] = r
    # 2 % random sequence
    rand_read = rng.random(n_reads) < 0.02
    k = int(rand_read.sum())
    if k:
        reads[rand_read] = rng.integers(0, 4, size=(k, read_len), dtype=np.uint8)
    # 1 % carry one N
    has_n = np.flatnonzero(rng.random(n_reads) < 0.01)
    if has_n.size:
        reads[has_n, rng.integers(0, read_len, size=has_n.size)] = 4
    truth = np.where(rand_read, -1, pos)
    return reads, truth.astype(np.int64), is_rev



def make_read_pairs(genome: np.ndarray, n_pairs: int, seed: int = 777, read_len: int = READ_LEN,
                    insert_mean: float = 400.0, insert_sd: float = 40.0, damaged_frac: float = 0.15,
                    discordant_frac: float = 0.05):
    """Paired-end reads (FR library): list of 2 * n_pairs arrays, ends of pair p at 2p and 2p + 1.

    A ``damaged_frac`` of the pairs carry one end with ~12 % substitutions (too few exact seeds to be found by
    seeding: the case mate rescue exists for); a ``discordant_frac`` have their second end drawn from an
    unrelated position.  Half of the pairs are flipped as a whole (fragment from the reverse strand)."""
    rng = np.random.default_rng(seed)
    n = genome.shape[0]
    out = []
    for _ in range(n_pairs):
        isz = int(max(read_len + 20, rng.normal(insert_mean, insert_sd)))
        p = int(rng.integers(0, n - isz - 1))
        frag = genome[p:p + isz]
        if rng.random() < 0.5:
            frag = revcomp(frag)
        e1 = frag[:read_len].copy()
        e2 = revcomp(frag[-read_len:]).copy()
        if rng.random() < discordant_frac:
            q = int(rng.integers(0, n - read_len - 1))
            e2 = genome[q:q + read_len].copy()
        for e in (e1, e2):                                   # sequencing errors
            m = rng.random(read_len) < 0.01
            e[m] = (e[m] + 1 + rng.integers(0, 3, size=int(m.sum()))) & 3
        if rng.random() < damaged_frac:
            e = e2 if rng.random() < 0.5 else e1
            m = rng.random(read_len) < 0.12
            e[m] = (e[m] + 1 + rng.integers(0, 3, size=int(m.sum()))) & 3
        out.append(e1.astype(np.uint8))
        out.append(e2.astype(np.uint8))
    return out


def make_read_pairs_bulk(genome: np.ndarray, n_pairs: int, seed: int = 777, read_len: int = READ_LEN,
                         insert_mean: float = 400.0, insert_sd: float = 40.0, damaged_frac: float = 0.05,
                         discordant_frac: float = 0.02, contig_bounds: np.ndarray | None = None) -> np.ndarray:
    """Vectorised form of make_read_pairs for large batches: uint8[2 * n_pairs, read_len], ends of pair p in rows
    2p and 2p + 1 (FR library, 1 % substitutions, a damaged_frac with one end at ~12 % substitutions, a
    discordant_frac with the second end from an unrelated position)."""
    rng = np.random.default_rng(seed)
    n = genome.shape[0]
    isz = np.maximum(read_len + 20, rng.normal(insert_mean, insert_sd, size=n_pairs)).astype(np.int64)
    pos = (rng.random(n_pairs) * (n - isz - 1)).astype(np.int64)
    if contig_bounds is not None:                            # a fragment that would straddle two sequences starts at the later one
        cb = np.asarray(contig_bounds, dtype=np.int64)
        c = np.searchsorted(cb, pos + isz, side="right") - 1
        pos = np.where(pos < cb[c], cb[c], pos)
    col = np.arange(read_len)[None, :]
    out = np.empty((2 * n_pairs, read_len), dtype=np.uint8)
    slab = 1 << 17
    for a in range(0, n_pairs, slab):
        b = min(a + slab, n_pairs)
        left = genome[(pos[a:b, None] + col)]
        right = genome[(pos[a:b, None] + isz[a:b, None] - read_len + col)]
        flip = rng.random(b - a) < 0.5
        e1 = np.where(flip[:, None], revcomp(right), left)
        e2 = np.where(flip[:, None], left, revcomp(right))
        disc = rng.random(b - a) < discordant_frac
        if disc.any():
            q = rng.integers(0, n - read_len - 1, size=int(disc.sum()))
            e2[disc] = genome[q[:, None] + col]
        dmg = rng.random(b - a) < damaged_frac
        which = rng.random(b - a) < 0.5
        for e, sel in ((e1, dmg & which), (e2, dmg & ~which)):
            rate = np.where(sel, 0.12, 0.01)[:, None]
            m = rng.random(e.shape) < rate
            e[m] = (e[m] + 1 + rng.integers(0, 3, size=int(m.sum()))) & 3
        out[2 * a:2 * b:2] = e1
        out[2 * a + 1:2 * b:2] = e2
    return out

def flatten_reads(reads) -> tuple[np.ndarray, np.ndarray]:
    """(enc_qdb bytes, cum_len int64[n+1]) from a 2-D array or a list of 1-D arrays.

    enc_qdb is the reference's concatenated one-code-per-byte read buffer
    (src/bwamem.cpp:700-706)."""
    if isinstance(reads, np.ndarray) and reads.ndim == 2:
        n, L = reads.shape
        return np.ascontiguousarray(reads).reshape(-1), np.arange(n + 1, dtype=np.int64) * L
    lens = np.array([len(r) for r in reads], dtype=np.int64)
    cum = np.zeros(len(reads) + 1, dtype=np.int64)
    np.cumsum(lens, out=cum[1:])
    enc = np.concatenate([np.asarray(r, dtype=np.uint8) for r in reads]) if len(reads) else np.zeros(0, np.uint8)
    return enc, cum
