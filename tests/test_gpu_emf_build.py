"""bwams_emf_build (EMF table construction on the GPU).  The reference's builder places collision nodes by an
order-dependent probe, so tables are compared by what they answer, not byte for byte:
  * structural invariants of the format, checked on the host over the fetched arrays: every L-mer of the forward strand
    is found by a walk from its bucket's slot (root not flagged, other nodes flagged, ascending chain), stored once, with
    every one of its locations (first in the entry, the rest in loc_table, split by orientation);
  * the restated find_perfect_match_entry (oracle) over the GPU-built table == over the host-built table (emf.build_emf)
    in code and location, for reads of L and more bases, both strands, mismatches and N;
  * the HIP probe over both tables agrees as well; the saved file is the reference's `.perfect.<L>` layout."""
import numpy as np
import pytest

from bwams import capi, emf, fmindex, simulate
from oracle import loader

pytestmark = pytest.mark.gpu


def _genome(n, seed, microsat=False):
    g = simulate.make_genome(n, seed=seed, repeat_frac=0.2, repeat_len=300, n_families=3, repeat_div=0.0)   # exact copies
    g[5000:5400] = g[1000:1400]
    g[9000:9400] = (3 - g[1000:1400][::-1])                  # a reverse-complement copy
    if microsat:
        # a poly-T run inside a (GA)n microsatellite, a poly-A run inside (CA)n: the table's hash (an XOR of 32-base words) is the same
        # wherever the run sits in the window, so ~90 DISTINCT L-mers share one bucket (met on the grch38_like genome at full size)
        ga, ca = np.tile(np.array([2, 0], np.uint8), 150), np.tile(np.array([1, 0], np.uint8), 130)
        g[12000:12632] = np.concatenate([ga, np.full(32, 3, np.uint8), ga])
        g[15000:15560] = np.concatenate([ca, np.full(40, 0, np.uint8), ca])
    return g


def _reads(g, L, n, seed):
    rng = np.random.default_rng(seed)
    reads = []
    for it in range(n):
        ln = L if it % 3 else int(rng.integers(L + 1, L + 70))
        st = int(rng.integers(0, len(g) - ln))
        rd = g[st: st + ln].copy()
        k = it % 6
        if k == 1:
            rd = (3 - rd[::-1]).astype(np.uint8)
        elif k == 2:
            rd[rng.integers(0, ln)] ^= 1
        elif k == 3 and it % 12 == 3:
            rd[rng.integers(0, ln)] = 4
        elif k == 4:
            rd[rng.integers(L - 1, ln)] ^= 2
        reads.append(rd)
    reads.append(g[0:L].copy())
    reads.append(g[len(g) - L:].copy())
    reads.append(g[0:L - 1].copy())
    return reads


def _canon(w):
    L = len(w)
    half = (L + 1) // 2
    rc = (3 - w[::-1]).astype(np.uint8)
    fw = bytes(w[:half]) <= bytes(rc[:half])
    return (bytes(w) if fw else bytes(rc)), fw


@pytest.mark.parametrize("L,n_bases,slack,microsat", [(50, 40000, 1.1, False), (150, 60000, 1.1, False), (33, 20000, 1.0, False),
                                                       (151, 30000, 2.0, False), (150, 30000, 1.1, True), (64, 30000, 1.1, True)])
def test_built_table_invariants_and_answers(L, n_bases, slack, microsat, tmp_path):
    g = _genome(n_bases, L, microsat)
    idx = fmindex.build_fmindex(g)
    ix = capi.Index.from_host(idx, 0)
    e = capi.Emf.build(ix, seed_len=L, slack=slack)
    info = e.info()
    loc, seeds = e.fetch_table()
    n_entry = info["num_seed_entry"]
    assert n_entry == max(int(slack * len(g)), 16) and len(seeds) == n_entry
    # every window, by a host walk of the fetched table
    want = {}
    for p in range(len(g) - L + 1):
        c, fw = _canon(g[p:p + L])
        want.setdefault(c, []).append((p, fw))
    assert info["n_used"] == len(want)
    ref = g
    found = 0
    roots = set()
    for c, occ in want.items():
        h = np.uint64(0)
        ca = np.frombuffer(c, dtype=np.uint8)
        full = L - L % 32
        with np.errstate(over="ignore"):
            for w0 in list(range(0, full, 32)) + ([full] if L % 32 else []):
                word = 0
                for b in ca[w0:min(w0 + 32, L)]:
                    word = (word << 2) | int(b)
                h ^= np.uint64(word)
            key = int(emf._fmix64(np.array([h]))[0] % np.uint64(n_entry))
        roots.add(key)
        slot, first, prev = key, True, None
        while True:
            fl, location, left, right = (int(x) for x in seeds[slot])
            assert location != emf.NO_ENTRY and bool(fl & 2) == (not first) and left == emf.NO_ENTRY
            ec, _ = _canon(ref[location:location + L])
            assert prev is None or prev < ec                  # ascending chain
            if ec == c:
                break
            assert ec < c and right != emf.NO_ENTRY, "L-mer not in its bucket"
            prev, slot, first = ec, right, False
        p0, f0 = occ[0]
        assert location == p0 and (fl & 1) == int(f0)
        same = [p for p, f in occ[1:] if f == f0]
        other = [p for p, f in occ[1:] if f != f0]
        if len(occ) > 1:
            m = fl >> 2
            assert m > 0
            assert int(loc[m]) == (len(same) << 16 | len(other))
            assert list(loc[m + 1:m + 1 + len(same)]) == same and list(loc[m + 1 + len(same):m + 1 + len(occ) - 1]) == other
        else:
            assert fl >> 2 == 0
        found += 1
    assert found == len(want) and info["n_key"] == len(roots)
    if microsat and L == 150:                                # the case is what it says: a bucket with more L-mers than a lane keeps (48)
        chain = np.zeros(n_entry, np.int64)
        for k in roots:
            s_, d = k, 0
            while s_ != emf.NO_ENTRY:
                d += 1
                s_ = int(seeds[s_][3])
            chain[k] = d
        assert chain.max() > 48
    used = seeds[:, 1] != emf.NO_ENTRY
    assert int(used.sum()) == len(want)                      # nothing else in the table
    # answers: oracle over this table == oracle over the host builder's table == the HIP probe over both
    reads = _reads(g, L, 1500, L)
    tab_gpu = emf.EmfTable(L, len(g), loc.copy(), seeds.copy(), info["n_used"], info["n_key"])
    tab_host = emf.build_emf(g, L, slack=slack)
    a = loader.OracleEMF(tab_gpu, idx.ref_0123).probe_many(reads)
    b_ = loader.OracleEMF(tab_host, idx.ref_0123).probe_many(reads)
    assert np.array_equal(a[:, 0], b_[:, 0]) and np.array_equal(a[:, 2], b_[:, 2])
    assert np.array_equal(a[:, 1] & 3, b_[:, 1] & 3)
    enc, cum = simulate.flatten_reads(reads)
    bt = capi.Batch(ix, len(reads), int(cum[-1]))
    perfect, code = bt.emf_probe(e, enc, cum)
    hit = (code == 3) | (code == 4)
    assert np.array_equal(code, a[:, 0].astype(np.uint8)) and np.array_equal(perfect[hit, 1], a[hit, 2].astype(np.uint32))
    assert np.array_equal(perfect[hit, 0], a[hit, 1].astype(np.uint32)) and hit.sum() > 300 and set(code) >= {0, 2, 3, 4}
    # the file
    path = str(tmp_path / f"g.perfect.{L}")
    e.save(path)
    t2 = emf.read_emf(path)
    assert t2.seed_len == L and t2.seq_len == len(g) and np.array_equal(t2.loc_table, loc) and np.array_equal(t2.seed_table, seeds)
    e2 = capi.Emf(ix, path=path)
    perfect2, code2 = bt.emf_probe(e2, enc, cum)
    assert np.array_equal(code2, code) and np.array_equal(perfect2[hit], perfect[hit])
    bt.close(); e.close(); e2.close(); ix.close()
