"""bwams_ert_build (ERT index construction on the GPU) against the CPU restatement of the reference's writer
(oracle/ert_oracle.c = src/ertindex.cpp): the k-mer table and the tree bytes must be identical, for every entry kind
(INVALID / SINGLE_HIT_LEAF / INFREQUENT trees / FREQUENT x-mer tables), pointer widths 2-4 and multi-hit leaves."""
import os

import numpy as np
import pytest

from bwams import capi, fmindex, simulate
from oracle import loader

pytestmark = pytest.mark.gpu


def _both(n_bases, seed, kmer, xmer, thr, read_len=151, repeat_frac=0.3, repeat_len=200, n_families=3):
    g = simulate.make_genome(n_bases, seed=seed, repeat_frac=repeat_frac, repeat_len=repeat_len, n_families=n_families)
    idx = fmindex.build_fmindex(g)
    text = fmindex.fw_rc_text(g)
    o = loader.OracleFMI(idx)
    e = loader.OracleERT(o, text, kmer=kmer, xmer=xmer, read_len=read_len, hit_threshold=thr)
    ix = capi.Index.from_host(idx, 0)
    ert = capi.Ert.build(ix, kmer=kmer, xmer=xmer, read_len=read_len, hit_threshold=thr)
    return g, idx, o, e, ix, ert


def _same(e, ert):
    kt, mt = ert.fetch()
    info = ert.info()
    assert info["mlt_bytes"] == len(e.mlt)
    bad = np.flatnonzero(kt != e.kmer_table)
    assert bad.size == 0, (bad[:5], [hex(int(x)) for x in kt[bad[:5]]], [hex(int(x)) for x in e.kmer_table[bad[:5]]])
    badb = np.flatnonzero(mt != e.mlt)
    assert badb.size == 0, (badb[:10], mt[badb[:10]], e.mlt[badb[:10]])


@pytest.mark.parametrize("kmer,xmer,thr,n_bases,seed,read_len", [
    (6, 2, 8, 20000, 1, 151),       # every entry kind, x-mer tables
    (8, 2, 16, 100000, 2, 151),
    (5, 3, 4, 4000, 3, 151),        # dense tables over a tiny text (suffixes running into the end of the text)
    (10, 4, 256, 300000, 4, 151),   # the reference's threshold
    (7, 2, 1000000, 50000, 5, 40),  # trees only, short read length: multi-hit leaves at max depth
    (4, 2, 100000, 30000, 6, 151),  # big trees under few k-mers: 3- and 4-byte pointers
])
def test_built_tables_equal_the_writer(kmer, xmer, thr, n_bases, seed, read_len):
    g, idx, o, e, ix, ert = _both(n_bases, seed, kmer, xmer, thr, read_len=read_len)
    widths = set(((e.kmer_table >> 22) & 3)[(e.kmer_table & 3) >= 2].tolist())
    _same(e, ert)
    if kmer == 4:
        assert {3, 0} & widths, widths           # 3-byte (code 3) or 4-byte (code 0) pointers were exercised
    ert.close(); ix.close()


def test_seeding_over_built_tables_and_save_open(tmp_path):
    g, idx, o, e, ix, ert = _both(200000, 9, 9, 3, 64, repeat_frac=0.2)
    _same(e, ert)
    reads, _, _ = simulate.make_reads(g, 2000, seed=3)
    enc, cum = simulate.flatten_reads(reads)
    go = capi.default_seed_opt()
    b = capi.Batch(ix, len(cum) - 1, int(cum[-1]))
    b.seed_upload(enc, cum)
    b.seed_run_ert(ert, go)
    sm_e, co_e, off_e = b.seed_fetch()
    b.seed_run(go)
    sm_f, co_f, off_f = b.seed_fetch()
    for f in ("rid", "m", "n", "s"):
        assert np.array_equal(sm_e[f], sm_f[f]), f
    assert np.array_equal(off_e, off_f) and np.all((co_e == co_f) | ((co_f == 0) & (co_e < 128)))
    # the reference's files: written, and (for k = 15 only) read back by bwams_ert_open
    prefix = os.path.join(tmp_path, "toy")
    ert.save(prefix)
    kt = np.fromfile(prefix + ".kmer_table", dtype=np.uint64)
    mt = np.fromfile(prefix + ".mlt_table", dtype=np.uint8)
    assert np.array_equal(kt, e.kmer_table) and np.array_equal(mt, e.mlt)
    b.close(); ert.close(); ix.close()


def test_real_kmer_size_build_save_open(tmp_path):
    """k = 15, x = 4, HIT_THRESHOLD 256 on a 2 Mbp text: built on the GPU, equal to the writer, saved, reopened."""
    g, idx, o, e, ix, ert = _both(2_000_000, 12, 15, 4, 256, repeat_frac=0.1)
    _same(e, ert)
    prefix = os.path.join(tmp_path, "k15")
    ert.save(prefix)
    assert os.path.getsize(prefix + ".kmer_table") == 8 << 30
    ert2 = capi.Ert(ix, prefix=prefix, read_len=151)
    reads, _, _ = simulate.make_reads(g, 1000, seed=5)
    enc, cum = simulate.flatten_reads(reads)
    b = capi.Batch(ix, len(cum) - 1, int(cum[-1]))
    b.seed_upload(enc, cum)
    b.seed_run_ert(ert, capi.default_seed_opt())
    a1 = b.seed_fetch()
    b.seed_run_ert(ert2, capi.default_seed_opt())
    a2 = b.seed_fetch()
    assert all(np.array_equal(x, y) for x, y in zip(a1, a2)) and len(a1[0]) > 1000
    os.remove(prefix + ".kmer_table"); os.remove(prefix + ".mlt_table")
    b.close(); ert.close(); ert2.close(); ix.close()


def test_a_leaf_of_65536_hits_is_refused_and_the_hard_genome_profile_builds():
    """The format counts the hits of a multi-hit leaf (a string of read_len bases) in 16 bits, and the reference's writer loops for ever
    on 65536 of them (ertindex.cpp:336-352, `uint16_t k`): such a text is refused with an error, not written truncated.  The
    grch38_like profile (satellite arrays, microsatellites, poly-A, segmental duplications) stays below that and builds to the writer's bytes."""
    rng = np.random.default_rng(4)
    g = rng.integers(0, 4, size=120000, dtype=np.uint8)
    g[20000:20000 + 65600] = 0                                 # A^40 occurs 65 561 times
    ix = capi.Index.build(g, 0)
    with pytest.raises(capi.BwamsError, match="65536"):
        capi.Ert.build(ix, kmer=8, xmer=2, read_len=40, hit_threshold=64)
    ert = capi.Ert.build(ix, kmer=8, xmer=2, read_len=255, hit_threshold=64)     # at 255 bases the leaf has 65 346
    ert.close(); ix.close()
    g = simulate.make_genome(150000, seed=8, profile="grch38_like")
    idx = fmindex.build_fmindex(g)
    o = loader.OracleFMI(idx)
    e = loader.OracleERT(o, fmindex.fw_rc_text(g), kmer=8, xmer=2, read_len=151, hit_threshold=32)
    ix = capi.Index.from_host(idx, 0)
    ert = capi.Ert.build(ix, kmer=8, xmer=2, read_len=151, hit_threshold=32)
    _same(e, ert)
    ert.close(); ix.close()
