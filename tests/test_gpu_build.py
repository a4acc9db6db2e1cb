"""bwams_index_build (FM-index construction on the GPU) against the host builder of bwams/fmindex.py — itself checked
against a naive suffix sort in tests/test_oracle_fmi.py — array by array, and the file it saves byte by byte."""
import os

import numpy as np
import pytest

from bwams import capi, fmindex, simulate
from oracle import loader

pytestmark = pytest.mark.gpu


def _same_index(got: fmindex.FMIndex, want: fmindex.FMIndex):
    assert got.ref_seq_len == want.ref_seq_len and got.sentinel_index == want.sentinel_index
    assert np.array_equal(got.count, want.count)
    assert np.array_equal(got.cp_occ, want.cp_occ)
    assert np.array_equal(got.sa_ms_byte, want.sa_ms_byte) and np.array_equal(got.sa_ls_word, want.sa_ls_word)
    assert np.array_equal(got.ref_0123, want.ref_0123)


def _genomes():
    rng = np.random.default_rng(1)
    out = {
        "one_base": np.array([2], np.uint8),
        "five": np.array([0, 3, 3, 1, 0], np.uint8),
        "len31": rng.integers(0, 4, 31, dtype=np.uint8),
        "len32": rng.integers(0, 4, 32, dtype=np.uint8),
        "len33": rng.integers(0, 4, 33, dtype=np.uint8),
        "len64": rng.integers(0, 4, 64, dtype=np.uint8),          # 2*64 + 1 rows: last CP_OCC block holds one row
        "len95": rng.integers(0, 4, 95, dtype=np.uint8),
        "poly_a": np.zeros(3000, np.uint8),                        # every suffix tied to the end: log2(6000/29) + 1 rounds
        "poly_at": np.tile(np.array([0, 3], np.uint8), 2500),      # its own reverse complement
        "tandem": np.tile(rng.integers(0, 4, 37, dtype=np.uint8), 300),
        "palindrome": None,
        "random_50k": rng.integers(0, 4, 50_000, dtype=np.uint8),
        "repeats_200k": simulate.make_genome(200_000, seed=5, repeat_frac=0.3, repeat_len=500, n_families=2, repeat_div=0.01),
    }
    h = rng.integers(0, 4, 4000, dtype=np.uint8)
    out["palindrome"] = np.concatenate([h, (3 - h[::-1]).astype(np.uint8)])      # fw == rc: every suffix of fw has a twin in rc
    return out


@pytest.mark.parametrize("name", list(_genomes().keys()))
def test_build_equals_host_builder(name):
    g = _genomes()[name]
    want = fmindex.build_fmindex(g)
    for chunk_rows in (0, max(64, (2 * len(g) + 1) // 5)):                       # one chunk / several chunks of the key space
        try:
            ix = capi.Index.build(g, 0, chunk_rows=chunk_rows)
        except capi.BwamsError as e:
            # a degenerate text (one 7-base prefix holds most suffixes) cannot be cut into small chunks: documented refusal
            assert chunk_rows and e.code == -6 and name in ("poly_a", "poly_at", "tandem", "one_base", "five"), (name, e)
            continue
        _same_index(ix.fetch(), want)
        ix.close()


def test_build_from_device_tensor_and_saved_files(tmp_path):
    import torch
    g = simulate.make_genome(120_000, seed=9)
    want = fmindex.build_fmindex(g)
    ix = capi.Index.build(torch.from_numpy(g).to("cuda:0"), 0)
    st = ix.build_stats
    assert st.rows == 2 * len(g) + 1 and st.chunks == 1 and st.rounds >= 1 and st.unresolved_after_first > 0
    _same_index(ix.fetch(), want)
    ix.save(str(tmp_path / "dev"))
    fmindex.write_index(str(tmp_path / "host"), want)
    for suffix in (".bwt.2bit.64", ".0123"):
        a = open(str(tmp_path / "dev") + suffix, "rb").read()
        b = open(str(tmp_path / "host") + suffix, "rb").read()
        assert a == b, suffix
    # the saved files load back through the reference-format reader of the library
    ix2 = capi.Index.open(str(tmp_path / "dev"), 0)
    _same_index(ix2.fetch(), want)
    ix2.close()
    ix.close()


def test_codes_above_3_are_refused():
    g = np.array([0, 1, 4, 2, 3] * 20, np.uint8)
    with pytest.raises(capi.BwamsError) as e:
        capi.Index.build(g, 0)
    assert e.value.code == -3


def test_seeding_on_a_device_built_index_equals_oracle():
    g = simulate.make_genome(150_000, seed=21)
    ix = capi.Index.build(g, 0)
    host = ix.fetch()
    reads, _, _ = simulate.make_reads(g, 1500, seed=8)
    enc, cum = simulate.flatten_reads(reads)
    b = capi.Batch(ix, len(reads), int(cum[-1]))
    sm, coord, off = b.seed(enc, cum)
    o = loader.OracleFMI(host)
    want = o.collect_smem(enc, cum)
    wcoord, woff = o.sa_lookup(want)
    assert len(sm) == len(want) and all(np.array_equal(sm[f], want[f]) for f in ("rid", "m", "n", "k", "l", "s"))
    assert np.array_equal(coord, wcoord) and np.array_equal(off, woff)
    b.close()
    ix.close()
