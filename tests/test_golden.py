"""Committed golden vectors (tests/golden, made by tests/make_golden.py): the oracle must
still reproduce them on CPU, and the HIP path must reproduce them on the GPU."""
import os

import numpy as np
import pytest

from bwams import fmindex, simulate
from oracle import loader
from util import OUT_FIELDS

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _seed_case():
    z = np.load(os.path.join(G, "seed_toy.npz"))
    idx = fmindex.build_fmindex(z["genome"])
    assert np.array_equal(idx.count, z["count"]) and idx.sentinel_index == int(z["sentinel"])
    return z, idx


def _check_seeds(z, sm, coord, off):
    for f in ("rid", "m", "n", "k", "l", "s"):
        assert np.array_equal(sm[f], z["smem_" + f]), f
    assert np.array_equal(coord, z["sa_coord"]) and np.array_equal(off, z["sa_off"])


def test_oracle_reproduces_seed_golden():
    z, idx = _seed_case()
    enc, cum = simulate.flatten_reads(z["reads"])
    o = loader.OracleFMI(idx)
    sm = o.collect_smem(enc, cum)
    coord, off = o.sa_lookup(sm, 500)
    _check_seeds(z, sm, coord, off)


def test_oracle_reproduces_bsw_golden_and_reference_was_consulted():
    z = np.load(os.path.join(G, "bsw_tasks.npz"))
    assert bool(z["checked_against_reference"][0]), "regenerate with oracle/_ref present"
    pairs = np.ascontiguousarray(z["pairs"]).view(loader.SEQPAIR_DTYPE).reshape(-1)
    for w, key in ((100, "out_w100"), (200, "out_w200")):
        ours, _ = loader.bsw_pairs(pairs, z["ref"], z["qer"], w)
        assert np.array_equal(np.stack([ours[f] for f in OUT_FIELDS], axis=1), z[key])


def _ksw_case():
    z = np.load(os.path.join(G, "ksw_cases.npz"))
    qo = np.r_[0, np.cumsum(z["qlen"])]
    to = np.r_[0, np.cumsum(z["tlen"])]
    cases = [(z["qer"][qo[i]:qo[i + 1]], z["ref"][to[i]:to[i + 1]]) for i in range(len(z["qlen"]))]
    return z, cases, qo, to


def _emf_case():
    from bwams import emf
    z = np.load(os.path.join(G, "emf_toy.npz"))
    tab = emf.EmfTable(int(z["seed_len"]), int(z["seq_len"]), z["loc_table"], z["seed_table"])
    ro = np.r_[0, np.cumsum(z["read_len"])]
    reads = [z["reads"][ro[i]:ro[i + 1]] for i in range(len(z["read_len"]))]
    return z, tab, reads


def test_oracle_reproduces_ksw_and_emf_golden():
    z, cases, _, _ = _ksw_case()
    assert bool(z["checked_against_reference"][0]), "regenerate with oracle/_ref present"
    for k, fl in enumerate(z["flags"]):
        got = np.array([loader.ksw_align2(q, t, int(fl)) for q, t in cases], dtype=np.int32)
        assert np.array_equal(got, z["out"][k])
    ze, tab, reads = _emf_case()
    zs, idx = _seed_case()
    o = loader.OracleEMF(tab, idx.ref_0123)
    assert np.array_equal(o.probe_many(reads), ze["expect"])
    assert set(ze["expect"][:, 0]) >= {2, 3, 4}


@pytest.mark.gpu
def test_gpu_reproduces_golden():
    from bwams import capi
    z, idx = _seed_case()
    enc, cum = simulate.flatten_reads(z["reads"])
    ix = capi.Index.from_host(idx, 0)
    b = capi.Batch(ix, len(z["reads"]) + 64, int(cum[-1]) + 16384)
    sm, coord, off = b.seed(enc, cum)
    _check_seeds(z, sm, coord, off)
    st = b.stats()
    c = z["counters"]
    assert [st.n_ext, st.n_ext_blocks, st.n_sa_lookups, st.n_lf_steps] + list(st.n_smem) == list(c)
    zb = np.load(os.path.join(G, "bsw_tasks.npz"))
    pairs = np.ascontiguousarray(zb["pairs"]).view(capi.SEQPAIR_DTYPE).reshape(-1)
    for w, key in ((100, "out_w100"), (200, "out_w200")):
        got = b.bsw(pairs, zb["ref"], zb["qer"], w)
        assert np.array_equal(np.stack([got[f] for f in OUT_FIELDS], axis=1), zb[key])
    # mate-rescue local SW
    zk, cases, qo, to = _ksw_case()
    for k, fl in enumerate(zk["flags"]):
        kp = np.zeros(len(cases), dtype=capi.SEQPAIR_DTYPE)
        kp["idr"], kp["idq"], kp["len1"], kp["len2"], kp["h0"] = to[:-1], qo[:-1], zk["tlen"], zk["qlen"], int(fl)
        assert np.array_equal(b.ksw_align(kp, zk["ref"], zk["qer"]), zk["out"][k])
    # EMF probe
    ze, tab, reads = _emf_case()
    e = capi.Emf(ix, table=tab)
    rl = ze["read_len"]
    perfect, code = b.emf_probe(e, ze["reads"], np.r_[0, np.cumsum(rl)].astype(np.int64))
    assert np.array_equal(code, ze["expect"][:, 0].astype(np.uint8))
    hit = (code == 3) | (code == 4)
    assert np.array_equal(perfect[hit], ze["expect"][hit, 1:].astype(np.uint32))
    e.close()
    b.close()
    ix.close()
