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


def _chain_case():
    z = np.load(os.path.join(G, "chain_toy.npz"))
    cum = np.r_[0, np.cumsum(z["read_len"])].astype(np.int64)
    chains = np.ascontiguousarray(z["chains"]).view(loader.CHAIN_DTYPE).reshape(-1)
    seeds = np.ascontiguousarray(z["seeds"]).view(loader.CHAIN_SEED_DTYPE).reshape(-1)
    regs = np.ascontiguousarray(z["regs"]).view(loader.ALNREG_DTYPE).reshape(-1)
    return z, cum, chains, seeds, regs


def _final_case(z):
    return np.ascontiguousarray(z["final"]).view(loader.ALNREG_DTYPE).reshape(-1), z["final_off"]


FINAL_F = ("rb", "re", "qb", "qe", "rid", "chain", "score", "truesc", "sub", "csub", "w", "seedcov", "seedlen0", "n_comp_is_alt",
           "frac_rep")


CHAIN_F = ("seqid", "n", "m", "first", "rid", "w_kept_alt", "frac_rep", "pos", "seed_off")
SEED_F = ("rbeg", "qbeg", "len", "score", "aln")
REG_F = ("rb", "re", "qb", "qe", "rid", "chain", "score", "truesc", "w", "seedcov", "seedlen0", "frac_rep")


def test_oracle_reproduces_chain_golden():
    z, cum, chains, seeds, regs = _chain_case()
    assert bool(z["klib_checked_against_reference"][0]), "regenerate with oracle/_ref present"
    zs, idx = _seed_case()
    o = loader.OracleFMI(idx)
    sm = o.collect_smem(z["reads"], cum)
    coord, off = o.sa_lookup(sm, 500)
    ch, sd, choff = loader.chain_seeds(sm, coord, off, cum, len(zs["genome"]))
    rg, roff, sd2 = loader.chain2aln(ch, sd, choff, z["reads"], cum, idx.ref_0123, len(zs["genome"]))
    assert np.array_equal(choff, z["chain_off"]) and np.array_equal(roff, z["reg_off"])
    for f in CHAIN_F:
        assert np.array_equal(ch[f], chains[f]), f
    for f in SEED_F:
        assert np.array_equal(sd2[f], seeds[f]), f
    for f in REG_F:
        assert np.array_equal(rg[f], regs[f]), f
    fin, fin_off = _final_case(z)
    gf, gfo = loader.regs_finish(rg, roff, z["reads"], cum, idx.ref_0123, len(zs["genome"]))
    assert np.array_equal(gfo, fin_off)
    for f in FINAL_F:
        assert np.array_equal(gf[f], fin[f]), f
    per_read = np.diff(choff)
    dup = sum(len(np.unique(ch["pos"][a:b])) != b - a for a, b in zip(choff[:-1], choff[1:]))
    assert per_read.max() >= 3 and dup > 0            # the fixture holds duplicate chain positions


@pytest.mark.gpu
def test_gpu_reproduces_chain_golden():
    from bwams import capi
    z, cum, chains, seeds, regs = _chain_case()
    zs, idx = _seed_case()
    ix = capi.Index.from_host(idx, 0)
    b = capi.Batch(ix, len(cum) + 64, int(cum[-1]) + 16384)
    b.seed_upload(z["reads"], cum)
    b.seed_run(capi.default_seed_opt(), with_sa=True)
    for extend_all in (1, 0):
        opt = capi.default_mem_opt()
        opt.extend_all = extend_all
        b.chain_run(opt)
        ch, sd, choff = b.chain_fetch()
        b.extend_run(opt)
        rg, roff, aln = b.extend_fetch()
        assert np.array_equal(choff, z["chain_off"]) and np.array_equal(roff, z["reg_off"])
        for f in CHAIN_F:
            assert np.array_equal(ch[f], chains[f]), f
        for f in SEED_F[:-1]:
            assert np.array_equal(sd[f], seeds[f]), f
        assert np.array_equal(aln, seeds["aln"])
        purged = (regs["qb"] == -1) & (regs["qe"] == -1)
        assert np.array_equal((rg["qb"] == -1) & (rg["qe"] == -1), purged)
        keep = slice(None) if extend_all else ~purged
        for f in REG_F:
            assert np.array_equal(rg[f][keep], regs[f][keep]), (extend_all, f)
        fin, fin_off = _final_case(z)
        assert b.dedup_run(opt) == len(fin)
        gf, gfo = b.dedup_fetch()
        assert np.array_equal(gfo, fin_off)
        for f in FINAL_F:
            assert np.array_equal(gf[f], fin[f]), (extend_all, f)
    b.close()
    ix.close()


PAIR_REG_F = ("rb", "re", "qb", "qe", "rid", "score", "truesc", "sub", "alt_sc", "csub", "sub_n", "w", "seedcov", "secondary",
              "secondary_all", "n_comp_is_alt", "hash")


def _pair_case():
    z = np.load(os.path.join(G, "pair_toy.npz"))
    cum = np.concatenate([[0], np.cumsum(z["read_len"])]).astype(np.int64)
    pes = np.ascontiguousarray(z["pes"]).view(loader.PESTAT_DTYPE).ravel()
    fin = np.ascontiguousarray(z["final"]).view(loader.ALNREG_DTYPE).ravel()
    out = np.ascontiguousarray(z["out"]).view(loader.ALNREG_DTYPE).ravel()
    pairs = np.ascontiguousarray(z["pairs"]).view(loader.PAIR_DTYPE).ravel()
    return z, cum, pes, fin, out, pairs


def test_oracle_reproduces_pair_golden():
    z, cum, pes, fin, out, pairs = _pair_case()
    zs, idx = _seed_case()
    l_pac = len(zs["genome"])
    assert np.array_equal(loader.pestat(fin, z["final_off"], l_pac), pes)
    got, got_off, got_pairs = loader.pair_pe(fin, z["final_off"], z["reads"], cum, idx.ref_0123, l_pac, pes, id_base=int(z["id_base"]))
    assert np.array_equal(got_off, z["out_off"]) and np.array_equal(got_pairs, pairs)
    for f in PAIR_REG_F:
        assert np.array_equal(got[f], out[f]), f
    assert pairs["n_matesw"].sum() > 10 and (pairs["score"] > 0).sum() > 80      # the fixture holds rescues and proper pairs


@pytest.mark.gpu
def test_gpu_reproduces_pair_golden():
    from bwams import capi
    z, cum, pes, fin, out, pairs = _pair_case()
    zs, idx = _seed_case()
    ix = capi.Index.from_host(idx, 0)
    b = capi.Batch(ix, len(cum) - 1, int(cum[-1]))
    opt = capi.default_mem_opt()
    b.seed_upload(z["reads"], cum)
    b.seed_run(capi.default_seed_opt(), with_sa=True)
    b.chain_run(opt)
    b.extend_run(opt)
    assert b.dedup_run(opt) == len(fin)
    assert np.array_equal(b.pestat(opt), pes)
    n, _ = b.pair_run(pes, opt, id_base=int(z["id_base"]))
    got, got_off, got_pairs = b.pair_fetch()
    assert n == len(out) and np.array_equal(got_off, z["out_off"])
    for f in ("score", "sub", "n_sub", "z", "n_pri", "n_matesw"):
        assert np.array_equal(got_pairs[f], pairs[f]), f
    for f in PAIR_REG_F:
        assert np.array_equal(got[f], out[f]), f
    b.close()
    ix.close()


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
