"""Full-size run (BASELINE configs[1] shape: 1 M x 150 bp reads, one GPU) of the whole path, checked through
properties that do not need the oracle to process a million reads — and, because reads are independent, against
the oracle on a random sample of the very same reads."""
import numpy as np
import pytest

from bwams import capi, fmindex, simulate
from oracle import loader

pytestmark = pytest.mark.gpu

N_READS = 1_000_000
GENOME = 48_000_000


@pytest.fixture(scope="module")
def big():
    import torch
    capi.lib()
    g = simulate.make_genome(GENOME, seed=77)
    idx_dev = fmindex.build_fmindex(g, device="cuda:0", keep_ref=True)
    torch.cuda.synchronize()
    ix = capi.Index.from_device(idx_dev, 0)
    reads, _, _ = simulate.make_reads(g, N_READS, seed=99)
    enc, cum = simulate.flatten_reads(reads)
    host = fmindex.FMIndex(idx_dev.ref_seq_len, idx_dev.count, idx_dev.cp_occ.cpu().numpy().view(np.uint64),
                           idx_dev.sa_ms_byte.cpu().numpy(), idx_dev.sa_ls_word.cpu().numpy().view(np.uint32),
                           idx_dev.sentinel_index, np.concatenate([g, (3 - g[::-1]).astype(np.uint8)]))
    yield g, host, ix, reads, enc, cum
    ix.close()


def _run(b, opt):
    b.seed_run(capi.default_seed_opt(), with_sa=True)
    b.chain_run(opt)
    b.extend_run(opt)
    n = b.dedup_run(opt)
    fin, off = b.dedup_fetch()
    return n, fin, off


def test_million_reads_properties_and_sampled_parity(big):
    g, host, ix, reads, enc, cum = big
    b = capi.Batch(ix, N_READS, int(cum[-1]))
    b.seed_upload(enc, cum)
    opt = capi.default_mem_opt()
    n, fin, off = _run(b, opt)
    sm, coord, sa_off = b.seed_fetch()
    # --- properties at full size
    key = (sm["rid"].astype(np.int64) << 32) | (sm["m"].astype(np.int64) << 16) | sm["n"].astype(np.int64)
    assert np.all(np.diff(key) >= 0)                                         # mem_collect_smem's (rid, m, n) order
    assert off[0] == 0 and off[-1] == n == len(fin) and np.all(np.diff(off) >= 0)
    per = np.diff(off)
    assert 0.97 < (per > 0).mean() <= 1.0                                    # (almost) every read has a region
    assert np.all(fin["qe"] > fin["qb"]) and np.all(fin["re"] > fin["rb"]) and np.all(fin["qb"] >= 0) and np.all(fin["qe"] <= 150)
    rid_of = np.repeat(np.arange(N_READS), per)
    same = rid_of[1:] == rid_of[:-1]
    s0, s1 = fin["score"][:-1], fin["score"][1:]
    r0, r1 = fin["rb"][:-1], fin["rb"][1:]
    q0, q1 = fin["qb"][:-1], fin["qb"][1:]
    ordered = (s0 > s1) | ((s0 == s1) & ((r0 < r1) | ((r0 == r1) & (q0 < q1))))
    assert np.all(ordered[same])                                              # sorted by (score desc, rb, qb), no identical hits
    best = fin[off[:-1][per > 0]]
    assert (best["score"] >= 100).mean() > 0.9                                # simulated reads align nearly end to end
    # --- idempotence and the extend-everything form: same final regions, same checksum
    def checksum(a):
        x = np.zeros(len(a), np.uint64)
        for f in ("rb", "re", "qb", "qe", "score", "truesc", "w", "seedcov", "n_comp_is_alt", "rid"):
            x = x * np.uint64(1000003) + a[f].astype(np.int64).astype(np.uint64)
        return int(np.bitwise_xor.reduce(x)), int(x.sum(dtype=np.uint64))
    n2, fin2, off2 = _run(b, opt)
    assert n2 == n and np.array_equal(off2, off) and checksum(fin2) == checksum(fin)
    opt_all = capi.default_mem_opt()
    opt_all.extend_all = 1
    n3, fin3, off3 = _run(b, opt_all)
    assert n3 == n and np.array_equal(off3, off) and checksum(fin3) == checksum(fin)
    # --- the oracle on a random sample of the same reads (a read's result does not depend on the others)
    rng = np.random.default_rng(5)
    pick = np.sort(rng.choice(N_READS, size=2500, replace=False))
    sub_enc, sub_cum = simulate.flatten_reads(reads[pick])
    o = loader.OracleFMI(host)
    osm = o.collect_smem(sub_enc, sub_cum)
    ocoord, ooff = o.sa_lookup(osm)
    l_pac = len(g)
    ch, sd, choff = loader.chain_seeds(osm, ocoord, ooff, sub_cum, l_pac)
    regs, reg_off, _ = loader.chain2aln(ch, sd, choff, sub_enc, sub_cum, host.ref_0123, l_pac)
    wfin, wfin_off = loader.regs_finish(regs, reg_off, sub_enc, sub_cum, host.ref_0123, l_pac)
    for k, r in enumerate(pick):
        a, w = fin[off[r]:off[r + 1]], wfin[wfin_off[k]:wfin_off[k + 1]]
        assert len(a) == len(w), r
        for f in ("rb", "re", "qb", "qe", "rid", "score", "truesc", "w", "seedcov", "seedlen0", "n_comp_is_alt", "frac_rep"):
            assert np.array_equal(a[f], w[f]), (r, f)
    b.close()
