"""The paired-end tail up to the pairing decision (mate rescue, mem_mark_primary_se, mem_pair): the oracle's driver
logic on simulated FR pairs, checked through what the domain guarantees (a rescued mate lands where its fragment
says, proper pairs pair, orientation statistics select FR) and on constructed cases."""
import numpy as np

from bwams import simulate
from oracle import loader
from tests import util


def _chunk(n_pairs=400, seed=5, **kw):
    g, idx = util.toy(60000, seed=3)
    reads = simulate.make_read_pairs(g, n_pairs, seed=seed, **kw)
    return g, idx, reads, util.oracle_pe_pipeline(g, idx, reads)


def test_rescue_pairs_the_damaged_ends():
    g, idx, reads, c = _chunk()
    pes = c["pes"]
    assert pes["failed"].tolist() == [1, 0, 1, 1]                      # an FR library
    assert 300 < pes["avg"][1] < 500
    out, out_off, pairs = loader.pair_pe(c["regs"], c["reg_off"], c["enc"], c["cum"], c["ref"], c["l_pac"], pes)
    n0, n1 = np.diff(c["reg_off"]), np.diff(out_off)
    assert pairs["n_matesw"].sum() > 20 and (n1 > n0).sum() > 10          # rescue ran and added regions
    # without rescue fewer pairs are proper
    _, _, pairs_nr = loader.pair_pe(c["regs"], c["reg_off"], c["enc"], c["cum"], c["ref"], c["l_pac"], pes, no_rescue=True)
    assert (pairs["score"] > 0).sum() > (pairs_nr["score"] > 0).sum()
    assert pairs_nr["n_matesw"].sum() == 0
    # a proper pair: z picks regions whose distance lies within the FR bounds
    l_pac = c["l_pac"]
    ok = 0
    for p in np.flatnonzero(pairs["score"] > 0):
        a = out[out_off[2 * p] + pairs["z"][p, 0]]
        b = out[out_off[2 * p + 1] + pairs["z"][p, 1]]
        fa = a["rb"] if a["rb"] < l_pac else 2 * l_pac - 1 - a["rb"]
        fb = b["rb"] if b["rb"] < l_pac else 2 * l_pac - 1 - b["rb"]
        assert (a["rb"] >= l_pac) != (b["rb"] >= l_pac)                      # opposite strands
        assert pes["low"][1] <= abs(int(fa) - int(fb)) <= pes["high"][1] + 1
        assert pairs["score"][p] <= a["score"] + b["score"]
        ok += 1
    assert ok > 300
    # every list is what mem_mark_primary_se leaves: score-descending among non-ALT, hashes set, the first is primary
    for r in range(len(out_off) - 1):
        a = out[out_off[r]:out_off[r + 1]]
        if len(a):
            assert a["secondary"][0] == -1 and np.all(np.diff(a["score"]) <= 0) and np.all(a["hash"] != 0)
    assert np.array_equal(pairs["n_pri"].ravel(), n1)                      # no ALT sequences here


def test_mark_primary_se_constructed():
    opt = loader.default_mem_opt()
    regs = np.zeros(4, loader.ALNREG_DTYPE)
    # two hits covering the same query span (the weaker becomes secondary of the stronger), one elsewhere, one ALT
    regs["qb"], regs["qe"] = [0, 5, 100, 0], [100, 100, 150, 100]
    regs["score"] = [95, 100, 50, 97]
    regs["rb"] = [1000, 5000, 9000, 20000]
    regs["re"] = regs["rb"] + (regs["qe"] - regs["qb"])
    regs["n_comp_is_alt"] = [1, 1, 1, 1 | (1 << 30)]
    out, n_pri = loader.mark_primary_se(regs, 7, opt)
    assert n_pri == 3
    # non-ALT first in score order, the ALT hit last
    assert out["score"].tolist() == [100, 95, 50, 97]
    assert out["secondary"].tolist() == [-1, 0, -1, 0x7fffffff]
    assert out["sub"].tolist() == [95, 0, 0, 0]                          # reset, then set by the second pass (primary assembly only)
    assert out["alt_sc"].tolist() == [0, 0, 0, 0] and out["secondary_all"].tolist() == [-1, 0, -1, 0]
    # sub_n is not reset between the passes: the 95 counts in both (100 - 95 <= 7), the ALT 97 in neither
    # (a non-ALT hit does not count an ALT competitor)
    assert out["sub_n"].tolist() == [2, 0, 0, 0]


def test_reorder_primary5_constructed():
    """mem_reorder_primary5 (bwamem.cpp:2009-2031) on hand-made marked lists: the leftmost primary non-ALT hit scoring >= T comes
    first, references to the two swapped places follow, everything else stays."""
    regs = np.zeros(6, loader.ALNREG_DTYPE)
    #            primary  sec of 0  primary(left)  sec of 2  weak left   ALT left
    regs["qb"] = [80,      85,       0,             5,        0,          0]
    regs["qe"] = [150,     150,      70,            70,       20,         60]
    regs["score"] = [70,   60,       65,            50,       20,         90]
    regs["secondary"] = [-1, 0, -1, 2, -1, -1]
    regs["secondary_all"] = [-1, 0, -1, 2, -1, 0]
    regs["n_comp_is_alt"] = [1, 1, 1, 1, 1, 1 | (1 << 30)]
    regs["rb"] = np.arange(6) * 1000
    out = loader.reorder_primary5(regs, 30)
    assert out["rb"].tolist() == [2000, 1000, 0, 3000, 4000, 5000]       # places 0 and 2 swapped (the weak hit at qb 0 scores < T)
    assert out["secondary"].tolist() == [-1, 2, -1, 0, -1, -1]
    assert out["secondary_all"].tolist() == [-1, 2, -1, 0, -1, 2]
    assert np.array_equal(loader.reorder_primary5(out, 30), out)         # the leftmost is first now: nothing to do
    # T above every other hit: a single candidate, untouched; T = 0: the weak hit at qb 0 ties with place 2, the first of equals wins
    assert np.array_equal(loader.reorder_primary5(regs, 66), regs)
    assert loader.reorder_primary5(regs, 0)["rb"].tolist() == [2000, 1000, 0, 3000, 4000, 5000]
    one = regs[:2].copy()
    assert np.array_equal(loader.reorder_primary5(one, 30), one)
    assert len(loader.reorder_primary5(regs[:0], 30)) == 0


def test_pair_primary5_and_nopairing_in_the_driver():
    g, idx, reads, c = _chunk(200, seed=9, damaged_frac=0.0, discordant_frac=0.0)
    base = loader.pair_pe(c["regs"], c["reg_off"], c["enc"], c["cum"], c["ref"], c["l_pac"], c["pes"])
    nop = loader.pair_pe(c["regs"], c["reg_off"], c["enc"], c["cum"], c["ref"], c["l_pac"], c["pes"], no_pairing=True)
    assert np.array_equal(nop[0], base[0]) and np.array_equal(nop[2]["n_pri"], base[2]["n_pri"])
    assert (nop[2]["score"] == 0).all() and (nop[2]["z"] == -1).all() and (base[2]["score"] > 0).any()
    p5 = loader.pair_pe(c["regs"], c["reg_off"], c["enc"], c["cum"], c["ref"], c["l_pac"], c["pes"], primary5_T=30)
    want = base[0].copy()
    for r in range(len(base[1]) - 1):
        a, e = base[1][r], base[1][r + 1]
        want[a:e] = loader.reorder_primary5(base[0][a:e], 30)
    assert np.array_equal(p5[0], want) and np.array_equal(p5[1], base[1])


def test_pair_scores_follow_the_insert_size_model():
    g, idx, reads, c = _chunk(200, seed=9, damaged_frac=0.0, discordant_frac=0.0)
    pes = c["pes"].copy()
    out, out_off, pairs = loader.pair_pe(c["regs"], c["reg_off"], c["enc"], c["cum"], c["ref"], c["l_pac"], pes)
    good = pairs["score"] > 0
    assert good.mean() > 0.9
    # the pairing score is the sum of the two scores minus the insert-size penalty: recompute it
    import math
    for p in np.flatnonzero(good)[:100]:
        a = out[out_off[2 * p] + pairs["z"][p, 0]]
        b = out[out_off[2 * p + 1] + pairs["z"][p, 1]]
        l_pac = c["l_pac"]
        fa = int(a["rb"]) if a["rb"] < l_pac else 2 * l_pac - 1 - int(a["rb"])
        fb = int(b["rb"]) if b["rb"] < l_pac else 2 * l_pac - 1 - int(b["rb"])
        ns = (abs(fa - fb) - pes["avg"][1]) / pes["std"][1]
        q = int(int(a["score"]) + int(b["score"]) + .721 * math.log(2. * math.erfc(abs(ns) * math.sqrt(.5))) + .499)
        assert pairs["score"][p] == max(q, 0)


def test_ert_variant_of_mate_rescue():
    """useErt routes mate rescue through mem_matesw_batch_post_ert (list kept sorted by end, mem_dedup_patch after each
    alignment, one mem_sort_dedup_patch or score sort at the end).  Without a rescue the detour through the end-sorted
    order changes nothing (the final score sort is total once identical hits are gone); with one, the surviving regions
    are the same set in almost every pair (the reference's own runs differ in 1 pair of 5000, SURVEY §8c addendum)."""
    g, idx, reads, c = _chunk(400, seed=5)
    args = (c["regs"], c["reg_off"], c["enc"], c["cum"], c["ref"], c["l_pac"], c["pes"])
    out0, off0, pr0 = loader.pair_pe(*args)
    out1, off1, pr1 = loader.pair_pe(*args, use_ert=True)
    assert np.array_equal(pr0["n_matesw"], pr1["n_matesw"]) and pr1["n_matesw"].sum() > 20
    same = 0
    for p in range(len(pr0)):
        a0, a1 = out0[off0[2 * p]:off0[2 * p + 2]], out1[off1[2 * p]:off1[2 * p + 2]]
        if pr0["n_matesw"][p] == 0:
            assert np.array_equal(a0, a1) and np.array_equal(pr0[p], pr1[p]), p
        k0 = sorted(zip(a0["rb"].tolist(), a0["re"].tolist(), a0["qb"].tolist(), a0["qe"].tolist(), a0["score"].tolist()))
        k1 = sorted(zip(a1["rb"].tolist(), a1["re"].tolist(), a1["qb"].tolist(), a1["qe"].tolist(), a1["score"].tolist()))
        same += k0 == k1
    assert same >= len(pr0) - 4
    assert (pr1["score"] > 0).sum() >= (pr0["score"] > 0).sum() - 2


def test_hash_64_and_bns_depos_equal_the_reference_headers():
    """hash_64 (utils.h:117-128) and bns_depos (bntseq.h:88-91) are header-only inlines: the reference's own, compiled into
    oracle/_ref/libref_chain.so, against the restatements the pairing / chaining oracles use."""
    import ctypes as C
    R = loader.ref_chain_lib()
    if R is None:
        pytest.skip("oracle/_ref not built (reference tree absent)")
    L = loader.lib()
    R.ref_hash_64.restype = L.orc_hash_64.restype = C.c_uint64
    R.ref_hash_64.argtypes = L.orc_hash_64.argtypes = [C.c_uint64]
    R.ref_bns_depos.restype = L.orc_depos.restype = C.c_int64
    R.ref_bns_depos.argtypes = L.orc_depos.argtypes = [C.c_int64, C.c_int64, C.c_void_p]
    rng = np.random.default_rng(12)
    keys = list(rng.integers(0, 2 ** 63, size=5000, dtype=np.uint64)) + [0, 1, 2 ** 64 - 1, 2 ** 32, 7_000_000]
    for k in keys:
        assert R.ref_hash_64(int(k)) == L.orc_hash_64(int(k))
    l_pac = 3_209_286_105
    for pos in list(rng.integers(0, 2 * l_pac, size=3000)) + [0, l_pac - 1, l_pac, 2 * l_pac - 1]:
        a, b = C.c_int(-1), C.c_int(-1)
        assert R.ref_bns_depos(l_pac, int(pos), C.byref(a)) == L.orc_depos(l_pac, int(pos), C.byref(b)) and a.value == b.value
