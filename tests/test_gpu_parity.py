"""GPU parity: the HIP path, called through the C-ABI, against the CPU oracle on the
same seeded inputs.  Integer work: every comparison is bit-exact."""
import numpy as np
import pytest

from bwams import capi, fmindex, simulate
from oracle import loader
from util import assert_pairs_equal, make_pairs, toy, toy_reads

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu_toy():
    capi.lib()
    g, idx = toy(200000, seed=13)
    ix = capi.Index.from_host(idx, 0)
    yield g, idx, ix
    ix.close()


def _seed_both(idx, ix, reads, opt=None, skip=None, oracle_counters=None):
    enc, cum = simulate.flatten_reads(reads)
    o = loader.OracleFMI(idx)
    oopt = loader.default_seed_opt()
    gopt = capi.default_seed_opt()
    if opt:
        for k, v in opt.items():
            setattr(oopt, k, v)
            setattr(gopt, k, v)
    want = o.collect_smem(enc, cum, oopt, skip=skip, counters=oracle_counters)
    wcoord, woff = o.sa_lookup(want, oopt.max_occ, counters=oracle_counters)
    b = capi.Batch(ix, max(len(cum) - 1, 1), max(int(cum[-1]), 1), max_smem=len(want) + 4096,
                   max_sa=len(wcoord) + 4096)
    got, coord, off = b.seed(enc, cum, gopt, skip=skip)
    st = b.stats()
    b.close()
    return want, wcoord, woff, got, coord, off, st


def test_smem_and_sa_match_oracle(gpu_toy):
    g, idx, ix = gpu_toy
    reads, _, _ = simulate.make_reads(g, 4000, seed=5)
    ctr = loader.Counters()
    want, wcoord, woff, got, coord, off, st = _seed_both(idx, ix, reads, oracle_counters=ctr)
    assert len(got) == len(want)
    for f in ("rid", "m", "n", "k", "l", "s"):
        assert np.array_equal(got[f], want[f]), f
    assert np.array_equal(off, woff)
    assert np.array_equal(coord, wcoord)
    # the kernel performs exactly the extension / LF events the reference algorithm does
    assert st.n_ext == ctr.n_ext and st.n_ext_blocks == ctr.n_ext_blocks
    assert st.n_lf_steps == ctr.n_lf_steps and st.n_sa_lookups == ctr.n_sa_lookups
    assert list(st.n_smem) == list(ctr.n_smem)


@pytest.mark.parametrize("env", [
    {"BWAMS_BWD_MIN_LIST": "1"},                                                       # every backward phase on the wave kernel
    {"BWAMS_BWD_MIN_LIST": "5", "BWAMS_BWD_COLS": "4", "BWAMS_BWD_LATE_LIST": "2"},     # at the forward end, or four columns in
    {"BWAMS_BWD_MIN_LIST": "200", "BWAMS_BWD_COLS": "2", "BWAMS_BWD_LATE_LIST": "1"},   # only ever two columns in
    {"BWAMS_BWD_MIN_LIST": "200", "BWAMS_BWD_COLS": "200", "BWAMS_BWD_LATE_LIST": "200",
     "BWAMS_BWD_DRY_MIN_LIST": "2", "BWAMS_BWD_DRY_COLS": "1", "BWAMS_BWD_DRY_LATE_LIST": "1"},                     # only while the launch drains
    {"BWAMS_BWD_MIN_LIST": "200", "BWAMS_BWD_COLS": "1", "BWAMS_BWD_LATE_LIST": "1"},   # one column in, one-entry lists too (the column ends in the iteration of the forward end)
    {"BWAMS_BWD_MIN_LIST": "40", "BWAMS_BWD_COLS": "1", "BWAMS_BWD_LATE_LIST": "33"},   # the whole-wavefront kernel only (lists beyond 32 entries)
    {"BWAMS_BWD_MIN_LIST": "0"},                                                       # never
])
def test_backward_phases_handed_to_the_wave_kernel(gpu_toy, monkeypatch, env):
    """Long interval lists leave the lane-per-read search (fmi_seed.hip, bwd_hand_over): a toy genome's lists never reach the
    production thresholds, so the thresholds come down to where every pivot (or every pivot a few columns in) takes that path.
    Repeat-rich reads and poly-A give lists beyond one batch of 64 entries; SMEMs, SA coordinates and the event counts stay exact."""
    g, idx, ix = gpu_toy
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    capi.debug_reload()                                      # the switches are read once: say that they changed
    reads, _, _ = simulate.make_reads(g, 3000, seed=21)
    reads = list(reads)
    rng = np.random.default_rng(4)
    for L in (150, 400, 1000):
        st = int(rng.integers(0, len(g) - 1100))
        reads.append(g[st:st + L].copy())
    reads.append(np.zeros(120, np.uint8))
    reads.append(np.tile(np.array([0, 1], np.uint8), 100))
    ctr = loader.Counters()
    want, wcoord, woff, got, coord, off, st = _seed_both(idx, ix, reads, {"min_seed_len": 12}, oracle_counters=ctr)
    assert len(got) == len(want)
    for f in ("rid", "m", "n", "k", "l", "s"):
        assert np.array_equal(got[f], want[f]), f
    assert np.array_equal(off, woff) and np.array_equal(coord, wcoord)
    assert st.n_ext == ctr.n_ext and st.n_ext_blocks == ctr.n_ext_blocks
    assert list(st.n_smem) == list(ctr.n_smem)


@pytest.mark.parametrize("opt", [
    {"min_seed_len": 10}, {"max_mem_intv": 0}, {"split_width": 50, "split_factor": 1.0},
    {"max_occ": 3}, {"min_seed_len": 30, "max_mem_intv": 5},
])
def test_seed_options(gpu_toy, opt):
    g, idx, ix = gpu_toy
    reads, _, _ = simulate.make_reads(g, 1500, seed=8)
    want, wcoord, woff, got, coord, off, _ = _seed_both(idx, ix, reads, opt)
    assert np.array_equal(got, want) or all(np.array_equal(got[f], want[f]) for f in ("rid", "m", "n", "k", "l", "s"))
    assert np.array_equal(off, woff) and np.array_equal(coord, wcoord)


def test_ragged_and_degenerate_reads(gpu_toy):
    g, idx, ix = gpu_toy
    rng = np.random.default_rng(3)
    reads = []
    for L in [0, 1, 5, 18, 19, 20, 37, 151, 250, 400, 1000]:
        st = int(rng.integers(0, len(g) - 1100))
        r = g[st:st + L].copy()
        reads.append(r)
    reads.append(np.full(60, 4, np.uint8))                    # all N
    r = g[5000:5150].copy(); r[[0, 30, 31, 149]] = 4; reads.append(r)   # N at the ends and inside
    r = g[7000:7150].copy(); r[75] = (r[75] + 1) & 3; reads.append(r)   # one mismatch
    reads.append(simulate.revcomp(g[9000:9150]))              # reverse strand
    reads.append(np.zeros(100, np.uint8))                     # poly-A: huge intervals
    want, wcoord, woff, got, coord, off, _ = _seed_both(idx, ix, reads)
    for f in ("rid", "m", "n", "k", "l", "s"):
        assert np.array_equal(got[f], want[f]), f
    assert np.array_equal(off, woff) and np.array_equal(coord, wcoord)


def test_skip_flags_and_empty_batch(gpu_toy):
    g, idx, ix = gpu_toy
    reads, _, _ = simulate.make_reads(g, 600, seed=21)
    skip = (np.arange(600) % 3 == 0).astype(np.uint8)
    want, wcoord, woff, got, coord, off, _ = _seed_both(idx, ix, reads, skip=skip)
    for f in ("rid", "m", "n", "k", "l", "s"):
        assert np.array_equal(got[f], want[f]), f
    assert np.array_equal(coord, wcoord)
    assert not np.any(skip[got["rid"]])
    b = capi.Batch(ix, 8, 1200)
    sm, coord, off = b.seed(np.zeros(0, np.uint8), np.zeros(1, np.int64))
    assert len(sm) == 0 and len(coord) == 0 and list(off) == [0]
    b.close()


def test_capacity_error_is_reported_not_truncated(gpu_toy):
    g, idx, ix = gpu_toy
    reads, _, _ = simulate.make_reads(g, 500, seed=2)
    enc, cum = simulate.flatten_reads(reads)
    b = capi.Batch(ix, 500, int(cum[-1]), max_smem=100, max_sa=100)
    with pytest.raises(capi.BwamsError) as e:
        b.seed_onecall(enc, cum)
    assert e.value.code == -4
    b.close()


def test_resident_buffers_grow_on_demand(gpu_toy):
    """The resident path sizes its SMEM / SA buffers from an overflowing pass and runs the stage again; the
    results (and the event counters) are those of a batch that was large enough from the start."""
    g, idx, ix = gpu_toy
    reads, _, _ = simulate.make_reads(g, 2000, seed=12)
    enc, cum = simulate.flatten_reads(reads)
    o = loader.OracleFMI(idx)
    ctr = loader.Counters()
    want = o.collect_smem(enc, cum, counters=ctr)
    wcoord, woff = o.sa_lookup(want, 500, counters=ctr)
    b = capi.Batch(ix, len(reads), int(cum[-1]), max_smem=64, max_sa=64)
    got, coord, off = b.seed(enc, cum)
    assert len(got) == len(want) > 64
    for f in ("rid", "m", "n", "k", "l", "s"):
        assert np.array_equal(got[f], want[f]), f
    assert np.array_equal(off, woff) and np.array_equal(coord, wcoord)
    st = b.stats()
    assert st.n_sa_lookups == ctr.n_sa_lookups and st.n_lf_steps == ctr.n_lf_steps and list(st.n_smem) == list(ctr.n_smem)
    # and the stages behind it run on the grown buffers
    nc, ns = b.chain_run()
    wch, wsd, wchoff = loader.chain_seeds(want, wcoord, woff, cum, len(g))
    assert nc == len(wch) and ns == len(wsd)
    b.close()


@pytest.mark.parametrize("mode", ["0", "1", "2"])
def test_round3_behind_beside_or_from_the_start(gpu_toy, monkeypatch, mode):
    """BWAMS_SEED_R3_BESIDE: round 3 behind round 2 (0), beside it (1), or from the start of the stage with a pool and counters of its
    own, appended behind round 2's records (2): schedules, not algorithms — the same SMEMs, coordinates and counts."""
    g, idx, ix = gpu_toy
    monkeypatch.setenv("BWAMS_SEED_R3_BESIDE", mode)
    capi.debug_reload()
    reads, _, _ = simulate.make_reads(g, 4000, seed=77)
    reads = list(reads) + [np.zeros(120, np.uint8), np.tile(np.array([0, 1], np.uint8), 100), np.full(60, 4, np.uint8)]
    for kw in ({}, {"min_seed_len": 12, "max_mem_intv": 8}, {"max_mem_intv": 0}):
        ctr = loader.Counters()
        want, wcoord, woff, got, coord, off, st = _seed_both(idx, ix, reads, kw, oracle_counters=ctr)
        assert len(got) == len(want)
        for f in ("rid", "m", "n", "k", "l", "s"):
            assert np.array_equal(got[f], want[f]), (kw, f)
        assert np.array_equal(off, woff) and np.array_equal(coord, wcoord)
        assert st.n_ext == ctr.n_ext and st.n_ext_blocks == ctr.n_ext_blocks
        assert list(st.n_smem) == list(ctr.n_smem)


def test_pool_with_no_room_to_spare_and_every_emitting_launch(gpu_toy, monkeypatch):
    """the SMEM pool is handed out in 64-slot chunks by SEVEN emitting launches (three searches, and the two kernels
    behind rounds 1 and 2 for the backward phases that left their lanes); a batch whose max_smem is the chunk's exact SMEM count
    must still succeed — the slack covers every launch's partly filled chunks — and one slot less must grow and re-run."""
    g, idx, ix = gpu_toy
    monkeypatch.setenv("BWAMS_BWD_MIN_LIST", "1")
    capi.debug_reload()
    reads, _, _ = simulate.make_reads(g, 2500, seed=31)
    enc, cum = simulate.flatten_reads(reads)
    o = loader.OracleFMI(idx)
    want = o.collect_smem(enc, cum)
    wcoord, woff = o.sa_lookup(want, 500)
    for room in (0, -1):
        b = capi.Batch(ix, len(reads), int(cum[-1]), max_smem=len(want) + room, max_sa=len(wcoord) + 8)
        got, coord, off = b.seed(enc, cum)
        for f in ("rid", "m", "n", "k", "l", "s"):
            assert np.array_equal(got[f], want[f]), (room, f)
        assert np.array_equal(off, woff) and np.array_equal(coord, wcoord)
        b.close()


def test_index_file_path(gpu_toy, tmp_path):
    g, idx, ix = gpu_toy
    fmindex.write_index(str(tmp_path / "t"), idx)
    ix2 = capi.Index.open(str(tmp_path / "t"), 0)
    assert ix2.nbytes == ix.nbytes
    reads, _, _ = simulate.make_reads(g, 300, seed=4)
    want, wcoord, woff, got, coord, off, _ = _seed_both(idx, ix2, reads)
    assert np.array_equal(got["k"], want["k"]) and np.array_equal(coord, wcoord)
    # the one-call entry point (caller buffers of capacity size) returns the same
    enc, cum = simulate.flatten_reads(reads)
    b = capi.Batch(ix2, len(reads), int(cum[-1]))
    sm1, c1, o1 = b.seed_onecall(enc, cum)
    assert np.array_equal(sm1["k"], want["k"]) and np.array_equal(c1, wcoord) and np.array_equal(o1, woff)
    b.close()
    ix2.close()


@pytest.mark.parametrize("w", [100, 200, 5])
@pytest.mark.parametrize("end_bonus,zdrop", [(5, 100), (0, 0), (5, 15)])
def test_bsw_matches_oracle(gpu_toy, w, end_bonus, zdrop):
    _, _, ix = gpu_toy
    pairs, ref, qer = make_pairs(3000, seed=w * 7 + end_bonus + zdrop)
    oopt = loader.default_sw_opt(end_bonus); oopt.zdrop = zdrop
    gopt = capi.default_sw_opt(end_bonus); gopt.zdrop = zdrop
    want, cells = loader.bsw_pairs(pairs, ref, qer, w, oopt)
    b = capi.Batch(ix, 8, 1200)
    got = b.bsw(pairs, ref, qer, w, gopt)
    st = b.stats()
    b.close()
    assert_pairs_equal(got, want, f"w={w}")
    assert st.bsw_cells == cells
    # inputs are untouched
    for f in ("idr", "idq", "id", "len1", "len2", "h0", "seqid", "regid"):
        assert np.array_equal(got[f], pairs[f])


def test_bsw_long_queries_and_edges(gpu_toy):
    _, _, ix = gpu_toy
    pairs, ref, qer = make_pairs(400, seed=77, max_q=700, max_extra_t=300, h0_max=600)
    pairs["len1"][::17] = 0                                   # empty targets
    want, _ = loader.bsw_pairs(pairs, ref, qer, 100)
    b = capi.Batch(ix, 8, 1200)
    got = b.bsw(pairs, ref, qer, 100)
    assert_pairs_equal(got, want, "long")
    # no pairs at all
    empty = b.bsw(pairs[:0], ref, qer, 100)
    assert len(empty) == 0
    b.close()


def test_bsw_scoring_matrix_variants(gpu_toy):
    _, _, ix = gpu_toy
    pairs, ref, qer = make_pairs(800, seed=5)
    for (a, bb, od, ed, oi, ei) in [(2, 3, 5, 2, 4, 1), (1, 1, 1, 1, 1, 1), (3, 7, 10, 3, 8, 2)]:
        oopt = loader.default_sw_opt(5, a, bb); gopt = capi.default_sw_opt(5, a, bb)
        for o in (oopt, gopt):
            o.o_del, o.e_del, o.o_ins, o.e_ins = od, ed, oi, ei
        want, _ = loader.bsw_pairs(pairs, ref, qer, 100, oopt)
        b = capi.Batch(ix, 8, 1200)
        got = b.bsw(pairs, ref, qer, 100, gopt)
        b.close()
        assert_pairs_equal(got, want, f"mat {a},{bb}")


def _local_tasks(cases, xtra):
    from oracle.loader import SEQPAIR_DTYPE
    pairs = np.zeros(len(cases), dtype=SEQPAIR_DTYPE)
    ro = qo = 0
    refs, qers = [], []
    for i, (q, t) in enumerate(cases):
        pairs[i]["idr"], pairs[i]["idq"], pairs[i]["id"] = ro, qo, i
        pairs[i]["len1"], pairs[i]["len2"], pairs[i]["h0"] = len(t), len(q), xtra
        pairs[i]["regid"] = i
        refs.append(t); qers.append(q)
        ro += len(t); qo += len(q)
    return pairs, np.concatenate(refs), np.concatenate(qers)


@pytest.mark.parametrize("flags", [
    loader.KSW_XSUBO | loader.KSW_XSTART | loader.KSW_XBYTE | 19,     # mem_matesw, 150-bp mates
    loader.KSW_XSUBO | loader.KSW_XSTART | 19,                         # 16-bit kernel
    loader.KSW_XBYTE, 0, loader.KSW_XSUBO | 40,
])
def test_ksw_local_matches_oracle(gpu_toy, flags):
    from util import make_local_cases
    _, _, ix = gpu_toy
    cases = make_local_cases(400, seed=flags % 977)
    pairs, ref, qer = _local_tasks(cases, flags)
    ref0, qer0 = ref.copy(), qer.copy()
    b = capi.Batch(ix, 8, 1200)
    got = b.ksw_align(pairs, ref, qer)
    b.close()
    want = np.array([loader.ksw_align2(q, t, flags) for q, t in cases], dtype=np.int32)
    bad = np.flatnonzero((got != want).any(axis=1))
    assert bad.size == 0, (bad[:5], got[bad[:3]], want[bad[:3]])
    assert np.array_equal(ref, ref0) and np.array_equal(qer, qer0)        # inputs are not reversed in place


def test_ksw_local_long_queries_scoring_and_limits(gpu_toy):
    from util import make_local_cases
    _, _, ix = gpu_toy
    flags = loader.KSW_XSUBO | loader.KSW_XSTART | 25
    cases = make_local_cases(120, seed=5, qmax=500, tmax=2000)
    cases.append((np.random.default_rng(1).integers(0, 4, 300, dtype=np.uint8),) * 2)   # byte-kernel overflow -> 255
    oopt = loader.default_sw_opt(5, 2, 3); gopt = capi.default_sw_opt(5, 2, 3)
    for o in (oopt, gopt):
        o.o_del, o.e_del, o.o_ins, o.e_ins = 5, 2, 4, 1
    for fl in (flags, flags | loader.KSW_XBYTE):
        pairs, ref, qer = _local_tasks(cases, fl)
        b = capi.Batch(ix, 8, 1200)
        got = b.ksw_align(pairs, ref, qer, gopt)
        want = np.array([loader.ksw_align2(q, t, fl, oopt) for q, t in cases], dtype=np.int32)
        assert np.array_equal(got, want), np.flatnonzero((got != want).any(axis=1))[:5]
        b.close()
    # unsupported scoring (insertion+deletion cheaper than a mismatch) is refused, not approximated
    bad = capi.default_sw_opt(5, 1, 30)
    b = capi.Batch(ix, 8, 1200)
    with pytest.raises(capi.BwamsError) as e:
        b.ksw_align(pairs[:2], ref, qer, bad)
    assert e.value.code == -6
    assert len(b.ksw_align(pairs[:0], ref, qer)) == 0
    b.close()
    # a long rescue window (15 kb target: the row-maxima lists need more than the default 64 KB of dynamic LDS) and one
    # beyond the supported length (refused, not truncated)
    rng = np.random.default_rng(77)
    t_long = rng.integers(0, 4, 15000, dtype=np.uint8)
    q_long = t_long[9000:9150].copy()
    q_long[[10, 70]] ^= 1
    lp, lref, lqer = _local_tasks([(q_long, t_long), (q_long[:100], t_long[:400])], flags)
    b = capi.Batch(ix, 8, 1200)
    got = b.ksw_align(lp, lref, lqer)
    want = np.array([loader.ksw_align2(q_long, t_long, flags), loader.ksw_align2(q_long[:100], t_long[:400], flags)], dtype=np.int32)
    assert np.array_equal(got, want) and got[0, 0] >= 140
    too_long = rng.integers(0, 4, 20001, dtype=np.uint8)
    tp, tref, tqer = _local_tasks([(q_long, too_long)], flags)
    with pytest.raises(capi.BwamsError) as e:
        b.ksw_align(tp, tref, tqer)
    assert e.value.code == -6
    b.close()


def test_fma_tables_and_seeding_with_fma(gpu_toy):
    """FMA (all_smem / last_smem): device-built tables == oracle-built tables; seeding with the tables
    == the oracle with the same tables (N reads included: the with_N quirk is reproduced); and at the
    reference's depths (11 / 13) the tables are pure accelerators on N-free reads."""
    g, idx, _ = gpu_toy
    ix = capi.Index.from_host(idx, 0)
    o = loader.OracleFMI(idx)
    want_all, want_last = o.build_fma(7, 8)
    ix.build_fma(7, 8)
    got_all, got_last = ix.fetch_fma()
    assert np.array_equal(got_last, want_last)
    assert np.array_equal(got_all, want_all)
    reads, _, _ = simulate.make_reads(g, 2500, seed=41)
    reads[::50, 3] = 4                                  # N inside the first table window of some reads
    enc, cum = simulate.flatten_reads(reads)
    for msl in (19, 5):                                 # 5: seeds are emitted at table time (no-break quirk)
        oopt = loader.default_seed_opt(); gopt = capi.default_seed_opt()
        oopt.min_seed_len = gopt.min_seed_len = msl
        ctr = loader.Counters()
        want = o.collect_smem(enc, cum, oopt, counters=ctr)
        b = capi.Batch(ix, len(reads), int(cum[-1]), max_smem=len(want) + 4096)
        got, _, _ = b.seed(enc, cum, gopt, with_sa=False)
        st = b.stats()
        b.close()
        for f in ("rid", "m", "n", "k", "l", "s"):
            assert np.array_equal(got[f], want[f]), (msl, f)
        assert st.n_ext == ctr.n_ext and list(st.n_smem) == list(ctr.n_smem)
    # tables uploaded from the host (the file path of bwams_index_open) behave the same
    ix.set_fma(want_all, 7, want_last, 8)
    b = capi.Batch(ix, len(reads), int(cum[-1]))
    got2, _, _ = b.seed(enc, cum, with_sa=False)
    o2 = loader.default_seed_opt()
    assert np.array_equal(got2["k"], o.collect_smem(enc, cum, o2)["k"])
    # reference depths: built on the device, identical SMEMs with and without on N-free reads
    clean = reads[(reads < 4).all(axis=1)]
    ence, cume = simulate.flatten_reads(clean)
    ix.build_fma(11, 13)
    with_fma, _, _ = b.seed(ence, cume, with_sa=False)
    n_fma = b.stats().n_ext
    ix.set_fma(None, 0, None, 0)
    without, _, _ = b.seed(ence, cume, with_sa=False)
    n_plain = b.stats().n_ext
    assert np.array_equal(with_fma, without) and n_fma < n_plain
    o.drop_fma()
    plain = o.collect_smem(ence, cume)
    assert np.array_equal(without["k"], plain["k"]) and np.array_equal(without["n"], plain["n"])
    b.close()
    ix.close()


@pytest.mark.parametrize("L", [150, 50])
def test_emf_probe_matches_oracle(gpu_toy, L, tmp_path):
    """Exact-match filter: the HIP probe == the restated find_perfect_match_entry on the same table,
    for reads of exactly L bases and longer ones (tail verification + multi-location lists), both
    strands, mismatches, N; and the resulting skip flags drive seeding as in the reference."""
    from bwams import emf
    g0, idx0, _ = gpu_toy
    g = g0[:60000].copy()
    g[5000:5400] = g[1000:1400]                              # exact repeats -> multi-location entries
    g[9000:9400] = (3 - g[1000:1400][::-1])                  # and a reverse-complement copy
    idx = fmindex.build_fmindex(g)
    ix = capi.Index.from_host(idx, 0)
    tab = emf.build_emf(g, L)
    emf.write_emf(str(tmp_path / "g.perfect"), tab)
    e = capi.Emf(ix, path=str(tmp_path / "g.perfect"))
    o = loader.OracleEMF(tab, idx.ref_0123)
    rng = np.random.default_rng(L)
    reads = []
    for it in range(3000):
        ln = L if it % 4 else int(rng.integers(L + 1, L + 60))
        st = int(rng.integers(0, len(g) - ln)) if it % 7 else int(rng.integers(1000, 1400 - min(ln, 350)) if ln < 350 else 0)
        rd = g[st: st + ln].copy()
        k = it % 6
        if k == 1:
            rd = (3 - rd[::-1]).astype(np.uint8)
        elif k == 2:
            rd[rng.integers(0, ln)] ^= 1
        elif k == 3 and it % 12 == 3:
            rd[rng.integers(0, ln)] = 4
        elif k == 4:
            rd[rng.integers(L - 1, ln)] ^= 2                  # breaks the tail (or the last seed base)
        reads.append(rd)
    reads.append(g[0:L - 1].copy())                           # shorter than the table's L -> code 0
    enc, cum = simulate.flatten_reads(reads)
    b = capi.Batch(ix, len(reads), int(cum[-1]))
    perfect, code = b.emf_probe(e, enc, cum)
    want = o.probe_many(reads)
    assert np.array_equal(code, want[:, 0].astype(np.uint8)), np.flatnonzero(code != want[:, 0])[:5]
    hit = (code == 3) | (code == 4)
    assert np.array_equal(perfect[hit, 0], want[hit, 1].astype(np.uint32))
    assert np.array_equal(perfect[hit, 1], want[hit, 2].astype(np.uint32))
    assert hit.sum() > 800 and (perfect[hit, 0] >> 2).astype(bool).sum() > 20 and set(code) >= {0, 1, 2, 3, 4}
    # the hits are exactly what seeding skips (bwamem.cpp:674-689)
    skip = hit.astype(np.uint8)
    sm, _, _ = b.seed(enc, cum, skip=skip, with_sa=False)
    assert not np.any(skip[sm["rid"]])
    b.close(); e.close(); ix.close()


def test_emf_resident_run_with_gpu_built_table(gpu_toy):
    """Table built on the GPU (torch builder) and adopted without a host copy; the resident probe sets the
    skip flags that the following seed run honours; results == the oracle on the same table."""
    import torch
    from bwams import emf
    g0, _, _ = gpu_toy
    g = g0[:80000].copy()
    idx = fmindex.build_fmindex(g)
    ix = capi.Index.from_host(idx, 0)
    t = emf.build_emf_torch(g, 150, "cuda:0")
    e = capi.Emf(ix, device_table=t)
    host_tab = emf.EmfTable(t.seed_len, t.seq_len, t.loc_table.cpu().numpy().view(np.uint32),
                            t.seed_table.cpu().numpy().view(np.uint32), t.num_seed_used, t.num_seed_key)
    o = loader.OracleEMF(host_tab, idx.ref_0123)
    reads, _, _ = simulate.make_reads(g, 4000, seed=51)
    enc, cum = simulate.flatten_reads(reads)
    b = capi.Batch(ix, len(reads), int(cum[-1]))
    b.seed_upload(enc, cum)
    b.emf_run(e)
    perfect, code = b.emf_fetch(len(reads))
    want = o.probe_many(list(reads))
    assert np.array_equal(code, want[:, 0].astype(np.uint8))
    hit = (code == 3) | (code == 4)
    assert np.array_equal(perfect[hit], want[hit, 1:].astype(np.uint32)) and 0.2 < hit.mean() < 0.8
    b.seed_run(with_sa=True)
    sm, coord, off = b.seed_fetch()
    assert not np.any(hit[sm["rid"]])
    oo = loader.OracleFMI(idx)
    want_sm = oo.collect_smem(enc, cum, skip=hit.astype(np.uint8))
    assert np.array_equal(sm["k"], want_sm["k"]) and np.array_equal(sm["rid"], want_sm["rid"])
    st = b.stats()
    assert st.emf_nodes >= hit.sum() and st.emf_cmp_bytes == st.emf_nodes * 150
    b.close(); e.close(); ix.close()


@pytest.mark.parametrize("L", [150, 64])
def test_emf_regions_match_oracle(gpu_toy, L, tmp_path):
    """mem_perfect2reg: every exact location of an EMF-resolved read (multi-location lists on both strands, tails
    verified for reads longer than L, overlapping locations of a tandem repeat collapsed) as a full-length region."""
    from bwams import emf
    g0, idx0, _ = gpu_toy
    g = g0[:60000].copy()
    g[5000:5400] = g[1000:1400]                              # exact repeats -> multi-location entries
    g[9000:9400] = (3 - g[1000:1400][::-1])                  # a reverse-complement copy
    g[30000:30400] = g[1000:1400]
    g[20000:20700] = np.tile(np.array([0, 1, 0, 2, 1], np.uint8), 140)      # tandem repeat: locations 5 bases apart
    idx = fmindex.build_fmindex(g)
    ix = capi.Index.from_host(idx, 0)
    contigs = np.zeros(3, capi.CONTIG_DTYPE)
    contigs["offset"], contigs["len"], contigs["is_alt"] = [0, 8000, 25000], [8000, 17000, len(g) - 25000], [0, 0, 1]
    ix.set_contigs(contigs)
    tab = emf.build_emf(g, L)
    e = capi.Emf(ix, table=tab)
    o = loader.OracleEMF(tab, idx.ref_0123)
    rng = np.random.default_rng(L + 1)
    reads = []
    for it in range(1500):
        ln = L if it % 3 else int(rng.integers(L + 1, L + 50))
        where = it % 5
        if where == 0:
            st = int(rng.integers(1000, 1400 - ln)) if ln < 400 else 1000
        elif where == 1:
            st = int(rng.integers(20000, 20700 - ln))
        else:
            st = int(rng.integers(0, len(g) - ln))
        rd = g[st:st + ln].copy()
        if it % 2:
            rd = simulate.revcomp(rd)
        if it % 11 == 0:
            rd[rng.integers(0, ln)] ^= 1
        reads.append(rd)
    enc, cum = simulate.flatten_reads(reads)
    b = capi.Batch(ix, len(reads), int(cum[-1]))
    b.seed_upload(enc, cum)
    b.emf_run(e)
    perfect, code = b.emf_fetch(len(reads))
    regs, off, rev = b.emf_regs(e)
    n_multi = n_collapsed = 0
    for r, rd in enumerate(reads):
        got = regs[off[r]:off[r + 1]]
        if code[r] not in (3, 4):
            assert len(got) == 0
            continue
        want, wrev = o.perfect2reg(rd, int(perfect[r, 0]), int(perfect[r, 1]), len(g), contigs=contigs)
        assert len(got) == len(want) and rev[r] == wrev, r
        for f in ("rb", "re", "qb", "qe", "rid", "score", "truesc", "w", "seedlen0", "n_comp_is_alt", "sub", "csub", "seedcov"):
            assert np.array_equal(got[f], want[f]), (r, f)
        n_multi += len(want) > 1
        if perfect[r, 0] >> 2:
            first = int(tab.loc_table[perfect[r, 0] >> 2])
            n_loc = 1 + ((first >> 16) & 0xffff) + (first & 0xffff) if not first & 0x80000000 else None
            if n_loc is not None and len(want) < n_loc:
                n_collapsed += 1
        # every region is an exact occurrence of the read — except that for a reverse-strand hit of a read longer than
        # L the reference places the region from the seed's location, not from the read's (mem_perfect2reg uses
        # mem_aln_perfect_t::loc, which init_mem_aln_perfect leaves unadjusted): restated as is
        ref = idx.ref_0123
        for a in want:
            if len(rd) == L or a["rb"] < len(g):
                assert np.array_equal(ref[a["rb"]:a["re"]], rd)
    assert n_multi > 100 and n_collapsed > 20
    b.close(); e.close()
    c = np.zeros(1, capi.CONTIG_DTYPE)
    c["len"] = len(g)
    ix.close()
