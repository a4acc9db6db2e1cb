"""N>1 path on CPU: world_size-2 gloo processes shard a chunk, each 'seeds' its shard with the
CPU oracle standing in for the GPU stage (the product kernels need a GPU), and the gathered
result must equal the single-process result in read order."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from bwams import shard, simulate
from util import toy, toy_reads


def test_shard_bounds_cover_and_balance():
    for n, w, m in [(10, 3, 1), (1_000_000, 8, 1), (7, 8, 1), (11, 4, 2), (0, 2, 1)]:
        b = shard.shard_bounds(n, w, m)
        assert b[0] == 0 and b[-1] == n and np.all(np.diff(b) >= 0)
        sizes = np.diff(b)
        assert sizes.max() - sizes.min() <= 2 * m - 1 or n < w * m
        if m > 1:
            assert all(s % m == 0 for s in sizes[:-1] if s)


def _worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "bwa-mem-scale_amd"), os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from oracle import loader
    g, idx = toy()
    reads, _, _ = toy_reads()
    enc, cum = simulate.flatten_reads(reads)
    e, c, first = shard.shard_reads(enc, cum, rank, world)
    local = loader.OracleFMI(idx).collect_smem(e, c)
    allsm = shard.gather_smems(local, first, dist)
    # the rest of the path on the shard: SA lookup, chaining, chain-to-alignment; regions gathered in read order
    o = loader.OracleFMI(idx)
    coord, off = o.sa_lookup(local)
    ch, sd, choff = loader.chain_seeds(local, coord, off, c, len(g))
    ref = np.concatenate([g, (3 - g[::-1]).astype(np.uint8)])
    regs, reg_off, _ = loader.chain2aln(ch, sd, choff, e, c, ref, len(g))
    allregs, all_off = shard.gather_regions(regs, reg_off, len(ch), dist)
    if rank == 0:
        np.save(os.path.join(out_dir, "gathered.npy"), allsm)
        np.save(os.path.join(out_dir, "regs.npy"), allregs)
        np.save(os.path.join(out_dir, "reg_off.npy"), all_off)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_gather_equals_single_process(tmp_path):
    from oracle import loader
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = np.load(tmp_path / "gathered.npy")
    g, idx = toy()
    reads, _, _ = toy_reads()
    enc, cum = simulate.flatten_reads(reads)
    want = loader.OracleFMI(idx).collect_smem(enc, cum)
    assert len(got) == len(want)
    for f in ("rid", "m", "n", "k", "l", "s"):
        assert np.array_equal(got[f], want[f]), f
    o = loader.OracleFMI(idx)
    coord, off = o.sa_lookup(want)
    ch, sd, choff = loader.chain_seeds(want, coord, off, cum, len(g))
    ref = np.concatenate([g, (3 - g[::-1]).astype(np.uint8)])
    wregs, wreg_off, _ = loader.chain2aln(ch, sd, choff, enc, cum, ref, len(g))
    gregs, goff = np.load(tmp_path / "regs.npy"), np.load(tmp_path / "reg_off.npy")
    assert np.array_equal(goff, wreg_off) and len(gregs) == len(wregs) > 0
    for f in ("rb", "re", "qb", "qe", "rid", "chain", "score", "truesc", "w", "seedcov", "seedlen0", "frac_rep"):
        assert np.array_equal(gregs[f], wregs[f]), f


def _pe_chunk():
    g, idx = toy(40000, seed=5)
    reads = simulate.make_read_pairs(g, 160, seed=21, damaged_frac=0.25, discordant_frac=0.1)
    return g, idx, reads


def _pe_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "bwa-mem-scale_amd"), os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from oracle import loader
    from util import oracle_pe_pipeline
    g, idx, reads = _pe_chunk()
    enc, cum = simulate.flatten_reads(reads)
    # shards hold whole pairs; the CPU oracle stands in for the GPU stages on each shard
    bounds = shard.shard_bounds(len(reads), world, 2)
    mine = reads[bounds[rank]:bounds[rank + 1]]
    c = oracle_pe_pipeline(g, idx, mine)
    # the exchange step: insert-size statistics of the WHOLE chunk from every shard's keys
    pes = shard.pestat_sharded(loader.pestat_keys(c["regs"], c["reg_off"], c["l_pac"]), dist)
    out, off, prs = loader.pair_pe(c["regs"], c["reg_off"], c["enc"], c["cum"], c["ref"], c["l_pac"], pes,
                                   id_base=int(bounds[rank]) // 2)
    allregs, alloff, allprs = shard.gather_pairs(out, off, prs, dist)
    if rank == 0:
        np.save(os.path.join(out_dir, "pes.npy"), pes)
        np.save(os.path.join(out_dir, "pe_regs.npy"), allregs)
        np.save(os.path.join(out_dir, "pe_off.npy"), alloff)
        np.save(os.path.join(out_dir, "pe_pairs.npy"), allprs)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_paired_end_equals_single_process(tmp_path):
    """Pairs sharded over two ranks: the all-gathered insert-size keys give the whole chunk's statistics, and the
    gathered regions / pairing decisions equal the single-process ones."""
    from oracle import loader
    from util import oracle_pe_pipeline
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_pe_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    g, idx, reads = _pe_chunk()
    c = oracle_pe_pipeline(g, idx, reads)
    assert np.array_equal(np.load(tmp_path / "pes.npy"), c["pes"])
    # and the host-only half of the product's mem_pestat agrees with the oracle on the same keys
    from bwams import capi
    assert np.array_equal(capi.pestat_from_keys(loader.pestat_keys(c["regs"], c["reg_off"], c["l_pac"])), c["pes"])
    want, want_off, want_prs = loader.pair_pe(c["regs"], c["reg_off"], c["enc"], c["cum"], c["ref"], c["l_pac"], c["pes"])
    got, got_off, got_prs = np.load(tmp_path / "pe_regs.npy"), np.load(tmp_path / "pe_off.npy"), np.load(tmp_path / "pe_pairs.npy")
    assert np.array_equal(got_off, want_off) and np.array_equal(got_prs, want_prs)
    for f in ("rb", "re", "qb", "qe", "rid", "score", "sub", "sub_n", "csub", "secondary", "secondary_all", "hash", "n_comp_is_alt"):
        assert np.array_equal(got[f], want[f]), f
    assert want_prs["n_matesw"].sum() > 5 and c["pes"]["failed"].tolist() == [1, 0, 1, 1]
