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
