"""Static sharding of a chunk of reads over the GPUs of one node.

Reads have no cross-read dependency in seeding and extension, so rank r of N takes
a contiguous slice of the chunk; the index is replicated.  No collective is on the
data path: ranks only exchange their per-rank result counts so that rank 0 can place
the gathered per-read results in read order (the reference emits SAM in input order).
"""
from __future__ import annotations

import numpy as np


def shard_bounds(n_reads: int, world: int, multiple_of: int = 1) -> np.ndarray:
    """world+1 offsets; every shard size is a multiple of `multiple_of` (2 for paired-end)
    except possibly the last; sizes differ by at most one unit."""
    units = (n_reads + multiple_of - 1) // multiple_of
    base, extra = divmod(units, world)
    sizes = np.array([(base + (1 if r < extra else 0)) * multiple_of for r in range(world)], dtype=np.int64)
    b = np.zeros(world + 1, dtype=np.int64)
    np.cumsum(sizes, out=b[1:])
    b[-1] = n_reads
    return np.minimum(b, n_reads)


def shard_reads(enc: np.ndarray, cum: np.ndarray, rank: int, world: int, multiple_of: int = 1):
    """(enc_shard, cum_shard rebased to 0, first_read) for this rank."""
    b = shard_bounds(len(cum) - 1, world, multiple_of)
    lo, hi = int(b[rank]), int(b[rank + 1])
    c = cum[lo:hi + 1] - cum[lo]
    return enc[cum[lo]:cum[hi]], c.astype(np.int64), lo


def gather_smems(local_smems: np.ndarray, first_read: int, dist=None):
    """Concatenate per-rank SMEM arrays in read order on every rank.

    rid is rebased from shard-local to chunk-global.  `dist` is torch.distributed (any
    backend) or None for a single process."""
    sm = local_smems.copy()
    sm["rid"] += np.uint32(first_read)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return sm
    return np.concatenate(_all_gather_bytes(sm, dist))


def _all_gather_bytes(arr: np.ndarray, dist):
    """Per-rank list of the ranks' arrays (same dtype), via two all_gathers of padded byte buffers."""
    import torch
    world = dist.get_world_size()
    # RCCL ("nccl") moves device buffers, gloo host buffers
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    raw = torch.from_numpy(np.ascontiguousarray(arr).view(np.uint8).reshape(-1).copy()).to(dev)
    n = torch.tensor([raw.numel()], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, n)
    mx = max(1, int(max(int(s.item()) for s in sizes)))
    pad = torch.zeros(mx, dtype=torch.uint8, device=dev)
    pad[:raw.numel()] = raw
    bufs = [torch.zeros(mx, dtype=torch.uint8, device=dev) for _ in range(world)]
    dist.all_gather(bufs, pad)
    return [bufs[r][:int(sizes[r].item())].cpu().numpy().view(arr.dtype) for r in range(world)]


def gather_regions(regs: np.ndarray, reg_off: np.ndarray, n_chains: int, dist=None):
    """Concatenate per-rank alignment regions (bwams_alnreg_t records grouped by read, reg_off[n_local + 1])
    in read order on every rank: -> (regs, reg_off over the whole chunk).  The chain index of a region is
    rebased by the number of chains of the preceding ranks, like rid in gather_smems."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return regs.copy(), reg_off.copy()
    parts = _all_gather_bytes(regs, dist)
    offs = _all_gather_bytes(np.ascontiguousarray(reg_off, np.int64), dist)
    nch = _all_gather_bytes(np.array([n_chains], np.int64), dist)
    out, out_off, base_reg, base_chain = [], [np.zeros(1, np.int64)], 0, 0
    for r, (p, o) in enumerate(zip(parts, offs)):
        p = p.copy()
        p["chain"] += base_chain
        out.append(p)
        out_off.append(o[1:] + base_reg)
        base_reg += len(p)
        base_chain += int(nch[r][0])
    return np.concatenate(out), np.concatenate(out_off)


def pestat_sharded(local_keys: np.ndarray, dist=None) -> np.ndarray:
    """mem_pestat for a chunk whose pairs are sharded: the one exchange step of the paired-end path.  Every rank
    contributes the insert-size keys of its qualifying pairs (capi.Batch.pestat_keys), the keys are all-gathered
    (8 bytes per pair) and every rank runs the reference's arithmetic on the union — bit-identical to the unsharded
    statistics, which depend only on the multiset of (orientation, insert size)."""
    from . import capi
    keys = np.ascontiguousarray(local_keys, np.uint64)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        keys = np.concatenate(_all_gather_bytes(keys, dist))
    return capi.pestat_from_keys(keys)


def gather_pairs(regs: np.ndarray, reg_off: np.ndarray, pairs: np.ndarray, dist=None):
    """Concatenate per-rank results of the paired-end tail in read order on every rank:
    -> (regs, reg_off over the whole chunk, pairs)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return regs.copy(), reg_off.copy(), pairs.copy()
    parts = _all_gather_bytes(regs, dist)
    offs = _all_gather_bytes(np.ascontiguousarray(reg_off, np.int64), dist)
    prs = _all_gather_bytes(pairs, dist)
    out_off, base = [np.zeros(1, np.int64)], 0
    for p, o in zip(parts, offs):
        out_off.append(o[1:] + base)
        base += len(p)
    return np.concatenate(parts), np.concatenate(out_off), np.concatenate(prs)
