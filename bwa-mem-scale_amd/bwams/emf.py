"""Exact-match filter (EMF) table: file format and a small-genome builder.

Format = the reference's `<prefix>.perfect.<L>` (/root/reference/src/perfect.h:188-213, :772-822):
  64-byte packed header  int seed_len; u32 num_loc_entry, num_seed_entry, num_seed_load;
                         3 stale pointers (24 B); u32 seq_len, num_seed_used, num_seed_key; 12 B pad
  u32 loc_table[num_loc_entry]            (entry 0 unused)
  seed_entry_t seed_table[num_seed_entry] ({u32 flags, location, left, right}; location = 0xffffffff: empty)

The reference's builder (perfect_index.cpp, out of scope: SURVEY.md §2 row 16) places collision
nodes by an order-dependent linear probe; any table that satisfies the format's invariants is
probed identically, so this builder (tooling for tests, numpy, small genomes) places them
simply: one BST per hash value ordered by the canonical L-mer, root at the hash slot, the other
nodes in free slots.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

NO_ENTRY = 0xFFFFFFFF
LOC_MANY = 256


@dataclass
class EmfTable:
    seed_len: int
    seq_len: int
    loc_table: np.ndarray        # uint32[num_loc_entry]
    seed_table: np.ndarray       # uint32[num_seed_entry, 4] = flags, location, left, right
    num_seed_used: int = 0
    num_seed_key: int = 0


def _fmix64(k: np.ndarray) -> np.ndarray:
    k = k.astype(np.uint64)
    k ^= k >> np.uint64(33); k *= np.uint64(0xff51afd7ed558ccd)
    k ^= k >> np.uint64(33); k *= np.uint64(0xc4ceb9fe1a85ec53)
    k ^= k >> np.uint64(33)
    return k


def canonical_and_hash(seqs: np.ndarray, num_seed_entry: int):
    """seqs: (n, L) uint8 codes 0..3.  Returns (canonical (n, L), fw_less bool[n], hash int64[n])."""
    n, L = seqs.shape
    half = (L + 1) // 2
    rc = (3 - seqs[:, ::-1]).astype(np.uint8)
    a, b = seqs[:, :half], rc[:, :half]
    diff = a != b
    first = np.where(diff.any(axis=1), diff.argmax(axis=1), 0)
    av = a[np.arange(n), first]; bv = b[np.arange(n), first]
    fw_less = ~diff.any(axis=1) | (av <= bv)
    canon = np.where(fw_less[:, None], seqs, rc).astype(np.uint8)
    h = np.zeros(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        full = L - L % 32
        for w0 in range(0, full, 32):
            word = np.zeros(n, dtype=np.uint64)
            for i in range(32):
                word = (word << np.uint64(2)) | canon[:, w0 + i].astype(np.uint64)
            h ^= word
        if L % 32:
            word = np.zeros(n, dtype=np.uint64)
            for i in range(full, L):
                word = (word << np.uint64(2)) | canon[:, i].astype(np.uint64)
            h ^= word
        hv = _fmix64(h) % np.uint64(num_seed_entry)
    return canon, fw_less, hv.astype(np.int64)


def build_emf(genome: np.ndarray, seed_len: int, slack: float = 1.1) -> EmfTable:
    """genome: forward strand codes 0..3 (one contig, no N)."""
    g = np.asarray(genome, dtype=np.uint8)
    l_pac = len(g)
    L = seed_len
    n_entry = max(int(slack * l_pac), 16)
    win = np.lib.stride_tricks.sliding_window_view(g, L)           # (l_pac - L + 1, L)
    canon, fw_less, hv = canonical_and_hash(win, n_entry)
    uniq, inv = np.unique(canon, axis=0, return_inverse=True)       # lexicographic order of canonical L-mers
    inv = inv.reshape(-1)
    n_u = len(uniq)
    order = np.argsort(inv, kind="stable")                          # locations grouped by seed, ascending inside
    starts = np.searchsorted(inv[order], np.arange(n_u))
    ends = np.append(starts[1:], len(order))
    seed_key = hv[order[starts]]

    seeds = np.zeros((n_entry, 4), dtype=np.uint32)
    seeds[:, 1:] = NO_ENTRY
    loc_table = [0]                                                 # entry 0 unused
    ent_flags = np.zeros(n_u, dtype=np.uint32)
    ent_loc = np.zeros(n_u, dtype=np.uint32)
    for u in range(n_u):
        locs = order[starts[u]:ends[u]]
        first = int(locs[0])
        fl = 1 if fw_less[first] else 0
        if len(locs) > 1:
            same = [int(x) for x in locs[1:] if bool(fw_less[x]) == bool(fw_less[first])]
            other = [int(x) for x in locs[1:] if bool(fw_less[x]) != bool(fw_less[first])]
            multi = len(loc_table)
            if len(same) < LOC_MANY and len(other) < LOC_MANY:
                loc_table.append((len(same) << 16) | len(other))
                loc_table += same + other
            else:
                loc_table.append(0x80000000 | (multi + 1))
                loc_table += [len(same), len(other)] + same + other
            fl |= multi << 2
        ent_flags[u] = fl
        ent_loc[u] = first

    # one BST per hash value: seeds of a key are already in canonical order
    by_key = np.argsort(seed_key, kind="stable")
    ks = seed_key[by_key]
    kstart = np.flatnonzero(np.r_[True, ks[1:] != ks[:-1]])
    kend = np.append(kstart[1:], len(ks))
    roots = set(int(k) for k in ks[kstart])
    free = (i for i in range(n_entry) if i not in roots)
    n_key = 0
    for s, e in zip(kstart, kend):
        members = by_key[s:e]                                       # ascending canonical order
        key = int(ks[s])
        slot = {}

        def place(lo, hi, is_root):
            if lo >= hi:
                return NO_ENTRY
            mid = (lo + hi) // 2
            u = int(members[mid])
            idx = key if is_root else next(free)
            slot[u] = idx
            left = place(lo, mid, False)
            right = place(mid + 1, hi, False)
            seeds[idx] = (int(ent_flags[u]) | (0 if is_root else 2), int(ent_loc[u]), left, right)
            return idx
        place(0, len(members), True)
        n_key += 1
    return EmfTable(L, l_pac, np.array(loc_table, dtype=np.uint32), seeds, n_u, n_key)


def write_emf(path: str, t: EmfTable) -> None:
    hdr = np.zeros(64, dtype=np.uint8)
    hdr[0:4] = np.array([t.seed_len], dtype="<i4").view(np.uint8)
    hdr[4:16] = np.array([len(t.loc_table), len(t.seed_table), len(t.seed_table)], dtype="<u4").view(np.uint8)
    hdr[40:52] = np.array([t.seq_len, t.num_seed_used, t.num_seed_key], dtype="<u4").view(np.uint8)
    with open(path, "wb") as f:
        f.write(hdr.tobytes())
        f.write(np.ascontiguousarray(t.loc_table, dtype="<u4").tobytes())
        f.write(np.ascontiguousarray(t.seed_table, dtype="<u4").tobytes())


def read_emf(path: str) -> EmfTable:
    raw = np.fromfile(path, dtype=np.uint8)
    seed_len = int(raw[0:4].view("<i4")[0])
    n_loc, n_seed, _ = (int(x) for x in raw[4:16].view("<u4"))
    seq_len, used, nkey = (int(x) for x in raw[40:52].view("<u4"))
    o = 64
    loc = raw[o:o + 4 * n_loc].view("<u4").copy(); o += 4 * n_loc
    seeds = raw[o:o + 16 * n_seed].view("<u4").reshape(n_seed, 4).copy()
    assert o + 16 * n_seed == len(raw), "file size does not match its header"
    return EmfTable(seed_len, seq_len, loc, seeds, used, nkey)
