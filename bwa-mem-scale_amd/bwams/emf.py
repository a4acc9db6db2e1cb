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


# ----------------------------------------------------------------------------------------------
# Bench-scale builder on torch (GPU or CPU): same invariants as build_emf, vectorised.
# ----------------------------------------------------------------------------------------------
def _bst_shape(c: int):
    """(root, left[], right[]) in-group indices of the balanced BST over c sorted nodes, built
    exactly like build_emf.place(): node (lo+hi)//2 is the root of [lo, hi)."""
    left = [-1] * c
    right = [-1] * c

    def rec(lo, hi):
        if lo >= hi:
            return -1
        mid = (lo + hi) // 2
        left[mid] = rec(lo, mid)
        right[mid] = rec(mid + 1, hi)
        return mid
    root = rec(0, c)
    return root, left, right


def build_emf_torch(genome: np.ndarray, seed_len: int, device: str = "cpu", slack: float = 1.1):
    """Returns (EmfTable-like with torch tensors on `device`: loc_table int32[], seed_table int32[n,4]).

    Restrictions (asserted): seed_len in [64, 255]... any L >= 33 works; multi-location lists use the
    short encoding only (fewer than 256 copies per strand), which holds for the synthetic genomes."""
    import torch

    L = seed_len
    g = torch.from_numpy(np.ascontiguousarray(genome, dtype=np.uint8)).to(device)
    l_pac = g.numel()
    n = l_pac - L + 1
    n_entry = max(int(slack * l_pac), 16)
    MIN = -(1 << 63)

    def rolling_words(seq):
        """canonical-order packed words of every window of `seq`: list of int64 tensors of length n
        (32 bases per word, first base most significant; last word right-aligned) and the 3-word
        key of the first (L+1)//2 bases."""
        w = {1: seq.to(torch.int64)}
        k = 1
        while k < 32:
            a = w[k]
            w[2 * k] = (a[:a.numel() - k] << (2 * k)) | a[k:]
            k *= 2

        def piece(start, length):
            """packed value of `length` (<= 32) bases starting `start` bases into the window"""
            val = None
            off = start
            rem = length
            for sz in (32, 16, 8, 4, 2, 1):
                while rem >= sz:
                    part = w[sz][off: off + n]
                    val = part.clone() if val is None else ((val << (2 * sz)) | part)
                    off += sz
                    rem -= sz
            return val
        words = [piece(s, min(32, L - s)) for s in range(0, L, 32)]
        half = (L + 1) // 2
        hkey = [piece(s, min(32, half - s)) for s in range(0, half, 32)]
        return words, hkey

    fw_words, fw_half = rolling_words(g)
    rcg = (3 - g).flip(0)
    rc_words_q, rc_half_q = rolling_words(rcg)
    # the reverse complement of window p is window (n - 1 - p) of the reverse-complemented genome
    rc_words = [x.flip(0) for x in rc_words_q]
    rc_half = [x.flip(0) for x in rc_half_q]
    del rc_words_q, rc_half_q, rcg

    # fw <= rc on the first half: lexicographic on unsigned words
    le = torch.ones(n, dtype=torch.bool, device=device)
    decided = torch.zeros(n, dtype=torch.bool, device=device)
    for a, b in zip(fw_half, rc_half):
        au, bu = a ^ MIN, b ^ MIN
        lt, gt = au < bu, au > bu
        le = torch.where(~decided & gt, torch.zeros_like(le), le)
        decided = decided | lt | gt
    fw_less = le
    del fw_half, rc_half, decided, le
    canon = [torch.where(fw_less, a, b) for a, b in zip(fw_words, rc_words)]
    del fw_words, rc_words

    # hash (perfect.h:541-707): XOR of the words, fmix64, unsigned modulo
    def lsr(x, s):
        return (x >> s) & ((1 << (64 - s)) - 1)
    h = canon[0].clone()
    for c in canon[1:]:
        h ^= c
    h = h ^ lsr(h, 33)
    h = h * (-49064778989728563)            # 0xff51afd7ed558ccd as int64
    h = h ^ lsr(h, 33)
    h = h * (-4265267296055464877)          # 0xc4ceb9fe1a85ec53 as int64
    h = h ^ lsr(h, 33)
    hi, lo = lsr(h, 32), h & 0xffffffff
    key = ((hi % n_entry) * ((1 << 32) % n_entry) + lo) % n_entry
    del h, hi, lo

    # group identical canonical L-mers: lexsort on the unsigned words (stable: positions ascending inside a group)
    order = torch.arange(n, dtype=torch.int64, device=device)
    for c in reversed(canon):
        _, idx = torch.sort((c ^ MIN)[order], stable=True)
        order = order[idx]
        del idx
    new_grp = torch.ones(n, dtype=torch.bool, device=device)
    acc = torch.zeros(max(n - 1, 0), dtype=torch.bool, device=device)
    for c in canon:
        cs = c[order]
        acc |= cs[1:] != cs[:-1]
        del cs
    new_grp[1:] = acc
    del acc, canon
    gid = torch.cumsum(new_grp.to(torch.int64), 0) - 1                 # seed id (canonical order) of every sorted position
    n_u = int(gid[-1].item()) + 1
    first_pos = order[new_grp]                                          # entry location of every seed
    ent_fw = fw_less[first_pos]
    seed_key = key[first_pos]
    gsize = torch.bincount(gid, minlength=n_u)

    # multi-location lists (short encoding), assembled on the host: such seeds are few
    multi_ids = torch.nonzero(gsize > 1).reshape(-1)
    ent_multi = torch.zeros(n_u, dtype=torch.int64, device=device)
    loc_list = [0]
    if multi_ids.numel():
        in_multi = (gsize[gid] > 1)
        mpos = order[in_multi].cpu().numpy()
        mgid = gid[in_multi].cpu().numpy()
        mfw = fw_less[order[in_multi]].cpu().numpy()
        starts = np.flatnonzero(np.r_[True, mgid[1:] != mgid[:-1]])
        ends = np.append(starts[1:], len(mgid))
        multi_of = np.zeros(len(starts), dtype=np.int64)
        for gi, (s, e) in enumerate(zip(starts, ends)):
            same = [int(x) for x, f in zip(mpos[s + 1:e], mfw[s + 1:e]) if f == mfw[s]]
            other = [int(x) for x, f in zip(mpos[s + 1:e], mfw[s + 1:e]) if f != mfw[s]]
            assert len(same) < LOC_MANY and len(other) < LOC_MANY, "long multi-location lists: use build_emf"
            multi_of[gi] = len(loc_list)
            loc_list.append((len(same) << 16) | len(other))
            loc_list += same + other
        ent_multi[torch.from_numpy(mgid[starts]).to(device)] = torch.from_numpy(multi_of).to(device)
    del order, gid, new_grp, key, fw_less

    # one BST per hash value; seeds of a key are in canonical order after a stable sort by key
    _, by_key = torch.sort(seed_key, stable=True)
    ks = seed_key[by_key]
    kstart = torch.ones(n_u, dtype=torch.bool, device=device)
    kstart[1:] = ks[1:] != ks[:-1]
    kid = torch.cumsum(kstart.to(torch.int64), 0) - 1
    n_key = int(kid[-1].item()) + 1
    ksize = torch.bincount(kid, minlength=n_key)
    kfirst = torch.nonzero(kstart).reshape(-1)
    in_grp = torch.arange(n_u, dtype=torch.int64, device=device) - kfirst[kid]      # index inside its key group
    csize = ksize[kid]
    cmax = int(ksize.max().item())
    tab_root = torch.zeros(cmax + 1, dtype=torch.int64)
    tab_left = torch.full((cmax + 1, cmax), -1, dtype=torch.int64)
    tab_right = torch.full((cmax + 1, cmax), -1, dtype=torch.int64)
    for c in range(1, cmax + 1):
        r, lf, rt = _bst_shape(c)
        tab_root[c] = r
        tab_left[c, :c] = torch.tensor(lf)
        tab_right[c, :c] = torch.tensor(rt)
    tab_root, tab_left, tab_right = tab_root.to(device), tab_left.to(device), tab_right.to(device)
    is_root = in_grp == tab_root[csize]
    # slots: roots at their key, the others in the free slots
    slot = torch.empty(n_u, dtype=torch.int64, device=device)
    slot[is_root] = ks[is_root]
    used = torch.zeros(n_entry, dtype=torch.bool, device=device)
    used[ks[is_root]] = True
    n_other = int((~is_root).sum().item())
    if n_other:
        free = torch.nonzero(~used).reshape(-1)[:n_other]
        slot[~is_root] = free
        del free
    del used
    li, ri = tab_left[csize, in_grp], tab_right[csize, in_grp]
    base = kfirst[kid]
    NO = NO_ENTRY - (1 << 32)                                            # 0xffffffff as int32 bit pattern
    left = torch.where(li >= 0, slot[(base + li.clamp(min=0))], torch.full_like(li, NO_ENTRY))
    right = torch.where(ri >= 0, slot[(base + ri.clamp(min=0))], torch.full_like(ri, NO_ENTRY))
    u = by_key                                                           # seed id of every key-sorted node
    flags = ent_fw[u].to(torch.int64) | ((~is_root).to(torch.int64) << 1) | (ent_multi[u] << 2)

    def as_i32(x):
        return torch.where(x >= (1 << 31), x - (1 << 32), x).to(torch.int32)
    seeds = torch.full((n_entry, 4), NO, dtype=torch.int32, device=device)
    seeds[:, 0] = 0
    seeds[slot, 0] = as_i32(flags)
    seeds[slot, 1] = as_i32(first_pos[u])
    seeds[slot, 2] = as_i32(left)
    seeds[slot, 3] = as_i32(right)
    loc_table = torch.tensor(np.array(loc_list, dtype=np.int64), device=device)
    return EmfTable(L, l_pac, as_i32(loc_table), seeds, n_u, n_key)
