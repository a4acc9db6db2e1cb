"""FM-index construction and the reference's on-disk format.

Produces exactly what ``FMI_search::build_index`` / ``build_fm_index`` write
(/root/reference/src/FMI_search.cpp:611-849):

  <prefix>.bwt.2bit.64 = int64 ref_seq_len (= 2*l_pac + 1)
                         int64 count[5]        (cumulative A,C,G,T,total; WITHOUT the
                                                +1 the loader adds, :880-883)
                         CP_OCC[(ref_seq_len >> 6) + 1]   (64 B each)
                         int8  sa_ms_byte[(ref_seq_len >> 3) + 1]
                         uint32 sa_ls_word[(ref_seq_len >> 3) + 1]
                         int64 sentinel_index
  <prefix>.0123        = 2*l_pac bytes, forward strand then reverse complement.

The text is fw || revcomp(fw) over {0,1,2,3}; BWT row 0 is the suffix that starts
at the appended terminator.  Suffix sorting here is prefix doubling on sort
primitives (numpy on the host, torch on a GPU for benchmark-sized genomes) — an
MI355X-first replacement for the reference's single-threaded SA-IS; the output
arrays are identical by definition of the suffix array.
"""
from __future__ import annotations

import os
from dataclasses import dataclass

import numpy as np

CP_SHIFT = 6
SA_COMPX = 3


@dataclass
class FMIndex:
    """In-memory FM-index as the reference holds it after load (count has +1)."""
    ref_seq_len: int
    count: np.ndarray            # int64[5], +1 applied
    cp_occ: np.ndarray           # uint64[(n_blk, 8)]: 4 counts (as int64 bits) + 4 one-hot strings
    sa_ms_byte: np.ndarray       # int8
    sa_ls_word: np.ndarray       # uint32
    sentinel_index: int
    ref_0123: np.ndarray | None = None   # uint8[2*l_pac]

    @property
    def l_pac(self) -> int:
        return (self.ref_seq_len - 1) // 2


def fw_rc_text(genome: np.ndarray) -> np.ndarray:
    g = np.asarray(genome, dtype=np.uint8)
    assert g.max(initial=0) < 4, "N must be replaced before indexing (bns_fasta2bntseq does)"
    return np.concatenate([g, (3 - g[::-1]).astype(np.uint8)])


# ----------------------------------------------------------------------------
# suffix array by prefix doubling
# ----------------------------------------------------------------------------
def _suffix_array_numpy(t: np.ndarray) -> np.ndarray:
    """SA of t + '$' ('$' smallest): int64[n+1], SA[0] == n."""
    n = t.shape[0]
    n1 = n + 1
    K = 27                                   # 5**27 < 2**63
    s = np.zeros(n1 + K, dtype=np.int64)
    s[:n] = t.astype(np.int64) + 1           # 0 = terminator / past the end
    key = np.zeros(n1, dtype=np.int64)
    for i in range(K):
        key = key * 5 + s[i:i + n1]
    _, rank = np.unique(key, return_inverse=True)
    rank = rank.astype(np.int64)
    h = K
    while int(rank.max()) < n1 - 1:
        r2 = np.zeros(n1, dtype=np.int64)
        if h < n1:
            r2[:n1 - h] = rank[h:] + 1
        key = rank * (n1 + 1) + r2
        _, rank = np.unique(key, return_inverse=True)
        rank = rank.astype(np.int64)
        h *= 2
    sa = np.empty(n1, dtype=np.int64)
    sa[rank] = np.arange(n1, dtype=np.int64)
    return sa


def _suffix_array_torch(t: np.ndarray, device: str) -> "torch.Tensor":
    """Same as _suffix_array_numpy, on a torch device; returns a device int64 tensor."""
    import torch

    n = t.shape[0]
    n1 = n + 1
    K = 27
    tt = torch.from_numpy(t).to(device)
    s = torch.zeros(n1 + K, dtype=torch.int64, device=device)
    s[:n] = tt.to(torch.int64) + 1
    del tt
    key = torch.zeros(n1, dtype=torch.int64, device=device)
    for i in range(K):
        key.mul_(5).add_(s[i:i + n1])
    del s

    def dense_rank(k):
        ks, order = torch.sort(k)
        del k
        flag = torch.ones(n1, dtype=torch.int64, device=device)
        flag[0] = 0
        flag[1:] = (ks[1:] != ks[:-1]).to(torch.int64)
        del ks
        rs = torch.cumsum(flag, 0)
        del flag
        r = torch.empty(n1, dtype=torch.int64, device=device)
        r[order] = rs
        return r, int(rs[-1].item())

    rank, mx = dense_rank(key)
    h = K
    while mx < n1 - 1:
        r2 = torch.zeros(n1, dtype=torch.int64, device=device)
        if h < n1:
            r2[:n1 - h] = rank[h:] + 1
        key = rank * (n1 + 1) + r2
        del r2
        rank, mx = dense_rank(key)
        h *= 2
    sa = torch.empty(n1, dtype=torch.int64, device=device)
    sa[rank] = torch.arange(n1, dtype=torch.int64, device=device)
    return sa


# ----------------------------------------------------------------------------
# BWT -> CP_OCC, sampled SA
# ----------------------------------------------------------------------------
def build_fmindex(genome: np.ndarray, device: str | None = None, keep_ref: bool = True):
    """Build the FM-index of fw||rc(genome).

    device=None -> numpy on the host, returns FMIndex of numpy arrays.
    device='cuda:0' -> torch on that GPU, returns FMIndex whose arrays are torch
    device tensors (cp_occ int64[n_blk, 8]); used by bench.py so that a
    benchmark-sized index never crosses PCIe.
    """
    text = fw_rc_text(genome)
    n = text.shape[0]
    L = n + 1
    if device is None:
        sa = _suffix_array_numpy(text)
        bwt = np.full(((L + 63) // 64) * 64, 6, dtype=np.uint8)   # DUMMY_CHAR padding
        prev = sa - 1
        sent = int(np.flatnonzero(sa == 0)[0])
        bwt[:L] = text[np.where(prev >= 0, prev, 0)]
        bwt[sent] = 4
        counts_raw = np.bincount(text, minlength=4).astype(np.int64)
        count = np.zeros(5, dtype=np.int64)
        count[1:] = np.cumsum(counts_raw)
        n_blk = (L >> CP_SHIFT) + 1
        cp = np.zeros((n_blk, 8), dtype=np.uint64)
        filled = (L + 63) // 64                      # blocks whose first row is < L
        b2 = bwt.reshape(-1, 64)
        weights = (np.uint64(1) << np.arange(63, -1, -1, dtype=np.uint64))
        for c in range(4):
            hot = (b2 == c)
            per_blk = hot.sum(axis=1).astype(np.int64)
            cum = np.zeros(filled + 1, dtype=np.int64)
            np.cumsum(per_blk, out=cum[1:])
            cp[:filled, c] = cum[:filled].astype(np.uint64)
            cp[:filled, 4 + c] = (hot.astype(np.uint64) * weights[None, :]).sum(axis=1, dtype=np.uint64)
        n_sa = (L >> SA_COMPX) + 1
        samp = np.zeros(n_sa, dtype=np.int64)
        s8 = sa[::8]
        samp[:s8.shape[0]] = s8
        idx = FMIndex(L, count + 1, cp, ((samp >> 32) & 0xff).astype(np.uint8).view(np.int8),
                      (samp & 0xffffffff).astype(np.uint32), sent,
                      text if keep_ref else None)
        return idx

    import torch
    sa = _suffix_array_torch(text, device)
    tt = torch.from_numpy(text).to(device)
    sent = int(torch.nonzero(sa == 0)[0].item())
    Lp = ((L + 63) // 64) * 64
    bwt = torch.full((Lp,), 6, dtype=torch.uint8, device=device)
    bwt[:L] = tt[torch.clamp(sa - 1, min=0)]
    bwt[sent] = 4
    counts_raw = torch.bincount(tt.to(torch.int64), minlength=4).cpu().numpy().astype(np.int64)
    count = np.zeros(5, dtype=np.int64)
    count[1:] = np.cumsum(counts_raw)
    n_blk = (L >> CP_SHIFT) + 1
    filled = (L + 63) // 64
    cp = torch.zeros((n_blk, 8), dtype=torch.int64, device=device)
    b2 = bwt.view(-1, 64)
    # bit 63-j for column j; build as two 32-bit halves to stay inside int64 arithmetic
    w_hi = (1 << torch.arange(31, -1, -1, dtype=torch.int64, device=device))
    for c in range(4):
        hot = (b2 == c)
        per_blk = hot.sum(dim=1, dtype=torch.int64)
        cum = torch.cumsum(per_blk, 0) - per_blk
        cp[:filled, c] = cum[:filled]
        hi = (hot[:, :32].to(torch.int64) * w_hi[None, :]).sum(dim=1)
        lo = (hot[:, 32:].to(torch.int64) * w_hi[None, :]).sum(dim=1)
        cp[:filled, 4 + c] = (hi << 32) | lo          # wraps into the sign bit as intended
        del hot, per_blk, cum, hi, lo
    del bwt, b2
    n_sa = (L >> SA_COMPX) + 1
    samp = torch.zeros(n_sa, dtype=torch.int64, device=device)
    s8 = sa[::8]
    samp[:s8.shape[0]] = s8
    del sa
    ms = ((samp >> 32) & 0xff).to(torch.uint8).view(torch.int8)
    ls = (samp & 0xffffffff).to(torch.int64)
    ls32 = torch.where(ls >= 2**31, ls - 2**32, ls).to(torch.int32)   # bit pattern of uint32
    return FMIndex(L, count + 1, cp, ms, ls32, sent, tt if keep_ref else None)


# ----------------------------------------------------------------------------
# file format
# ----------------------------------------------------------------------------
def write_index(prefix: str, idx: FMIndex) -> None:
    """Write <prefix>.bwt.2bit.64 (+ .0123) byte-for-byte as the reference does."""
    with open(prefix + ".bwt.2bit.64", "wb") as f:
        f.write(np.int64(idx.ref_seq_len).tobytes())
        f.write((np.asarray(idx.count, dtype=np.int64) - 1).tobytes())
        f.write(np.ascontiguousarray(idx.cp_occ).tobytes())
        f.write(np.ascontiguousarray(idx.sa_ms_byte).tobytes())
        f.write(np.ascontiguousarray(idx.sa_ls_word).tobytes())
        f.write(np.int64(idx.sentinel_index).tobytes())
    if idx.ref_0123 is not None:
        np.asarray(idx.ref_0123, dtype=np.uint8).tofile(prefix + ".0123")


def read_index(prefix: str) -> FMIndex:
    """Reader mirroring __load_BWT_from_file (src/FMI_search.cpp:855-930)."""
    path = prefix + ".bwt.2bit.64"
    raw = np.memmap(path, dtype=np.uint8, mode="r")
    L = int(raw[:8].view(np.int64)[0])
    count = raw[8:48].view(np.int64).astype(np.int64) + 1
    n_blk = (L >> CP_SHIFT) + 1
    n_sa = (L >> SA_COMPX) + 1
    o = 48
    cp = np.array(raw[o:o + n_blk * 64]).view(np.uint64).reshape(n_blk, 8)
    o += n_blk * 64
    ms = np.array(raw[o:o + n_sa]).view(np.int8)
    o += n_sa
    ls = np.array(raw[o:o + 4 * n_sa]).view(np.uint32)
    o += 4 * n_sa
    sent = int(np.array(raw[o:o + 8]).view(np.int64)[0])
    ref = None
    if os.path.exists(prefix + ".0123"):
        ref = np.fromfile(prefix + ".0123", dtype=np.uint8)
    return FMIndex(L, count, cp, ms, ls, sent, ref)
