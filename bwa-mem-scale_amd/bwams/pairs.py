"""Extension tasks (SeqPair + flat byte buffers) from seeds — benchmark input builder.

For each read the longest non-repetitive SMEM is taken as a one-seed chain and its
left and right extension tasks are laid out the way mem_chain2aln_across_reads_V2
does for such a chain (/root/reference/src/bwamem.cpp:2880-2922 window from
cal_max_gap :94-104, :2953-3188 pair construction): left = reversed query prefix
vs reversed reference window, h0 = seed_len * a; right = query suffix vs the window
after the seed.  (The reference sets the right task's h0 to the left task's score;
this builder uses the seed score for both, which changes no code path of the DP.)
This is NOT the reference's chaining — that lives on the host side of the boundary
 and is the next row to be built; it only gives the extension
kernel a realistic task mix for bench.py.
"""
from __future__ import annotations

import numpy as np

from .capi import SEQPAIR_DTYPE


def _max_gap(qlen, a=1, o=6, e=1, w=100):
    l = ((qlen * a - o) / e + 1.0).astype(np.int64)
    l = np.maximum(l, 1)
    return np.minimum(l, w << 1)


def _gather(src: np.ndarray, start: np.ndarray, length: np.ndarray, reverse: bool) -> tuple[np.ndarray, np.ndarray]:
    """Concatenate src[start[i] : start[i]+length[i]] (optionally each reversed)."""
    off = np.zeros(len(length) + 1, dtype=np.int64)
    np.cumsum(length, out=off[1:])
    tot = int(off[-1])
    if tot == 0:
        return np.zeros(0, np.uint8), off
    seg = np.repeat(np.arange(len(length)), length)
    pos = np.arange(tot, dtype=np.int64) - off[seg]
    if reverse:
        idx = start[seg] + (length[seg] - 1 - pos)
    else:
        idx = start[seg] + pos
    return src[idx], off


def pairs_from_seeds(reads: np.ndarray, smems, coord, off, ref_0123: np.ndarray, max_occ: int = 500,
                     a: int = 1, w: int = 100):
    """reads: (n, L) uint8.  Returns (pairs, ref_buf, qer_buf)."""
    n, L = reads.shape
    l_pac = len(ref_0123) // 2
    slen = (smems["n"].astype(np.int64) - smems["m"].astype(np.int64) + 1)
    ok = (smems["s"] <= max_occ) & (smems["s"] > 0)
    # best (longest) usable seed per read: sort by (rid, -len) and take the first
    order = np.lexsort((-slen, ~ok, smems["rid"]))
    rid_sorted = smems["rid"][order]
    first = np.ones(len(order), bool)
    first[1:] = rid_sorted[1:] != rid_sorted[:-1]
    pick = order[first]
    pick = pick[ok[pick]]
    rid = smems["rid"][pick].astype(np.int64)
    qbeg = smems["m"][pick].astype(np.int64)
    ln = slen[pick]
    rbeg = coord[off[pick]]
    keep = ~((rbeg < l_pac) & (rbeg + ln > l_pac))
    rid, qbeg, ln, rbeg = rid[keep], qbeg[keep], ln[keep], rbeg[keep]
    qend = qbeg + ln
    r0 = np.maximum(rbeg - (qbeg + _max_gap(qbeg, a, w=w)), 0)
    r1 = np.minimum(rbeg + ln + (L - qend) + _max_gap(L - qend, a, w=w), 2 * l_pac)
    fw = rbeg < l_pac
    r1 = np.where(fw, np.minimum(r1, l_pac), r1)
    r0 = np.where(~fw, np.maximum(r0, l_pac), r0)

    left = qbeg > 0
    right = qend < L
    flat = np.ascontiguousarray(reads).reshape(-1)
    lq, lqo = _gather(flat, rid[left] * L, qbeg[left], True)
    lr, lro = _gather(ref_0123, r0[left], (rbeg - r0)[left], True)
    rq, rqo = _gather(flat, rid[right] * L + qend[right], (L - qend)[right], False)
    rr, rro = _gather(ref_0123, (rbeg + ln)[right], (r1 - rbeg - ln)[right], False)

    nl, nr = int(left.sum()), int(right.sum())
    pairs = np.zeros(nl + nr, dtype=SEQPAIR_DTYPE)
    pairs["id"] = np.arange(nl + nr)
    pairs["idq"][:nl] = lqo[:-1]
    pairs["idr"][:nl] = lro[:-1]
    pairs["len2"][:nl] = np.diff(lqo)
    pairs["len1"][:nl] = np.diff(lro)
    pairs["h0"][:nl] = ln[left] * a
    pairs["seqid"][:nl] = rid[left]
    pairs["idq"][nl:] = rqo[:-1] + len(lq)
    pairs["idr"][nl:] = rro[:-1] + len(lr)
    pairs["len2"][nl:] = np.diff(rqo)
    pairs["len1"][nl:] = np.diff(rro)
    pairs["h0"][nl:] = ln[right] * a
    pairs["seqid"][nl:] = rid[right]
    pairs["regid"][nl:] = 1
    assert len(lq) + len(rq) < 2**31 and len(lr) + len(rr) < 2**31
    return pairs, np.concatenate([lr, rr]), np.concatenate([lq, rq])
