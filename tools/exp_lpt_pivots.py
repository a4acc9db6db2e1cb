"""Experiment (profiles/r03_notes.md 86): per-pivot log of SMEM round 1 (BWAMS_EXP_LPT_DUMP): what if the backward phase of the
pivots whose interval list is long went to a cooperative kernel (one lane per list entry)?"""
import sys, heapq
import numpy as np
raw = np.fromfile(sys.argv[1], dtype=np.uint32)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
feat = raw[:8 * n].reshape(8, n).astype(np.int64)
cost = feat[0]
npv = int(raw[8 * n])
log = raw[8 * n + 16: 8 * n + 16 + 2 * min(npv, 4 << 20)].reshape(-1, 2)
rid = log[:, 0].astype(np.int64)
np0 = (log[:, 1] & 255).astype(np.int64)
cols = ((log[:, 1] >> 8) & 255).astype(np.int64)
bc = (log[:, 1] >> 16).astype(np.int64)
print("pivots logged", npv, "backward share of the work %.3f" % (bc.sum() / cost.sum()))
print("backward cost quantiles", np.quantile(bc, [.5, .9, .99, .999, .9999]), "max", bc.max())
print("list length quantiles", np.quantile(np0, [.5, .9, .99, .999]), "max", np0.max())
print("corr(list length, backward cost) %.3f  corr(list x columns, cost) %.3f" % (np.corrcoef(np0, bc)[0, 1], np.corrcoef(np0 * cols, bc)[0, 1]))
L = 196608
def makespan(w):
    h = list(w[:L]); heapq.heapify(h)
    for v in w[L:]:
        t = heapq.heappop(h); heapq.heappush(h, t + int(v))
    return max(h)
print("lane-per-read makespan, given order:", makespan(cost), " ideal", cost.sum() // L, " max read", cost.max())
for T in (20, 22, 24, 26, 28, 32):
    hv = np0 >= T
    off = np.bincount(rid[hv], weights=bc[hv], minlength=n).astype(np.int64)
    rest = cost - off
    print("T %2d: pivots offloaded %.4f (%d), their work %.3f of all, columns %d (mean %.1f, max %d) | rest: makespan %d, max read %d, ideal %d"
          % (T, hv.mean(), hv.sum(), bc[hv].sum() / cost.sum(), cols[hv].sum(), cols[hv].mean(), cols[hv].max(), makespan(rest), rest.max(), rest.sum() // L))
# what if EVERY backward phase is a separate work item (forward kernel + backward-item kernel)
fwd = cost - np.bincount(rid, weights=bc, minlength=n).astype(np.int64)
print("forward-only per read: mean %.1f max %d makespan %d" % (fwd.mean(), fwd.max(), makespan(fwd)))
print("backward items, lane per item, given order: makespan %d, most first %d, ideal %d" % (makespan(bc), makespan(np.sort(bc)[::-1]), bc.sum() // L))
