"""Experiment (profiles/r03_notes.md 86): how well do cheap per-read features predict a read's round-1 SMEM cost?
Input: the dump of BWAMS_EXP_LPT_DUMP (8 x nseq uint32: cost, first-pivot cost, next_x after the first pivot,
list length at the first forward end, interval size after 20 / 28 / 36 forward steps, pivots)."""
import sys, heapq
import numpy as np

a = np.fromfile(sys.argv[1], dtype=np.uint32)
n = a.size // 8
a = a.reshape(8, n).astype(np.int64)
cost, c1, nx, np1, s20, s28, s36, npiv = a
LANES = int(sys.argv[2]) if len(sys.argv) > 2 else 196608


def makespan(order, w):
    if len(order) <= LANES:
        return int(w[order].max()) if len(order) else 0
    h = list(w[order[:LANES]])
    heapq.heapify(h)
    for i in order[LANES:]:
        t = heapq.heappop(h)
        heapq.heappush(h, t + int(w[i]))
    return max(h)


print("reads", n, "mean cost %.1f" % cost.mean(), "ideal (sum / lanes) %.0f" % (cost.sum() / LANES))
heavy = cost > 700
print("heavy (> 700): %.3f of the reads, %.3f of the work" % (heavy.mean(), cost[heavy].sum() / cost.sum()))
for name, f in (("first-pivot cost", c1), ("next_x", nx), ("list length", np1), ("s20", np.minimum(s20, 1000)), ("s28", np.minimum(s28, 1000)),
                ("s36", np.minimum(s36, 1000)), ("pivots", npiv)):
    print("  %-18s corr with cost %.3f   mean heavy %.1f / light %.1f" % (name, np.corrcoef(f, cost)[0, 1], f[heavy].mean(), f[~heavy].mean()))
for name, f in (("s20 > 1", s20 > 1), ("s28 > 1", s28 > 1), ("s36 > 1", s36 > 1), ("s28 > 2", s28 > 2), ("list >= 20", np1 >= 20), ("list >= 24", np1 >= 24)):
    tp = (f & heavy).sum(); print("  classifier %-10s flags %.3f, recall of heavy %.3f, precision %.3f" % (name, f.mean(), tp / max(1, heavy.sum()), tp / max(1, f.sum())))
ident = np.arange(n)
print("makespan (units of extensions; uniform step time):")
print("  given order        ", makespan(ident, cost))
print("  true cost, most first", makespan(np.argsort(-cost, kind="stable"), cost))
for name, f in (("s20", s20), ("s28", s28), ("s36", s36), ("list length", np1)):
    print("  by %-12s     " % name, makespan(np.argsort(-np.minimum(f, 1 << 20), kind="stable"), cost))
for name, f in (("s28 > 1", s28 > 1), ("s36 > 1", s36 > 1), ("s20 > 1", s20 > 1)):
    print("  two classes %-8s" % name, makespan(np.argsort(~f, kind="stable"), cost))
rest = cost - c1
print("two launches (first pivot, then the rest by first-pivot cost):", makespan(ident, c1), "+", makespan(np.argsort(-c1, kind="stable"), rest),
      "  rest by true rest:", makespan(np.argsort(-rest, kind="stable"), rest))
