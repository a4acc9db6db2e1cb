"""Debug: random extension tasks through the device kernels vs the oracle; list the tasks that differ."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bwa-mem-scale_amd"), os.path.join(ROOT, "tests")]
import numpy as np
from bwams import capi, fmindex, simulate
from oracle import loader
from util import make_pairs, toy
_, idx = toy(3000)
ix = capi.Index.from_host(idx, 0)
b = capi.Batch(ix, 16, 4096)
tot = bad_n = 0
for seed in range(1, 9):
    for mq in (30, 60, 100, 140, 190):
        pairs, ref, qer = make_pairs(3000, seed=seed * 10 + mq, max_q=mq)
        for w in (100, 200, 20):
            got = b.bsw(pairs, ref, qer, w)
            want, _ = loader.bsw_pairs(pairs, ref, qer, w)
            bad = np.zeros(len(got), bool)
            for f in ("score", "tle", "gtle", "qle", "gscore", "max_off"):
                bad |= got[f] != want[f]
            tot += len(got); bad_n += int(bad.sum())
            for t in np.flatnonzero(bad)[:6]:
                print(f"seed {seed} mq {mq} w {w} task {t}: qlen {got['len2'][t]} tlen {got['len1'][t]} h0 {got['h0'][t]} got",
                      [int(got[f][t]) for f in ("score", "tle", "gtle", "qle", "gscore", "max_off")], "want",
                      [int(want[f][t]) for f in ("score", "tle", "gtle", "qle", "gscore", "max_off")], flush=True)
print("tasks", tot, "differing", bad_n)
