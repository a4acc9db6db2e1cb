"""Round 4 debugging aid: the seeding parity test's inputs under several hand-over threshold sets; prints the SMEMs that differ."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "bwa-mem-scale_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
torch.cuda.init()
from bwams import capi, simulate
from oracle import loader
from util import toy

g, idx = toy(200000, seed=13)
ix = capi.Index.from_host(idx, 0)
reads, _, _ = simulate.make_reads(g, 3000, seed=21)
reads = list(reads)
rng = np.random.default_rng(4)
for L in (150, 400, 1000):
    st = int(rng.integers(0, len(g) - 1100))
    reads.append(g[st:st + L].copy())
reads.append(np.zeros(120, np.uint8))
reads.append(np.tile(np.array([0, 1], np.uint8), 100))
enc, cum = simulate.flatten_reads(reads)
o = loader.OracleFMI(idx)
oopt = loader.default_seed_opt(); gopt = capi.default_seed_opt()
oopt.min_seed_len = 12; gopt.min_seed_len = 12
want = o.collect_smem(enc, cum, oopt)
wcoord, woff = o.sa_lookup(want, oopt.max_occ)
variants = sys.argv[1:] or ["200,200,200,2,1,1"]
names = ["BWAMS_BWD_MIN_LIST", "BWAMS_BWD_COLS", "BWAMS_BWD_LATE_LIST", "BWAMS_BWD_DRY_MIN_LIST", "BWAMS_BWD_DRY_COLS", "BWAMS_BWD_DRY_LATE_LIST"]
for v in variants:
    for k, x in zip(names, v.split(",")):
        os.environ[k] = x
    b = capi.Batch(ix, len(cum) - 1, int(cum[-1]), max_smem=len(want) + 4096, max_sa=len(wcoord) + 4096)
    got, coord, off = b.seed(enc, cum, gopt)
    b.close()
    key = lambda a: set(zip(a["rid"].tolist(), a["m"].tolist(), a["n"].tolist(), a["k"].tolist(), a["s"].tolist()))
    kw, kg = key(want), key(got)
    print(v, "want", len(want), "got", len(got), "missing", sorted(kw - kg)[:5], "extra", sorted(kg - kw)[:5], flush=True)
    if len(got) != len(want) and not (kw ^ kg):
        # duplicates
        from collections import Counter
        cg = Counter(zip(got["rid"].tolist(), got["m"].tolist(), got["n"].tolist())); cw = Counter(zip(want["rid"].tolist(), want["m"].tolist(), want["n"].tolist()))
        print("  dup diff:", [(k_, cg[k_], cw[k_]) for k_ in cg if cg[k_] != cw[k_]][:5], "read len", [len(reads[k_[0]]) for k_ in cg if cg[k_] != cw[k_]][:5])
