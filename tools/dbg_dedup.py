import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "bwa-mem-scale_amd")): sys.path.insert(0, p)
import numpy as np
from bwams import capi, fmindex, simulate
def P(*a): print(*a, flush=True)
g = simulate.make_genome(120000, seed=31, repeat_frac=0.45, repeat_len=260, n_families=4)
idx = fmindex.build_fmindex(g)
ix = capi.Index.from_host(idx, 0)
rng = np.random.default_rng(71)
reads = []
nlong = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for i in range(nlong):
    st = int(rng.integers(0, len(g) - 6000)); gap = int(rng.integers(120, 200))
    r = np.concatenate([g[st:st + 2500], g[st + 2500 + gap:st + 5200 + gap]])
    pos = rng.integers(0, len(r), size=20); r[pos] = (r[pos] + 1) & 3
    reads.append(simulate.revcomp(r) if i % 2 else r)
reads += list(simulate.make_reads(g, 300, seed=6)[0])
enc, cum = simulate.flatten_reads(reads)
b = capi.Batch(ix, len(reads), int(cum[-1]))
b.seed_upload(enc, cum); b.seed_run(capi.default_seed_opt(), with_sa=True)
opt = capi.default_mem_opt()
P("chain", b.chain_run(opt))
t = time.time(); P("extend", b.extend_run(opt), time.time() - t)
regs, off, aln = b.extend_fetch()
P("regs/read max", np.diff(off).max(), "reads >32:", (np.diff(off) > 32).sum())
t = time.time(); P("dedup", b.dedup_run(opt), time.time() - t)
