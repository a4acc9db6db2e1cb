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
for i in range(40):
    st = int(rng.integers(0, len(g) - 6000)); gap = int(rng.integers(120, 200))
    r = np.concatenate([g[st:st + 2500], g[st + 2500 + gap:st + 5200 + gap]])
    pos = rng.integers(0, len(r), size=20); r[pos] = (r[pos] + 1) & 3
    reads.append(simulate.revcomp(r) if i % 2 else r)
lo, hi = int(sys.argv[1]), int(sys.argv[2])
sub = reads[lo:hi]
enc, cum = simulate.flatten_reads(sub)
b = capi.Batch(ix, len(sub), int(cum[-1]))
b.seed_upload(enc, cum); b.seed_run(capi.default_seed_opt(), with_sa=True)
opt = capi.default_mem_opt()
b.chain_run(opt); b.extend_run(opt)
regs, off, aln = b.extend_fetch()
t = time.time(); n = b.dedup_run(opt); P(lo, hi, "regs", np.diff(off).tolist(), "dedup", n, "%.2fs" % (time.time() - t))
