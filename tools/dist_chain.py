"""Distribution of seeds / chains / regions per read on the bench workload (diagnostic)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "bwa-mem-scale_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
from bwams import capi, fmindex, simulate
G = int(float(sys.argv[1]) * 1e6) if len(sys.argv) > 1 else 1000_000_000
R = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
genome = simulate.make_genome(G, seed=2024)
idx = fmindex.build_fmindex(genome, device="cuda:0", keep_ref=True)
ix = capi.Index.from_device(idx, 0)
reads, _, _ = simulate.make_reads(genome, R, seed=12345)
enc, cum = simulate.flatten_reads(reads)
b = capi.Batch(ix, R, R * reads.shape[1], max_smem=32 * R, max_sa=128 * R)
sm, coord, off = b.seed(enc, cum, capi.default_seed_opt())
per_smem = np.diff(off)
seeds_per_read = np.bincount(sm["rid"], weights=per_smem, minlength=R).astype(np.int64)
def q(x, name):
    x = np.asarray(x)
    print(name, "mean %.2f" % x.mean(), "p50", np.percentile(x, 50), "p90", np.percentile(x, 90), "p99", np.percentile(x, 99),
          "p99.9", np.percentile(x, 99.9), "max", x.max(), "sum", x.sum(), flush=True)
q(seeds_per_read, "seeds/read")
print("reads with >64 seeds:", (seeds_per_read > 64).sum(), " >512:", (seeds_per_read > 512).sum(), " >2048:", (seeds_per_read > 2048).sum())
print("seed share of reads >64: %.3f  >512: %.3f" % (seeds_per_read[seeds_per_read > 64].sum() / seeds_per_read.sum(), seeds_per_read[seeds_per_read > 512].sum() / seeds_per_read.sum()))
opt = capi.default_mem_opt()
b.chain_run(opt)
ch, sd, choff = b.chain_fetch()
q(np.diff(choff), "kept chains/read")
q(ch["n"], "seeds/chain")
b.extend_run(opt)
regs, roff, aln = b.extend_fetch()
q(np.diff(roff), "regions/read")
st = b.stats()
print("left", st.n_left, "right", st.n_right, "ms chain %.1f left %.1f right %.1f purge %.1f" % (st.ms_chain, st.ms_ext_left, st.ms_ext_right, st.ms_ext_purge))
for side in (0, 1):
    p, _, _ = b.extend_tasks_fetch(side)
    q(p["len2"], f"side{side} qlen"); q(p["len1"], f"side{side} tlen")
    print("  qlen>128:", (p["len2"] > 128).sum(), " qlen<=64:", (p["len2"] <= 64).sum())
