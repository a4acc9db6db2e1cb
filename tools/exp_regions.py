"""distribution of regions per read before de-duplication (bench workload)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bwa-mem-scale_amd")]
import numpy as np, torch
from bwams import capi, simulate
torch.cuda.init()
G = 3_209_286_105
genome = simulate.make_genome(G, seed=2024)
contigs = simulate.chromosomes(G); cb = simulate.contig_bounds(contigs)
ix = capi.Index.build(genome, 0); ix.set_contigs(contigs)
reads = simulate.make_reads(genome, 1_000_000, seed=12345, contig_bounds=cb)[0]
enc, cum = simulate.flatten_reads(reads)
b = capi.Batch(ix, len(reads), int(cum[-1]), max_smem=32 * len(reads), max_sa=128 * len(reads))
b.seed_upload(enc, cum); b.seed_run(); mo = capi.default_mem_opt(); b.chain_run(mo); b.extend_run(mo)
regs, off, _ = b.extend_fetch()
n = np.diff(off)
alive = np.add.reduceat((regs["qe"] > regs["qb"]).astype(np.int64), off[:-1][n > 0]) if len(regs) else []
print("regions per read: max", n.max(), "p99", np.percentile(n, 99), "p99.9", np.percentile(n, 99.9), flush=True)
for t in (32, 128, 256, 512, 1024, 2048):
    print(f"reads with > {t} regions: {(n > t).sum()}  (regions in them: {n[n > t].sum()})", flush=True)
