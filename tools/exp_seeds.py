"""distribution of seeds (sampled hits) per read on the bench workload; chain-stage time with BWAMS_VERBOSE"""
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
b.seed_upload(enc, cum); b.seed_run()
sm, co, off = b.seed_fetch()
per = np.bincount(sm["rid"], weights=np.diff(off), minlength=len(reads)).astype(np.int64)
print("seeds per read: mean %.1f max %d p99 %d p99.9 %d" % (per.mean(), per.max(), np.percentile(per, 99), np.percentile(per, 99.9)), flush=True)
for t in (32, 128, 256, 512, 850, 1700, 3000, 5000):
    print(f"reads with > {t} seeds: {(per > t).sum()}  (seeds in them: {per[per > t].sum()})", flush=True)
os.environ["BWAMS_VERBOSE"] = "1"
mo = capi.default_mem_opt()
for _ in range(2):
    b.chain_run(mo); b.sync()
st = b.stats()
print("chain ms", st.ms_chain, flush=True)
