"""Seeds per read on the bench workload (what the size classes of chain_wave_kernel see), and chains per read after chaining."""
import sys, time
import numpy as np
sys.path.insert(0, "bwa-mem-scale_amd")
from bwams import capi, simulate
n = 3_209_286_105
N = 1_000_000
g = simulate.make_genome(n, seed=2024)
ix = capi.Index.build(g, 0)
contigs = simulate.chromosomes(n)
ix.set_contigs(contigs)
reads = simulate.make_reads(g, N, seed=12345, contig_bounds=simulate.contig_bounds(contigs))[0]
enc, cum = simulate.flatten_reads(reads)
b = capi.Batch(ix, N, int(cum[-1]), max_smem=32 * N, max_sa=128 * N)
b.seed_upload(enc, cum)
opt = capi.default_mem_opt()
b.seed_run(capi.default_seed_opt(), with_sa=True)
sm, coord, sa_off = b.seed_fetch()
cnt = (sa_off[1:] - sa_off[:-1]).astype(np.int64)
per = np.bincount(sm["rid"], weights=cnt, minlength=N).astype(np.int64)
edges = [0, 32, 128, 256, 512, 850, 1275, 1700, 2500, 4096, 10 ** 9]
for lo, hi in zip(edges[:-1], edges[1:]):
    m = (per > lo) & (per <= hi)
    print(f"seeds ({lo}, {hi}]: {int(m.sum())} reads, {int(per[m].sum())} seeds")
b.chain_run(opt)
ch, sd, choff = b.chain_fetch()
nch = np.diff(choff)
for lo, hi in zip(edges[:-1], edges[1:]):
    m = (per > lo) & (per <= hi)
    if m.any(): print(f"  chains kept of reads with seeds ({lo}, {hi}]: mean {nch[m].mean():.1f} max {nch[m].max()}")
