"""SMEM round 1 with the reads in their given order against the same reads sorted by descending a-posteriori cost (their SMEM
count): how much of the kernel is the tail of the slowest reads?"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
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
so = capi.default_seed_opt()
b = capi.Batch(ix, len(reads), reads.size, max_smem=32 * len(reads), max_sa=128 * len(reads))


def run(rd, label):
    enc, cum = simulate.flatten_reads(rd)
    b.seed_upload(enc, cum)
    b.seed_run(so, with_sa=True)
    r1 = []
    for _ in range(3):
        b.seed_run(so, with_sa=True)
        st = b.stats()
        r1.append((st.ms_smem_r1, st.ms_smem_r2, st.ms_smem_r3, st.ms_seed_total))
    print(label, "r1 r2 r3 seed_total:", np.round(np.mean(r1, axis=0), 2), flush=True)
    sm, coord, off = b.seed_fetch()
    return sm


sm = run(reads, "given order     ")
cnt = np.bincount(sm["rid"], minlength=len(reads))
ext = np.zeros(len(reads), np.int64)
np.add.at(ext, sm["rid"], (sm["n"] - sm["m"] + 1).astype(np.int64))
print("SMEMs per read: mean %.1f p99 %d max %d" % (cnt.mean(), np.percentile(cnt, 99), cnt.max()))
for label, key in (("most SMEMs first", -cnt), ("fewest first    ", cnt), ("random shuffle  ", np.random.default_rng(1).random(len(reads)))):
    order = np.argsort(key, kind="stable")
    run(reads[order], label)
