"""Does running two half-chunks on two batches (two host threads, two streams) overlap the memory-bound seeding of one with the
VALU-bound extension of the other?  Serial 1 M reads vs 2 x 500 k concurrently (bench workload)."""
import os, sys, time, threading
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
so, mo = capi.default_seed_opt(), capi.default_mem_opt()


def make(n_parts):
    parts = np.array_split(reads, n_parts)
    out = []
    for p in parts:
        enc, cum = simulate.flatten_reads(p)
        b = capi.Batch(ix, len(p), int(cum[-1]), max_smem=32 * len(p), max_sa=128 * len(p))
        d = torch.from_numpy(enc).cuda()
        out.append((b, d, cum))
    return out


def run(item):
    b, d, cum = item
    b.seed_upload_device(d.data_ptr(), cum)
    b.seed_run(so, with_sa=True); b.chain_run(mo); b.extend_run(mo); b.dedup_run(mo)
    b.stats()


def timed(items, threads, reps=4):
    def once():
        if threads:
            ts = [threading.Thread(target=run, args=(it,)) for it in items]
            [t.start() for t in ts]; [t.join() for t in ts]
        else:
            for it in items:
                run(it)
    once(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        once()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for n_parts in (1, 2, 3, 4):
    items = make(n_parts)
    s = timed(items, False)
    c = timed(items, True) if n_parts > 1 else s
    print(f"{n_parts} part(s): serial {s:.1f} ms, concurrent {c:.1f} ms  ({1e3 / c:.2f} Mreads/s)", flush=True)
    for b, d, cum in items:
        b.close()
