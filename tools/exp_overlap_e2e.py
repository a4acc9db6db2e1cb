"""bwams_process_chunk on one batch against two batches driven by two host threads (chunks alternate): does the SAM side of one chunk
overlap the seeding of the next?"""
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
ix.set_contig_names([b"chr%d" % (i + 1) for i in range(len(contigs))])
N = 1_000_000
texts = []
for c in range(2):
    rd = simulate.make_reads(genome, N, seed=12345 + c, contig_bounds=cb)[0]
    RL = rd.shape[1]
    row = np.empty((N, 1 + 9 + 1 + RL + 3 + RL + 1), np.uint8)
    row[:, 0] = ord("@"); row[:, 1] = ord("r")
    ids = np.arange(N, dtype=np.int64) + c * N
    for d in range(8):
        row[:, 2 + d] = ord("0") + (ids // 10 ** (7 - d)) % 10
    row[:, 10] = 10
    row[:, 11:11 + RL] = np.frombuffer(b"ACGTN", np.uint8)[rd]
    row[:, 11 + RL:14 + RL] = np.frombuffer(b"\n+\n", np.uint8)
    row[:, 14 + RL:14 + 2 * RL] = ord("I")
    row[:, 14 + 2 * RL] = 10
    texts.append((torch.from_numpy(row.reshape(-1)).cuda(), row.size))
batches = [capi.Batch(ix, N, N * RL, max_smem=32 * N, max_sa=128 * N) for _ in range(2)]


def run(bi, ci):
    d, n = texts[ci]
    return batches[bi].process_chunk((d.data_ptr(), n), n_processed=ci * N, fetch=False)


for bi in range(2):
    run(bi, bi)
torch.cuda.synchronize()
K = 6
t0 = time.perf_counter()
for k in range(K):
    run(0, k & 1)
torch.cuda.synchronize()
serial = (time.perf_counter() - t0) / K * 1e3
def worker(bi):
    for k in range(K // 2):
        run(bi, bi)
t0 = time.perf_counter()
ts = [threading.Thread(target=worker, args=(bi,)) for bi in range(2)]
[t.start() for t in ts]; [t.join() for t in ts]
torch.cuda.synchronize()
conc = (time.perf_counter() - t0) / K * 1e3
print(f"one batch: {serial:.1f} ms per chunk ({N / serial / 1e3:.2f} Mreads/s); two batches / two threads: {conc:.1f} ms per chunk ({N / conc / 1e3:.2f} Mreads/s)")
