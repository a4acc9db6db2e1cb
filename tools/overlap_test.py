"""Experiment: one batch of R reads vs K concurrent batches of R/K reads (own streams, host threads)."""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bwa-mem-scale_amd"))
import numpy as np
import torch
from bwams import capi, fmindex, simulate

G = int(float(sys.argv[1]) * 1e6) if len(sys.argv) > 1 else 1000_000_000
R = 1_000_000
torch.cuda.set_device(0)
capi.lib()
genome = simulate.make_genome(G, seed=2024)
idx_dev = fmindex.build_fmindex(genome, device="cuda:0", keep_ref=True)
ix = capi.Index.from_device(idx_dev, 0)
torch.cuda.empty_cache()
reads, _, _ = simulate.make_reads(genome, R, seed=12345)
so, mo = capi.default_seed_opt(), capi.default_mem_opt()

def run(K, steps=4):
    n = R // K
    batches = []
    for k in range(K):
        enc, cum = simulate.flatten_reads(reads[k * n:(k + 1) * n])
        b = capi.Batch(ix, n, n * reads.shape[1], max_smem=32 * n, max_sa=128 * n)
        b.seed_upload(enc, cum)
        batches.append(b)
    def work(b):
        b.seed_run(so, with_sa=True); b.chain_run(mo); b.extend_run(mo); b.dedup_run(mo)
    for b in batches:
        work(b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        th = [threading.Thread(target=work, args=(b,)) for b in batches]
        for t in th: t.start()
        for t in th: t.join()
    for b in batches: b.sync()
    dt = (time.perf_counter() - t0) / steps
    print(f"K={K}: {dt*1e3:.1f} ms per {R} reads = {R/dt/1e6:.2f} Mreads/s", flush=True)
    for b in batches: b.close()

for K in (1, 2, 4):
    run(K)
