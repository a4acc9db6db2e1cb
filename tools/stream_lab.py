"""Streaming lab (round 4): chunks through the compiled mem_process_seqs() with 1..3 chunks in flight; host staging and PCIe inside the clock.
    python tools/stream_lab.py [--chunks 6] [--depth 3] [--index-set fm|full]"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "bwa-mem-scale_amd")):
    sys.path.insert(0, p)
ap = argparse.ArgumentParser()
ap.add_argument("--genome-mbp", type=float, default=3209.286105)
ap.add_argument("--reads", type=int, default=1_000_000)
ap.add_argument("--chunks", type=int, default=6)
ap.add_argument("--index-set", default="fm")
ap.add_argument("--extra-batch", action="store_true", help="an idle batch beside the worker, as bench.py holds one")
ap.add_argument("--depths", default="1,2,3")
ap.add_argument("--leave-free-gib", type=float, default=0.0, help="hog HBM so that only this much is free when the worker is made (memory-exhaustion rehearsal)")
args = ap.parse_args()
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from bwams import capi, simulate, stream
torch.cuda.set_device(0)
capi.lib()
G = int(round(args.genome_mbp * 1e6))
genome = simulate.make_genome(G, seed=2024)
contigs = simulate.chromosomes(G) if G >= 2 ** 31 else None
cb = None if contigs is None else simulate.contig_bounds(contigs)
ix = capi.Index.build(genome, 0)
if contigs is not None:
    ix.set_contigs(contigs)
ix.set_contig_names([b"chr%d" % (i + 1) for i in range(len(contigs) if contigs is not None else 1)])
emf = ert = None
if args.index_set == "full":
    ert = capi.Ert.build(ix)
    emf = capi.Emf.build(ix, seed_len=150, slack=1.1)
reads = simulate.make_reads(genome, args.reads, seed=12345, contig_bounds=cb)[0]
RL = reads.shape[1]
opt = capi.mem_opt_init(False)
hog = None
if args.leave_free_gib > 0:
    free_b, _ = torch.cuda.mem_get_info()
    n_hog = int(free_b - args.leave_free_gib * 2 ** 30)
    if n_hog > 0:
        hog = torch.empty(n_hog, dtype=torch.uint8, device="cuda:0")
    print(f"[stream] HBM free {torch.cuda.mem_get_info()[0] / 2**30:.1f} GiB", flush=True)
extra = None
if args.extra_batch:
    extra = capi.Batch(ix, len(reads), len(reads) * RL, max_smem=32 * len(reads), max_sa=128 * len(reads))
    d_reads = torch.from_numpy(reads.reshape(-1)).to("cuda:0")
    extra.seed_upload_device(d_reads.data_ptr(), np.arange(len(reads) + 1, dtype=np.int64) * RL)
    extra.seed_run(capi.default_seed_opt(), with_sa=True); extra.chain_run(capi.default_mem_opt()); extra.extend_run(capi.default_mem_opt()); extra.dedup_run(capi.default_mem_opt())
for depth in [int(x) for x in args.depths.split(",")]:
    overlap = depth > 1
    pre = [capi.Seqs(reads, first_id=k * len(reads)) for k in range(args.chunks + 1)]
    print(f"[stream] depth {depth}: creating the worker", flush=True)
    w = capi.Worker([ix], len(reads), len(reads) * RL, emfs=[emf] if emf else None, erts=[ert] if ert else None, depth=depth)
    print(f"[stream] depth {depth}: warm-up chunk", flush=True)
    stream.run_job(w, opt, lambda k: pre[args.chunks], 1, None)
    secs, n = stream.run_job(w, opt, lambda k: pre[k], args.chunks, None, overlap=overlap)
    w.close()
    print(f"[stream] depth {depth} overlap {overlap}: {n / secs / 1e6:.3f} Mreads/s, {secs / args.chunks * 1e3:.1f} ms per chunk", flush=True)
