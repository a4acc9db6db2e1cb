"""Seeding lab (round 4): the bench genome, index and reads built ONCE, then the seeding stage alone under a list of
environment variants (the hand-over thresholds of fmi_seed.hip are read per call).

    python tools/seed_lab.py [--genome-mbp M] [--reads N] [--steps K] name:ENV=V,ENV=V ...

Prints one line per variant: round 1 / 2 / 3 / SA lookup / seed_total in ms (HIP events of the library), round 1's fraction of
the byte peak in algorithmic bytes, and the event counts (which must not move: the variants are schedules, not algorithms).
A different build of the library is a different process: BWAMS_LIB=<path> python tools/seed_lab.py ...
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "bwa-mem-scale_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome-mbp", type=float, default=3209.286105)
    ap.add_argument("--reads", type=int, default=1_000_000)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--profile", default="uniform")
    ap.add_argument("--step", action="store_true", help="run the whole step (seed .. dedup), not just seeding")
    ap.add_argument("variants", nargs="*")
    args = ap.parse_args()
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import torch

    from bwams import capi, simulate
    torch.cuda.set_device(0)
    capi.lib()
    G = int(round(args.genome_mbp * 1e6))
    t0 = time.time()
    genome = simulate.make_genome(G, seed=2024) if args.profile == "uniform" else simulate.make_genome(G, seed=2024, profile=args.profile)
    contigs = simulate.chromosomes(G) if G >= 2 ** 31 else None
    cb = None if contigs is None else simulate.contig_bounds(contigs)
    ix = capi.Index.build(genome, 0)
    if contigs is not None:
        ix.set_contigs(contigs)
    reads = simulate.make_reads(genome, args.reads, seed=12345, contig_bounds=cb)[0]
    RL = reads.shape[1]
    d_reads = torch.from_numpy(reads.reshape(-1)).to("cuda:0")
    cum = np.arange(len(reads) + 1, dtype=np.int64) * RL
    print(f"[lab] inputs in {time.time()-t0:.1f}s: {G/1e6:.0f} Mbp, {len(reads)} reads, lib {capi.LIB_PATH}", flush=True)
    batch = capi.Batch(ix, len(reads), len(reads) * RL, max_smem=32 * len(reads), max_sa=128 * len(reads))
    opt = capi.default_seed_opt()
    mopt = capi.default_mem_opt()
    ref = None
    for v in (args.variants or ["default:"]):
        name, _, envs = v.partition(":")
        sets = [e.split("=", 1) for e in envs.split(",") if e]
        old = {k: os.environ.get(k) for k, _ in sets}
        for k, val in sets:
            os.environ[k] = val
        capi.debug_reload()
        rows = []
        for i in range(args.warmup + args.steps):
            batch.seed_upload_device(d_reads.data_ptr(), cum)
            t_0 = time.perf_counter()
            batch.seed_run(opt, with_sa=True)
            if args.step:
                batch.chain_run(mopt)
                batch.extend_run(mopt)
                batch.dedup_run(mopt)
            st = batch.stats()
            wall = (time.perf_counter() - t_0) * 1e3
            if i >= args.warmup:
                rows.append((st.ms_smem_r1, st.ms_smem_r2, st.ms_smem_r3, st.ms_sal, st.ms_seed_total, st.ms_chain, st.ms_ext_total, st.ms_dedup, wall,
                             st.ms_ext_plan, st.ms_ext_left, st.ms_ext_right, st.ms_ext_purge))
        for k, _ in sets:
            if old[k] is None:
                del os.environ[k]
            else:
                os.environ[k] = old[k]
        m = np.mean(rows, axis=0)
        mn = np.min(rows, axis=0)
        ev = (int(st.n_ext), int(st.n_ext_blocks), tuple(int(x) for x in st.n_smem), int(st.n_sa_lookups))
        if ref is None:
            ref = ev
        bytes_r1 = 64 * int(st.n_blk_round[0]) + len(reads) * RL + 40 * int(st.n_smem[0])
        frac = bytes_r1 / (m[0] * 1e-3) / 8e12
        if args.step:
            print(f"[lab] {name:28s} step {m[8]:7.2f} (min {mn[8]:7.2f})  seed {m[4]:6.2f}  chain {m[5]:6.2f}  ext {m[6]:6.2f} (plan {m[9]:5.2f} left {m[10]:6.2f} right {m[11]:6.2f} purge {m[12]:5.2f})"
                  f"  dedup {m[7]:6.2f}  rounds {int(st.n_ext_rounds)} tasks {int(st.n_left + st.n_right)} cells {int(st.bsw_cells)} chains {int(st.n_chains)} regs {int(st.n_final_regs)}", flush=True)
        print(f"[lab] {name:28s} r1 {m[0]:6.2f} (min {mn[0]:6.2f})  r2 {m[1]:6.2f}  r3 {m[2]:6.2f}  sal {m[3]:5.2f}  total {m[4]:6.2f} (min {mn[4]:6.2f})"
              f"  frac {frac:.4f}  events {'same' if ev == ref else 'DIFFER ' + str(ev)}", flush=True)
    batch.close()
    ix.close()


if __name__ == "__main__":
    main()
