#!/usr/bin/env python3
"""bench.py — throughput of the MI355X seed-and-extend hot path.

One "step" = one pass of the hot path over one resident batch of synthetic reads:
  FM-index seeding (SMEM rounds 1-3 -> (rid,m,n) sort -> SA lookup) -> seed chaining and
  chain filtering -> extension tasks of the kept chains' seeds -> banded-SW left and right
  extension with the band-retry rule -> region bookkeeping and purge (in rounds: a seed the
  reference would extend and then discard is not extended) -> mem_sort_dedup_patch,
all on the GPU through the C-ABI, with reads and index resident in HBM when the clock starts.
Workload = BASELINE.json configs[1] (1M x 150 bp single-end, FM-index only, 1 GPU);
GRCh38 is not available offline, so the index is built (on the GPU) over a seeded
synthetic genome whose size is stated in the output.  With --gpus N every rank
holds a replica of the index and its own 1M-read shard (weak scaling, no collective
on the data path).

Prints ONE JSON line (rank 0).  See DESIGN.md "Measurement" for the definitions of
roofline.achieved (algorithmic bytes) and cpu_baseline.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "bwa-mem-scale_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def cpu_baseline(idx_host_arrays, reads, n_sample, threads):
    """Time the oracle (CPU restatement, kind="port") on a bounded sample of the same reads:
    seeding + SA lookup + chaining + chain-to-alignment (the same stages as the GPU step).  The
    oracle is the checker; it is timed here only as the reported CPU column."""
    from concurrent.futures import ThreadPoolExecutor

    from bwams import simulate
    from oracle import loader

    o = loader.OracleFMI(idx_host_arrays)
    l_pac = (idx_host_arrays.ref_seq_len - 1) // 2
    sample = reads[:n_sample]
    chunks = np.array_split(np.arange(len(sample)), threads)

    def work(ix):
        n = 0
        for a in range(0, len(ix), 4096):            # bounded scratch per call
            sub = sample[ix[a:a + 4096]]
            enc, cum = simulate.flatten_reads(sub)
            sm = o.collect_smem(enc, cum)
            coord, off = o.sa_lookup(sm, 500)
            ch, sd, choff = loader.chain_seeds(sm, coord, off, cum, l_pac)
            regs, reg_off, _ = loader.chain2aln(ch, sd, choff, enc, cum, idx_host_arrays.ref_0123, l_pac)
            fin, _ = loader.regs_finish(regs, reg_off, enc, cum, idx_host_arrays.ref_0123, l_pac)
            n += len(fin)
        return n

    # one timed region around everything
    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(work, chunks))
    dt = time.perf_counter() - t0
    return len(sample) / dt / 1e6, dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--genome-mbp", type=float, default=float(os.environ.get("BWAMS_GENOME_MBP", "1000")))
    ap.add_argument("--reads", type=int, default=1_000_000, help="reads per GPU")
    ap.add_argument("--cpu-sample", type=int, default=1_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fma", action="store_true", help="build and use the FMA tables (all_smem.11 / last_smem.13)")
    ap.add_argument("--emf", action="store_true", help="build the exact-match filter table (L=150) on the GPU and probe it first")
    ap.add_argument("--pcie", action="store_true", help="also time the one-call host-buffer form (PCIe inclusive)")
    ap.add_argument("--no-pe", action="store_true", help="skip the paired-end leg (mate rescue + pairing; reported beside, never `value`)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from bwams import capi, fmindex, simulate

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists in the product path)")
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device(dev))
    if not os.path.exists(capi.LIB_PATH):
        if rank == 0:
            capi.build()
        if world > 1:
            dist.barrier()
    capi.lib()

    # ---------------- inputs (untimed) ----------------
    t0 = time.time()
    G = int(args.genome_mbp * 1e6)
    genome = simulate.make_genome(G, seed=2024)
    log(f"genome {G/1e6:.0f} Mbp generated in {time.time()-t0:.1f}s")
    t0 = time.time()
    idx_dev = fmindex.build_fmindex(genome, device=dev, keep_ref=True)
    torch.cuda.synchronize()
    log(f"FM-index built on GPU in {time.time()-t0:.1f}s: text {idx_dev.ref_seq_len/1e9:.2f} G rows, "
        f"CP_OCC {idx_dev.cp_occ.numel()*8/2**30:.2f} GiB")
    ix = capi.Index.from_device(idx_dev, local)
    torch.cuda.empty_cache()
    if args.fma:
        t0 = time.time()
        ix.build_fma(11, 13)
        log(f"FMA tables built on GPU in {time.time()-t0:.1f}s (512 MiB + 1 GiB)")

    emf_h = None
    if args.emf:
        from bwams import emf as emf_mod
        t0 = time.time()
        emf_tab = emf_mod.build_emf_torch(genome, 150, dev)
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        emf_h = capi.Emf(ix, device_table=emf_tab)
        log(f"EMF table built on GPU in {time.time()-t0:.1f}s: {emf_tab.seed_table.shape[0]/1e6:.0f} M entries "
            f"({emf_tab.seed_table.numel()*4/2**30:.1f} GiB), {emf_tab.num_seed_used/1e6:.0f} M seeds")
    R = args.reads
    t0 = time.time()
    reads, _, _ = simulate.make_reads(genome, R, seed=12345 + rank)
    enc, cum = simulate.flatten_reads(reads)
    log(f"{R} reads generated in {time.time()-t0:.1f}s")

    batch = capi.Batch(ix, R, R * reads.shape[1], max_smem=32 * R, max_sa=128 * R)
    batch.seed_upload(enc, cum)
    ref_host = np.concatenate([genome, (3 - genome[::-1]).astype(np.uint8)])

    seed_opt = capi.default_seed_opt()
    mem_opt = capi.default_mem_opt()

    def step():
        # reads -> [EMF] -> seeds -> chains -> extension tasks -> banded SW (left, right, retries) -> regions
        if emf_h is not None:
            batch.emf_run(emf_h)
        batch.seed_run(seed_opt, with_sa=True)
        batch.chain_run(mem_opt)
        batch.extend_run(mem_opt)
        batch.dedup_run(mem_opt)

    for _ in range(args.warmup):
        step()
    batch.sync()

    # ---------------- timed region ----------------
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    per_step = []
    for _ in range(args.steps):
        step()
        st = batch.stats()                 # synchronises the batch stream; part of the timed region
        per_step.append(st)
    batch.sync()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t_start
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---------------- report ----------------
    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        value = R * world * args.steps / elapsed / 1e6
        st = per_step[-1]
        r1_ms = float(np.mean([s.ms_smem_r1 for s in per_step]))
        # algorithmic bytes of the round-1 search kernel per launch:
        # 64 B per CP_OCC block an extension touches + reads in (1 B/base) + SMEMs out (40 B)
        r1_bytes = 64 * st.n_blk_round[0] + int(cum[-1]) + 40 * st.n_smem[0]
        achieved = r1_bytes / (r1_ms * 1e-3) / 1e9
        all_bytes = (64 * st.n_ext_blocks + int(cum[-1]) * 3 + 40 * sum(st.n_smem) +
                     64 * st.n_lf_steps + 13 * st.n_sa_lookups)
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "r01_pmc_summary.json")
        if os.path.exists(tfile):
            try:
                tj = json.load(open(tfile))
                if tj.get("genome_mbp") == args.genome_mbp and tj.get("reads") == R:
                    traffic = tj.get("smem_round1_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            # BASELINE.json: "Mreads/sec aligned, GRCh38 150bp SE, at 1/2/4/8 MI355X; % HBM roofline" — same metric; GRCh38 is
            # not available offline, so the genome is synthetic (config.workload) and the roofline share is `roofline.frac`
            "metric": f"Mreads/sec aligned, 150bp SE (synthetic {args.genome_mbp:.0f} Mbp genome in place of GRCh38), at {world} MI355X; "
                      "% HBM roofline in roofline.frac",
            "value": round(value, 4),
            "unit": "Mreads/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_step, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int64 intervals / int32 DP",
            "data": "synthetic",
            "config": {
                "workload": f"{R} synthetic 150bp SE reads per GPU vs synthetic {args.genome_mbp:.0f} Mbp genome "
                            f"(GRCh38 unavailable offline), FM-index{' + FMA tables' if args.fma else ''}{' + EMF (L=150)' if args.emf else ''}"
                            f"{'' if (args.fma or args.emf) else ' only (no ERT/FMA/EMF)'}; step = {'EMF probe, ' if args.emf else ''}pack reads, SMEM r1-r3, sort, "
                            f"SA lookup, chaining + chain filter, extension tasks of the seeds of the kept chains, "
                            f"banded-SW left then right (w=100, retry at 200), region bookkeeping + purge (seeds the reference would extend "
                            f"and then discard are not extended), mem_sort_dedup_patch; everything on the GPU",
                "genome_mbp": args.genome_mbp,
                "index_bytes": ix.nbytes,
                "reads_per_gpu": R,
                "chains": int(st.n_chains), "regions": int(st.n_chain_seeds),
                "bsw_tasks": int(st.n_left + st.n_right), "bsw_retries": int(st.n_retry_left + st.n_retry_right),
                "ext_rounds": int(st.n_ext_rounds), "final_regions": int(st.n_final_regs),
                "parallelism": f"reads sharded x{world}, index replicated",
            },
            "stage_ms": {
                "smem_round1": round(r1_ms, 3),
                "smem_round2": round(float(np.mean([s.ms_smem_r2 for s in per_step])), 3),
                "smem_round3": round(float(np.mean([s.ms_smem_r3 for s in per_step])), 3),
                "sort": round(float(np.mean([s.ms_sort for s in per_step])), 3),
                "sa_lookup": round(float(np.mean([s.ms_sal for s in per_step])), 3),
                "seed_total": round(float(np.mean([s.ms_seed_total for s in per_step])), 3),
                "emf": round(float(np.mean([s.ms_emf for s in per_step])), 3),
                "chain": round(float(np.mean([s.ms_chain for s in per_step])), 3),
                "ext_tasks": round(float(np.mean([s.ms_ext_plan for s in per_step])), 3),
                "ext_left": round(float(np.mean([s.ms_ext_left for s in per_step])), 3),
                "ext_right": round(float(np.mean([s.ms_ext_right for s in per_step])), 3),
                "ext_select": round(float(np.mean([s.ms_ext_purge for s in per_step])), 3),
                "ext_total": round(float(np.mean([s.ms_ext_total for s in per_step])), 3),
                "dedup": round(float(np.mean([s.ms_dedup for s in per_step])), 3),
                "note": "ext_tasks/left/right/select are the first extension round; ext_total covers all rounds",
            },
            "events_per_read": {
                "backward_ext": round(st.n_ext / R, 2),
                "backward_ext_by_round": [round(x / R, 2) for x in st.n_ext_round],
                "cp_occ_blocks_by_round": [round(x / R, 2) for x in st.n_blk_round],
                "cp_occ_blocks": round(st.n_ext_blocks / R, 2),
                "smems": round(sum(st.n_smem) / R, 2),
                "sa_lookups": round(st.n_sa_lookups / R, 2),
                "lf_steps": round(st.n_lf_steps / R, 2),
                "bsw_cells": round(st.bsw_cells / R, 1),
                "algorithmic_bytes": round(all_bytes / R, 1),
            },
            "roofline": {
                "kernel": "smem_search_kernel<ALL_POS> (SMEM round 1)",
                "bound": "hbm",
                "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "bytes_per_launch": int(r1_bytes),
                "launch_ms": round(r1_ms, 3),
            },
        }
        ext_ms = float(np.mean([s.ms_ext_total for s in per_step]))
        out["extension"] = {
            "kernels": "bsw_kernel_reg<1|2|3> (banded SW, integer VALU bound: neither of the contract's roofs applies)",
            "tasks": int(st.n_left + st.n_right), "dp_cells": int(st.bsw_cells), "ms_all_rounds": round(ext_ms, 3),
            "Gcells_per_s": round(st.bsw_cells / (ext_ms * 1e-3) / 1e9, 2) if ext_ms > 0 else None,
            "Mtasks_per_s": round((st.n_left + st.n_right) / (ext_ms * 1e-3) / 1e6, 2) if ext_ms > 0 else None,
        }
        if emf_h is not None:
            _, codes = batch.emf_fetch(R)
            emf_ms = float(np.mean([s.ms_emf for s in per_step]))
            emf_bytes = 16 * st.emf_nodes + st.emf_cmp_bytes + int(cum[-1])
            out["emf"] = {"resolved_fraction": round(float(((codes == 3) | (codes == 4)).mean()), 4),
                          "launch_ms": round(emf_ms, 3), "nodes_per_read": round(st.emf_nodes / R, 3),
                          "algorithmic_bytes": int(emf_bytes),
                          "achieved_GBps": round(emf_bytes / (emf_ms * 1e-3) / 1e9, 1) if emf_ms > 0 else None}
        if args.pcie:
            # host buffers in, host buffers out (bwams_seed_fmi + bwams_bsw_extend): never `value`
            t0 = time.perf_counter()
            for _ in range(2):
                batch.seed(enc, cum, seed_opt)              # upload reads, run, download SMEMs + SA coordinates
                batch.chain_run(mem_opt)
                batch.chain_fetch()                         # download chains
                batch.extend_run(mem_opt)
                batch.dedup_run(mem_opt)
                batch.dedup_fetch()                         # download the final regions
            dt = (time.perf_counter() - t0) / 2
            out["pcie_inclusive"] = {"value": round(R / dt / 1e6, 4), "unit": "Mreads/s", "ms_per_batch": round(dt * 1e3, 2),
                                     "note": "pageable host buffers: reads up; SMEMs, SA coordinates, chains and final regions down; includes numpy copies"}
        if not args.no_pe and world == 1:
            # paired-end leg (BASELINE config 5's path on one GPU): the same step on R/2 FR pairs, then mem_pestat,
            # mate rescue, mem_mark_primary_se and mem_pair.  Reported beside the headline, never `value`.
            t0 = time.time()
            pr = simulate.make_read_pairs_bulk(genome, R // 2, seed=4242)
            penc, pcum = simulate.flatten_reads(pr)
            log(f"{R // 2} read pairs generated in {time.time()-t0:.1f}s")
            batch.seed_upload(penc, pcum)

            def pe_step():
                step()
                pes_ = batch.pestat(mem_opt)
                n_, nt_ = batch.pair_run(pes_, mem_opt)
                return pes_, n_, nt_

            pe_step()
            batch.sync()
            t0 = time.perf_counter()
            for _ in range(2):
                pes_, n_pe, nt_pe = pe_step()
            batch.sync()
            dt = (time.perf_counter() - t0) / 2
            pst = batch.stats()
            _, _, prs = batch.pair_fetch()
            out["paired_end"] = {
                "value": round(2 * (R // 2) / dt / 1e6, 4), "unit": "Mreads/s", "ms_per_batch": round(dt * 1e3, 2),
                "pairs": R // 2, "ms_pestat_plus_pair": round(dt * 1e3 - float(pst.ms_seed_total + pst.ms_chain + pst.ms_ext_total + pst.ms_dedup), 2),
                "ms_pair_run": round(float(pst.ms_pair), 3), "rescue_alignments": int(nt_pe), "reads_redone": int(pst.n_pair_redone),
                "regions_after_rescue": int(n_pe), "proper_pairs": round(float((prs["score"] > 0).mean()), 4),
                "orientations_failed": [int(x) for x in pes_["failed"]], "insert_avg_std": [round(float(pes_["avg"][1]), 2), round(float(pes_["std"][1]), 2)],
                "note": "2x150bp FR pairs (insert 400 +- 40, 5 % with a damaged end, 2 % discordant): SE step + mem_pestat + mate rescue "
                        "(ksw_align2 on the GPU) + mem_mark_primary_se + mem_pair; regions and pairing decisions stay on the device",
            }
            batch.seed_upload(enc, cum)
        if not args.no_cpu_baseline and world == 1:
            log("timing the CPU oracle on a sample (cpu_baseline)...")
            threads = min(16, os.cpu_count() or 1)
            n_s = min(args.cpu_sample, R)
            # host copy of the index for the oracle
            host = fmindex.FMIndex(
                idx_dev.ref_seq_len, idx_dev.count,
                idx_dev.cp_occ.cpu().numpy().view(np.uint64),
                idx_dev.sa_ms_byte.cpu().numpy(), idx_dev.sa_ls_word.cpu().numpy().view(np.uint32),
                idx_dev.sentinel_index, ref_host)
            v, dt = cpu_baseline(host, reads, n_s, threads)
            out["cpu_baseline"] = {
                "value": round(v, 5), "unit": "Mreads/s", "cores": threads, "kind": "port",
                "sample": f"first {n_s} reads of the same batch, same index; oracle seeding+SA+chaining+chain2aln+dedup, "
                          f"{dt:.1f}s wall on {threads} threads",
            }
        print(json.dumps(out), flush=True)
    batch.close()
    ix.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
