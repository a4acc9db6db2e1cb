#!/usr/bin/env python3
"""bench.py — throughput of the MI355X seed-and-extend hot path.

One "step" = one pass of the hot path over this rank's resident reads, chunk by chunk:
  reads (in HBM) -> pack -> FM-index seeding (SMEM rounds 1-3 -> (rid,m,n) sort -> SA lookup) -> seed chaining and
  chain filtering -> extension tasks of the kept chains' seeds -> banded-SW left and right extension with the
  band-retry rule -> region bookkeeping and purge (in rounds: a seed the reference would extend and then discard
  is not extended) -> mem_sort_dedup_patch,
all on the GPU through the C-ABI, with reads and index resident in HBM when the clock starts.

Workload = BASELINE.json configs[1]: 1 M x 150 bp single-end reads per GPU, FM-index only, against an index of
GRCh38's size.  GRCh38 itself is not available offline: the genome is a seeded synthetic one of GRCh38's own l_pac
(3 209 286 105 bases in 24 sequences, 10 % interspersed repeats), its FM-index (6.4 G rows) built on the GPU by
bwams_index_build.  `--gpus N` = one rank per GPU (launched by torchrun / the driver, or spawned here when WORLD_SIZE
is unset), every rank holding a replica of the index and its own shard of reads, no collective on the data path
(`--scaling weak`: --reads per rank; `--scaling strong`: --reads-total split over the ranks, BASELINE config 4's shape).

Prints ONE JSON line (rank 0).  DESIGN.md "Measurement" defines roofline.achieved (algorithmic bytes) and cpu_baseline.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
# counter-measured figures come from the newest committed PMC summary (profiles/rNN_pmc_summary.json, stamped with commit and workload)
PMC_SUMMARY = next((p for p in (os.path.join(ROOT, "profiles", f"r{n:02d}_pmc_summary.json") for n in range(9, 0, -1)) if os.path.exists(p)),
                   os.path.join(ROOT, "profiles", "r02_pmc_summary.json"))
for p in (ROOT, os.path.join(ROOT, "bwa-mem-scale_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
GRCH38_L_PAC = 3_209_286_105   # /root/reference/src/bwa_shm.cpp:1386 (reference_seq_len = 2 * l_pac + 1)
# MI355X_MICROARCH.md: 256 CUs x 4 SIMD, one VALU wave-instruction per 2 cycles per SIMD (more than one wave resident),
# one SALU instruction per cycle per CU, 2.4 GHz
# integer VALU issue measured on the chip with four waves per SIMD (tools/ubench_valu.hip, gpurun 2026-10-04): v_max_i32, v_cndmask_b32,
# v_cmp, v_bfe, v_perm, DPP and the v_pk_*_i16 forms 555-570 G wave-instructions/s (4.4 cycles per SIMD), v_add_u32 / v_sub_u32 910 G/s
# (2.7 cycles); the banded-SW kernel's mix is about one add/sub in four: 1 / (0.75 / 560 + 0.25 / 910)
VALU_PEAK_GINST = round(1.0 / (0.75 / 560.0 + 0.25 / 910.0), 1)
SALU_PEAK_GINST = 256 * 2.4
# the reference itself, timed in the build container: the tree does not travel to the GPU box
REFERENCE_MEASURED = {
    "value": 0.0797, "unit": "Mreads/s", "cores": 8, "isa": "AVX512BW", "mode": "plain (bwa-mem2-equivalent) FM-index, mem -t 8",
    "genome": "100 Mbp synthetic random genome, 400k x 150bp SE", "where": "build container (8 cores, 62 GB), BASELINE.md section 3b",
    "per_thread_kreads_s": 9.96, "with_fma": 0.100, "with_fma_emf": 0.139,
    "note": "whole mem_process_seqs incl. SAM; measured once by the survey stage, not in this run",
}


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        free = ""
        try:
            import torch
            if torch.cuda.is_available() and torch.cuda.is_initialized():
                f_, t_ = torch.cuda.mem_get_info()
                free = f" [HBM free {f_ / 2**30:.0f} of {t_ / 2**30:.0f} GiB]"
        except Exception:                              # noqa: BLE001 - a progress line must never end the run
            pass
        print("[bench]", *a, free, file=sys.stderr, flush=True)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--genome-mbp", type=float, default=float(os.environ.get("BWAMS_GENOME_MBP", GRCH38_L_PAC / 1e6)),
                    help="synthetic genome size; default = GRCh38's l_pac")
    ap.add_argument("--reads", type=int, default=1_000_000, help="reads per GPU (weak scaling)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--reads-total", type=int, default=8_000_000, help="reads of the whole job (strong scaling)")
    ap.add_argument("--chunk-reads", type=int, default=1_000_000, help="reads per resident chunk (one bwams batch)")
    ap.add_argument("--cpu-sample", type=int, default=1_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fma", action="store_true", help="build and use the FMA tables (all_smem.11 / last_smem.13)")
    ap.add_argument("--emf", action="store_true", help="build the exact-match filter table (L=150) on the GPU; reads it resolves skip seeding and get their regions from it")
    ap.add_argument("--no-ert-leg", action="store_true", help="skip the beside leg that repeats the steps with seeding over the ERT")
    ap.add_argument("--ert", action="store_true", help="build the ERT index (k-mer table + radix trees) on the GPU and seed over it instead of the FM-index")
    ap.add_argument("--pcie", action="store_true", help="also time the one-call host-buffer form (PCIe inclusive)")
    ap.add_argument("--no-pe", action="store_true", help="skip the paired-end leg (mate rescue + pairing; reported beside, never `value`)")
    ap.add_argument("--no-hard-genome", action="store_true", help="skip the beside leg that repeats the steps on the grch38_like genome profile")
    ap.add_argument("--dry-run", action="store_true", help="no GPU work: rendezvous (gloo), barriers and the JSON line only")
    return ap.parse_args(argv)


def spawn_ranks(args) -> int:
    """`bench.py --gpus N` without a launcher: start N child ranks — BEFORE anything in this process touches the GPU —
    and relay rank 0's JSON line.  (With a launcher, RANK / LOCAL_RANK / WORLD_SIZE come from the environment.)"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


def cpu_baseline(idx_host_arrays, reads, n_sample, threads, contigs):
    """Time the oracle (CPU restatement, kind="port") on a bounded sample of the same reads:
    seeding + SA lookup + chaining + chain-to-alignment + dedup (the same stages as the GPU step).  The
    oracle is the checker; it is timed here only as the reported CPU column."""
    from concurrent.futures import ThreadPoolExecutor

    from bwams import simulate
    from oracle import loader

    o = loader.OracleFMI(idx_host_arrays)
    l_pac = (idx_host_arrays.ref_seq_len - 1) // 2
    sample = reads[:n_sample]
    chunks = np.array_split(np.arange(len(sample)), threads)
    done = [0] * threads
    budget_s = float(os.environ.get("BWAMS_CPU_BUDGET_S", "20"))
    t0 = time.perf_counter()

    def work(k):
        ix = chunks[k]
        for a in range(0, len(ix), 1024):            # bounded scratch per call; stop when the time budget is spent
            if time.perf_counter() - t0 > budget_s:
                break
            sub = sample[ix[a:a + 1024]]
            enc, cum = simulate.flatten_reads(sub)
            sm = o.collect_smem(enc, cum)
            coord, off = o.sa_lookup(sm, 500)
            ch, sd, choff = loader.chain_seeds(sm, coord, off, cum, l_pac, contigs=contigs)
            regs, reg_off, _ = loader.chain2aln(ch, sd, choff, enc, cum, idx_host_arrays.ref_0123, l_pac, contigs=contigs)
            loader.regs_finish(regs, reg_off, enc, cum, idx_host_arrays.ref_0123, l_pac, contigs=contigs)
            done[k] += len(sub)

    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(work, range(threads)))
    dt = time.perf_counter() - t0
    n = sum(done)
    return n / dt / 1e6, dt, n


def dry_run(args, rank, world):
    """The launch / rendezvous / timing skeleton without a GPU (CPU-box rehearsal of --gpus N)."""
    import torch
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pass
    if world > 1:
        dist.barrier()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    if rank == 0:
        print(json.dumps({"metric": "dry run (no GPU work)", "value": 0.0, "unit": "Mreads/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": el / max(args.steps, 1) * 1e3, "higher_is_better": True,
                          "scaling": args.scaling, "vs_baseline": None, "dry_run": True}), flush=True)
    if world > 1:
        dist.destroy_process_group()



def load_pmc_summary(genome_mbp, reads):
    """counter-measured figures of the same workload from the committed PMC passes (separate rocprofv3 runs cannot be taken
    inside this one); the file is stamped with the commit and workload it was measured on"""
    tfile = PMC_SUMMARY
    try:
        tj = json.load(open(tfile))
        if abs(tj.get("genome_mbp", 0) - genome_mbp) < 1 and tj.get("reads") == reads:
            return tj
    except Exception:
        pass
    return None


def ert_report(st, mean, CHn, n_bases, ert_info, pmc=None):
    """stage times, roofline object and index facts of a run whose seeding went over the ERT (st: last chunk's stats)"""
    # algorithmic bytes of the walk kernel: 8 B per k-mer entry + one 32-B sector per tree record decoded + the .0123
    # bytes compared + the reads in + the L_m bytes out
    w_ms = mean("ms_smem_r1")
    w_bytes = 8 * st.ert_kmer_lookups + 32 * st.ert_node_reads + st.ert_ref_bytes + n_bases + 2 * n_bases
    stage = {"ert_walk": round(w_ms, 3), "ert_rounds": round(mean("ms_smem_r2"), 3), "ert_locate": round(mean("ms_smem_r3"), 3),
             "ert_locate_plus_hits": round(mean("ms_sal"), 3), "sort": round(mean("ms_sort"), 3), "seed_total": round(mean("ms_seed_total"), 3)}
    roof = {
        "kernel": "ert_profile_kernel (one forward ERT walk per read position)",
        "bound": "hbm", "achieved": round(w_bytes / (w_ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(w_bytes / (w_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
        "traffic": None if not pmc else pmc.get("ert_walk_hbm_bytes_per_launch"),
        "bytes_per_launch": int(w_bytes), "launch_ms": round(w_ms, 3),
        "random_reads_per_s_G": round((st.ert_kmer_lookups + st.ert_node_reads) / (w_ms * 1e-3) / 1e9, 2),
        "note": "random 8-B / 32-B reads: the distinct-line ceiling of this part is 48 G lines/s (tools/ubench_gather), i.e. 0.38 of the "
                "byte peak even for whole 64-B lines; random reads per second is the figure to read.  Round 4: the walk reads a k-mer's entry "
                "and the first 56 bytes of its tree from ONE line of a resident 64-byte-per-k-mer table derived from the two files' bytes "
                "(BWAMS_ERT_FAT=0: the two tables themselves) and writes one 24-byte record per read position; the algorithmic bytes are "
                "still the reference layout's",
    }
    info = {
        "kmer_size": ert_info["kmer"], "xmer_size": ert_info["xmer"], "read_len": ert_info["read_len"],
        "kmer_table_bytes": 8 * 4 ** ert_info["kmer"], "tree_bytes": ert_info["mlt_bytes"],
        "build_s": {"sizes": round(ert_info["build_ms"][0] / 1e3, 2), "bytes": round(ert_info["build_ms"][2] / 1e3, 2)},
        "events_per_read": {"kmer_lookups": round(st.ert_kmer_lookups / CHn, 2), "tree_records": round(st.ert_node_reads / CHn, 2),
                            "text_bytes_compared": round(st.ert_ref_bytes / CHn, 1)},
        "Gwalks_per_s": round(n_bases / (w_ms * 1e-3) / 1e9, 2),
        "note": "seeds and sampled hit positions are identical to the FM-index path's (tests/test_gpu_ert.py, tools/ert_scale.py)",
    }
    return stage, roof, info



def fastq_rows(rd, first):
    """One four-line FASTQ record per row: '@r%08d' names, the bases of `rd`, '+', a constant quality string."""
    n, RL = rd.shape
    row = np.empty((n, 1 + 9 + 1 + RL + 3 + RL + 1), np.uint8)
    row[:, 0] = ord("@"); row[:, 1] = ord("r")
    ids_ = first + np.arange(n, dtype=np.int64)
    for d_ in range(8):
        row[:, 2 + d_] = ord("0") + (ids_ // 10 ** (7 - d_)) % 10
    row[:, 10] = 10
    row[:, 11:11 + RL] = np.frombuffer(b"ACGTN", np.uint8)[rd]
    row[:, 11 + RL:14 + RL] = np.frombuffer(b"\n+\n", np.uint8)
    row[:, 14 + RL:14 + 2 * RL] = ord("I")
    row[:, 14 + 2 * RL] = 10
    return row


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if args.dry_run:
        return dry_run(args, rank, world)

    # the library forks its size classes onto auxiliary streams; HIP's default of four hardware queues makes some of them
    # share a queue and run one after the other (chaining 44 -> 40 ms with eight)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import torch
    import torch.distributed as dist

    from bwams import capi, shard, simulate

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
    G = int(round(args.genome_mbp * 1e6))
    genome = simulate.make_genome(G, seed=2024)
    contigs = simulate.chromosomes(G) if G >= 2 ** 31 else None          # bntann1_t.len is 32 bits: several sequences, as GRCh38
    cb = None if contigs is None else simulate.contig_bounds(contigs)
    log(f"genome {G/1e6:.0f} Mbp generated in {time.time()-t0:.1f}s")
    t0 = time.time()
    ix = capi.Index.build(genome, local)
    bst = ix.build_stats
    if contigs is not None:
        ix.set_contigs(contigs)
    log(f"FM-index built on GPU in {time.time()-t0:.1f}s: {bst.rows/1e9:.2f} G rows, {bst.chunks} chunks + {bst.rounds} doubling rounds, "
        f"{ix.nbytes/2**30:.1f} GiB resident")
    if args.fma:
        t0 = time.time()
        ix.build_fma(11, 13)
        log(f"FMA tables built on GPU in {time.time()-t0:.1f}s (512 MiB + 1 GiB)")

    emf_h = None
    emf_info = None
    if args.emf:
        t0 = time.time()
        emf_h = capi.Emf.build(ix, seed_len=150, slack=1.1)       # `perfect-index -l 150`: num_seed_entry = 1.1 x l_pac
        emf_info = emf_h.info()
        log(f"EMF table built on GPU in {time.time()-t0:.1f}s: {emf_info['num_seed_entry']/1e6:.0f} M entries "
            f"({emf_info['num_seed_entry']*16/2**30:.1f} GiB), {emf_info['n_used']/1e6:.0f} M distinct 150-mers in {emf_info['n_key']/1e6:.0f} M buckets")

    ert_h = None
    ert_info = None
    if args.ert:
        t0 = time.time()
        ert_h = capi.Ert.build(ix)                 # kmerSize 15, xmerSize 4, READ_LEN 151, HIT_THRESHOLD 256 (src/macro.h)
        ert_info = ert_h.info()
        log(f"ERT built on GPU in {time.time()-t0:.1f}s: k-mer table 8 GiB + trees {ert_info['mlt_bytes']/2**30:.1f} GiB "
            f"(kernels: sizes {ert_info['build_ms'][0]/1e3:.1f}s, bytes {ert_info['build_ms'][2]/1e3:.1f}s)")

    if args.scaling == "weak":
        R = args.reads
        first = rank * R
    else:
        b = shard.shard_bounds(args.reads_total, world)
        first, R = int(b[rank]), int(b[rank + 1] - b[rank])
    total_reads = R * world if args.scaling == "weak" else args.reads_total
    t0 = time.time()
    # a rank's reads are a function of its shard only (seed = 12345 + first read / chunk): a strong-scaling job maps the
    # same chunks onto fewer or more ranks
    CH = max(1, min(args.chunk_reads, R))
    n_chunks = (R + CH - 1) // CH
    chunk_sizes = [min(CH, R - c * CH) for c in range(n_chunks)]
    reads_l = [simulate.make_reads(genome, n, seed=12345 + (first + c * CH) // CH if args.scaling == "strong" else 12345 + rank * 1000 + c,
                                   contig_bounds=cb)[0] for c, n in enumerate(chunk_sizes)]
    reads = reads_l[0]
    RL = reads.shape[1]
    d_reads = [torch.from_numpy(r.reshape(-1)).to(dev) for r in reads_l]          # resident input: the clock starts with these in HBM
    cums = [np.arange(len(r) + 1, dtype=np.int64) * RL for r in reads_l]
    log(f"{R} reads generated in {time.time()-t0:.1f}s ({n_chunks} resident chunk(s) of <= {CH})")

    batch = capi.Batch(ix, CH, CH * RL, max_smem=32 * CH, max_sa=128 * CH)
    seed_opt = capi.default_seed_opt()
    mem_opt = capi.default_mem_opt()
    agg = {}

    def run_chunk(c):
        # reads (HBM) -> [EMF] -> seeds -> chains -> extension tasks -> banded SW (left, right, retries) -> regions -> dedup
        batch.seed_upload_device(d_reads[c].data_ptr(), cums[c])         # device-to-device copy + 2-bit packing
        if emf_h is not None:
            batch.emf_run(emf_h)                                          # probe; the matched reads skip seeding ...
            n_emf = capi.C.c_int64(0)                                     # ... and get their regions directly (mem_perfect2reg)
            capi._chk(capi.lib().bwams_emf_regs_run(batch.h, emf_h.h, capi.C.byref(mem_opt), capi.C.byref(n_emf)), "bwams_emf_regs_run")
        if ert_h is not None:
            batch.seed_run_ert(ert_h, seed_opt, with_sa=True)
        else:
            batch.seed_run(seed_opt, with_sa=True)
        batch.chain_run(mem_opt)
        batch.extend_run(mem_opt)
        batch.dedup_run(mem_opt)

    def step(collect=None):
        for c in range(n_chunks):
            run_chunk(c)
            st_ = batch.stats()                 # synchronises the batch stream; part of the timed region
            if collect is not None:
                collect.append(st_)

    for _ in range(args.warmup):
        step()
    batch.sync()

    # ---------------- timed region ----------------
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    per_chunk = []
    for _ in range(args.steps):
        step(per_chunk)
    batch.sync()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t_start
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---------------- the same steps with seeding over the ERT (reported beside, never `value`; N = 1 only) ----------------
    ert_side = None
    if not args.ert and not args.no_ert_leg and world == 1 and not (args.fma or args.emf):
        t0 = time.time()
        ert_h = capi.Ert.build(ix)
        ert_info = ert_h.info()
        log(f"ERT built on GPU in {time.time()-t0:.1f}s: k-mer table 8 GiB + trees {ert_info['mlt_bytes']/2**30:.1f} GiB")
        step()
        batch.sync()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pc_e = []
        for _ in range(args.steps):
            step(pc_e)
        batch.sync()
        torch.cuda.synchronize()
        el_e = time.perf_counter() - t0
        mean_e = lambda f: float(np.mean([getattr(s_, f) for s_ in pc_e]))       # noqa: E731
        stg, roof, einfo = ert_report(pc_e[-1], mean_e, len(reads_l[-1]), int(cums[-1][-1]), ert_info,
                                      load_pmc_summary(args.genome_mbp, len(reads_l[-1])))
        ert_side = {"value": round(total_reads * args.steps / el_e / 1e6, 4), "unit": "Mreads/s", "ms_per_step": round(el_e / args.steps * 1e3, 3),
                    "stage_ms": stg, "roofline": roof, "index": einfo,
                    "final_regions": int(pc_e[-1].n_final_regs),
                    "note": "configs[2]'s seeding backend on the same reads: the ERT (k-mer table + radix trees, built on the GPU from the "
                            "resident FM-index) replaces SMEM search + SA lookup; chaining, extension and dedup unchanged; same final regions"}
        # ... and with the exact-match filter in front of it: configs[2]'s index set (ERT + EMF) resident.  The walk's entry + tree-head table
        # (64 GiB) is given back first: with the EMF beside it the GPU's memory is better spent on the chunks in flight (with 58 GiB free
        # the streaming leg below ran at 4.4 .. 5.6 Mreads/s and three chunks in flight did not finish; profiles/r04_notes.md)
        ert_h.set_fat(False)
        t0 = time.time()
        emf_h = capi.Emf.build(ix, seed_len=150, slack=1.1)
        emf_info = emf_h.info()
        log(f"EMF table built on GPU in {time.time()-t0:.1f}s: {emf_info['num_seed_entry']*16/2**30:.1f} GiB")
        step()
        batch.sync()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pc_c = []
        for _ in range(args.steps):
            step(pc_c)
        batch.sync()
        torch.cuda.synchronize()
        el_c = time.perf_counter() - t0
        mean_c = lambda f: float(np.mean([getattr(s_, f) for s_ in pc_c]))       # noqa: E731
        _, codes_c = batch.emf_fetch(len(reads_l[-1]))
        ert_side["with_emf"] = {
            "value": round(total_reads * args.steps / el_c / 1e6, 4), "unit": "Mreads/s", "ms_per_step": round(el_c / args.steps * 1e3, 3),
            "stage_ms": {"emf": round(mean_c("ms_emf"), 3), "seed_total": round(mean_c("ms_seed_total"), 3), "chain": round(mean_c("ms_chain"), 3),
                         "ext_total": round(mean_c("ms_ext_total"), 3), "dedup": round(mean_c("ms_dedup"), 3)},
            "emf": {"table_bytes": emf_info["num_seed_entry"] * 16, "distinct_lmers": emf_info["n_used"], "build_s": round(emf_info["build_ms"] / 1e3, 2),
                    "resolved_fraction": round(float(((codes_c == 3) | (codes_c == 4)).mean()), 4)},
            "note": "EMF probe first; the reads it resolves skip seeding and get their regions from mem_perfect2reg (bwams_emf_regs_run), inside the step",
        }
        # configs[2] text to text: bwams_process_chunk with both handles (exact-match records through put_perfect)
        ix.set_contig_names([b"chr%d" % (i + 1) for i in range(len(contigs) if contigs is not None else 1)])
        row_c = fastq_rows(reads_l[n_chunks - 1], first)
        d_fq_c = torch.from_numpy(row_c.reshape(-1)).to(dev)
        pc_args = dict(emf=emf_h, ert=ert_h, seed_opt=seed_opt, opt=mem_opt, sopt=capi.default_sam_opt(), n_processed=first, fetch=False)
        batch.process_chunk((d_fq_c.data_ptr(), row_c.size), **pc_args)
        batch.sync()
        t0 = time.perf_counter()
        for _ in range(2):
            bytes_c = batch.process_chunk((d_fq_c.data_ptr(), row_c.size), **pc_args)
        batch.sync()
        ms_c = (time.perf_counter() - t0) / 2 * 1e3
        ert_side["with_emf"]["fastq_to_sam"] = {
            "ms_per_chunk": round(ms_c, 2), "Mreads_per_s": round(len(row_c) / (ms_c * 1e-3) / 1e6, 3), "sam_bytes": int(bytes_c),
            "note": "bwams_process_chunk with the EMF and the ERT (configs[2]'s index set), single-end, FASTQ text in HBM -> SAM text in HBM"}
        del row_c, d_fq_c
        log(f"configs[2] index set: step {ert_side['with_emf']['ms_per_step']} ms, text to text {ms_c:.1f} ms per chunk")
        # configs[2] as a JOB: chunks streamed through the compiled mem_process_seqs() with three chunks in flight (the reader's
        # thread stages chunk i + 1, the writer's collects chunk i - 1: kt_pipeline's three steps, bwams/stream.py) — records in host
        # memory in, SAM strings in host memory out, i.e. host staging and PCIe inside the clock
        try:
            from bwams import stream
            n_job = int(os.environ.get("BWAMS_STREAM_CHUNKS", "6"))
            rd_job = reads_l[n_chunks - 1]
            o_t = capi.mem_opt_init(False)
            res_j = {}
            depths_ = [int(x) for x in os.environ.get("BWAMS_STREAM_DEPTHS", "2,3,1").split(",")]
            if torch.cuda.mem_get_info()[0] < 100 * 2 ** 30:           # three chunks in flight want three batches of buffers
                depths_ = [d_ for d_ in depths_ if d_ != 3]
            for depth_ in depths_:
                pre = [capi.Seqs(rd_job, first_id=first + k * len(rd_job)) for k in range(n_job + 1)]      # step 0's output, parsed beforehand
                wk = capi.Worker([ix], len(rd_job), len(rd_job) * RL, emfs=[emf_h], erts=[ert_h], depth=depth_)
                stream.run_job(wk, o_t, lambda k: pre[n_job], 1, None, n_processed0=first)                  # warm-up: buffers, first touch
                nj_ = n_job if depth_ > 1 else 2
                secs_j, n_j = stream.run_job(wk, o_t, lambda k: pre[k], nj_, None, n_processed0=first, overlap=depth_ > 1)
                wk.close()
                del pre
                res_j[depth_] = n_j / secs_j / 1e6
                log(f"configs2_stream: {depth_} chunk(s) in flight, {nj_} chunks: {res_j[depth_]:.3f} Mreads/s")
            resident_ = len(rd_job) / (ms_c * 1e-3) / 1e6
            ert_side["with_emf"]["configs2_stream"] = {
                "Mreads_per_s": round(res_j[2], 3), "chunks": n_job, "reads_per_chunk": len(rd_job), "chunks_in_flight": 2,
                "ms_per_chunk": round(len(rd_job) / res_j[2] / 1e3, 2), "ratio_to_resident": round(res_j[2] / resident_, 3),
                "three_in_flight_Mreads_per_s": round(res_j[3], 3) if 3 in res_j else None, "no_overlap_Mreads_per_s": round(res_j[1], 3) if 1 in res_j else None,
                "note": "BASELINE configs[2] as a job through the compiled mem_process_seqs() (host/mem_process_seqs_hip.cpp): parsed records "
                        "(bseq1_t) in host memory -> page-locked arrays -> GPU -> one malloc'ed SAM string per 512-read work item in host "
                        "memory, two chunks in flight (stage on the reader's thread, collect on the writer's); ratio_to_resident compares "
                        "with bwams_process_chunk on text already in HBM; no_overlap = mem_process_seqs alone, one chunk at a time"}
        except Exception as e_:                       # the leg is reported beside: a failure is recorded, not hidden
            ert_side["with_emf"]["configs2_stream"] = {"error": repr(e_)}
        emf_h.close()
        emf_h = None
        ert_h.close()
        ert_h = None
        step()                                  # leave the batch as the timed configuration left it (the legs below read it)
        batch.sync()

    # ---------------- SAM-side alignment of the last chunk's final regions (reported beside, never `value`) ----------------
    batch.mark_primary_se(mem_opt, id_base=first)
    batch.sync()
    t0 = time.perf_counter()
    batch.mark_primary_se(mem_opt, id_base=first)           # mem_mark_primary_se of every read (mem_reg2sam's first step)
    batch.sync()
    mark_ms = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    aln_, cig_, md_ = batch.reg2aln(mem_opt, 1)
    reg2aln_ms = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    n_aln = capi.C.c_int64(0); n_cig = capi.C.c_int64(0); n_md = capi.C.c_int64(0)
    capi._chk(capi.lib().bwams_reg2aln_run(batch.h, capi.C.byref(mem_opt), 1, capi.C.byref(n_aln), capi.C.byref(n_cig), capi.C.byref(n_md)), "bwams_reg2aln_run")
    reg2aln_run_ms = (time.perf_counter() - t0) * 1e3
    sam_side = {"regions": int(n_aln.value), "cigar_ops": int(n_cig.value), "md_bytes": int(n_md.value),
                "gapped_fraction": round(float(np.mean([(int(c) & 0xf) in (1, 2) for c in cig_[:200000]])), 4) if len(cig_) else 0.0,
                "ms_mark_primary_se": round(mark_ms, 2), "mean_mapq": round(float(aln_["mapq"].mean()), 2) if len(aln_) else None,
                "ms_run": round(reg2aln_run_ms, 2), "ms_run_plus_fetch": round(reg2aln_ms, 2),
                "Malignments_per_s": round(n_aln.value / (reg2aln_run_ms * 1e-3) / 1e6, 2) if reg2aln_run_ms > 0 else None,
                "note": "single-end SAM side of one chunk on the device: mem_mark_primary_se of every read, then mem_reg2aln (band inference, "
                        "banded global alignment with traceback, CIGAR / NM / MD, position) of every final region; mapping quality on the host "
                        "side of the library"}
    # mem_reg2aln restricted to what the SAM text reads (the reference aligns only what it prints)
    sopt_ = capi.default_sam_opt()
    batch.reg2aln_sam(mem_opt, sopt_, fetch=False)
    t0 = time.perf_counter()
    _, n_needed_ = batch.reg2aln_sam(mem_opt, sopt_, fetch=False)
    reg2aln_sam_ms = (time.perf_counter() - t0) * 1e3
    aln_r_ = batch.reg2aln_sam(mem_opt, sopt_)[0]
    done_ = aln_r_["rid"] >= 0
    sam_side["reg2aln_for_sam"] = {"regions_aligned": int(n_needed_), "of": int(n_aln.value), "ms_run": round(reg2aln_sam_ms, 2),
                                   "note": "bwams_reg2aln_run_sam: the records mem_reg2sam prints and the members of their XA strings only; "
                                           "the SAM text below is produced from this run"}
    # the SAM text of the chunk (mem_reg2sam + mem_gen_alt + mem_aln2sam on the device), with size-independent checks
    cum_ = cums[n_chunks - 1]
    n_seq_ = len(cum_) - 1
    names_ = [b"r%08d" % (first + i) for i in range(n_seq_)]          # the names the FASTQ text below carries
    quals_ = np.full(int(cum_[-1]), ord("I"), np.uint8)
    ix.set_contig_names([b"chr%d" % (i + 1) for i in range(len(contigs) if contigs is not None else 1)])
    t0 = time.perf_counter()
    batch.sam_upload(names_, quals_)
    sam_up_ms = (time.perf_counter() - t0) * 1e3
    batch.sam_run(mem_opt, sopt_)
    t0 = time.perf_counter()
    sam_bytes = batch.sam_run(mem_opt, sopt_)
    sam_ms = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    text_, roff_, mq_ = batch.sam_fetch(len(aln_))
    sam_fetch_ms = (time.perf_counter() - t0) * 1e3
    tb = np.frombuffer(text_, np.uint8)
    n_lines = int((tb == 10).sum())
    n_tabs = int((tb == 9).sum())
    checks = {
        "device_mapq_equals_host_mapq": bool(np.array_equal(mq_[done_], aln_["mapq"][done_])),
        "restricted_records_equal_full_run": bool(all(np.array_equal(aln_r_[f_][done_], aln_[f_][done_]) for f_ in ("pos", "rid", "NM", "n_cigar", "md_len", "score"))
                                                  and int(done_.sum()) == int(n_needed_)),
        "one_block_per_read": bool(roff_[0] == 0 and roff_[-1] == sam_bytes and np.all(np.diff(roff_) > 0)),
        "every_block_ends_a_line": bool(np.all(tb[roff_[1:] - 1] == 10)),
        "lines_at_least_reads": n_lines >= n_seq_,
        "eleven_fields_per_line": n_tabs >= 10 * n_lines,
    }
    sam_side["sam_text"] = {"bytes": int(sam_bytes), "lines": n_lines, "ms_run": round(sam_ms, 2), "ms_upload_names_quals": round(sam_up_ms, 2),
                            "ms_fetch": round(sam_fetch_ms, 2), "Mreads_per_s": round(n_seq_ / (sam_ms * 1e-3) / 1e6, 2) if sam_ms > 0 else None,
                            "GBps_written": round(sam_bytes / (sam_ms * 1e-3) / 1e9, 2) if sam_ms > 0 else None,
                            "xa_tags": int(text_.count(b"\tXA:Z:")), "unaligned_records": int(text_.count(b"\t4\t*\t0\t0\t*\t")),
                            "checks": checks,
                            "note": "mem_reg2sam + mem_gen_alt + mem_aln2sam (single-end) and mem_approx_mapq_se on the device, names "
                                    "and qualities already in HBM when timed; byte-equal to the oracle in tests/test_gpu_sam.py"}
    if not all(checks.values()):
        raise SystemExit(f"[bench] SAM text property check failed: {checks}")
    # read input: the chunk as FASTQ text -> encoded bases, names, qualities on the device (kseq_read + trim_readno + base encoding)
    rd_ = reads_l[n_chunks - 1]
    row = fastq_rows(rd_, first)
    fq_text = row.tobytes()
    d_fq = torch.from_numpy(row.reshape(-1)).to(dev)               # the text resident in HBM, as the other inputs are
    fq = capi.Fastq(d_fq.data_ptr(), device=local, n_bytes=len(fq_text)); fq.close()
    t0 = time.perf_counter()
    fq = capi.Fastq(d_fq.data_ptr(), device=local, n_bytes=len(fq_text))
    fq_wall_ms = (time.perf_counter() - t0) * 1e3
    fqi = fq.info()
    got_ = fq.fetch()
    fq_checks = {"bases_equal_the_reads": bool(np.array_equal(got_["enc"], rd_.reshape(-1))),
                 "cum_is_regular": bool(np.array_equal(got_["cum"], cum_)),
                 "names_round_trip": bool(got_["names"][0] == b"r%08d" % first and got_["names"][-1] == b"r%08d" % (first + n_seq_ - 1)),
                 "qualities_round_trip": bool((got_["quals"] == ord("I")).all())}
    fq.close()
    sam_side["fastq_decode"] = {"bytes": len(fq_text), "reads": int(fqi["n_reads"]), "ms_kernels": round(fqi["ms"], 3), "ms_call": round(fq_wall_ms, 2),
                                "GBps_text": round(len(fq_text) / (fqi["ms"] * 1e-3) / 1e9, 1) if fqi["ms"] > 0 else None,
                                "Mreads_per_s": round(fqi["n_reads"] / (fqi["ms"] * 1e-3) / 1e6, 1) if fqi["ms"] > 0 else None,
                                "checks": fq_checks,
                                "note": "four-line FASTQ text already in HBM -> encoded bases + cum, names (trim_readno), comments, qualities: line "
                                        "ends (count + rocPRIM select), a lane per record to validate and measure, a wave per record to copy / encode; "
                                        "ms_call adds the allocations, three scans and the offset arrays' copy to the host"}
    if not all(fq_checks.values()):
        raise SystemExit(f"[bench] FASTQ decode property check failed: {fq_checks}")
    # the whole single-end path in one go, text to text: FASTQ bytes (in HBM) -> ... -> SAM bytes (in HBM)
    def fastq_to_sam():                                  # bwams_process_chunk: the outer boundary, one call
        return batch.process_chunk((d_fq.data_ptr(), len(fq_text)), seed_opt=seed_opt, opt=mem_opt, sopt=sopt_, n_processed=first, fetch=False)
    fastq_to_sam()
    batch.sync()
    t0 = time.perf_counter()
    for _ in range(2):
        e2e_bytes = fastq_to_sam()
    batch.sync()
    e2e_ms = (time.perf_counter() - t0) / 2 * 1e3
    sam_side["fastq_to_sam"] = {"ms_per_chunk": round(e2e_ms, 2), "Mreads_per_s": round(n_seq_ / (e2e_ms * 1e-3) / 1e6, 3), "sam_bytes": int(e2e_bytes),
                                "note": "bwams_process_chunk — single-end, FM-index seeding, one chunk end to end on the device: FASTQ decode, seed -> chain -> extend -> dedup, "
                                        "mem_mark_primary_se, mem_reg2aln of what the text needs, SAM text; FASTQ text resident in HBM at the start, "
                                        "SAM text left in HBM at the end (host transfers and gz excluded); never `value`"}
    if e2e_bytes != sam_bytes:
        raise SystemExit(f"[bench] FASTQ -> SAM run produced {e2e_bytes} bytes of text, the staged run {sam_bytes}")
    # the same call with the text in pageable host memory on both sides (what a caller without device buffers pays): never `value`
    sam_h = np.empty(int(sam_bytes) + 16, np.uint8)
    off_h = np.empty(n_seq_ + 1, np.int64)
    t0 = time.perf_counter()
    hb_ = batch.process_chunk(fq_text, seed_opt=seed_opt, opt=mem_opt, sopt=sopt_, n_processed=first, fetch=False)
    capi._chk(capi.lib().bwams_sam_fetch(batch.h, capi._p(sam_h), len(sam_h), capi._p(off_h), None, 0), "bwams_sam_fetch")
    pcie_ms = (time.perf_counter() - t0) * 1e3
    sam_side["fastq_to_sam"]["pcie_inclusive"] = {
        "ms_per_chunk": round(pcie_ms, 2), "Mreads_per_s": round(n_seq_ / (pcie_ms * 1e-3) / 1e6, 3),
        "host_bytes_up": len(fq_text), "host_bytes_down": int(hb_) + 8 * (n_seq_ + 1),
        "note": "FASTQ text in pageable host memory -> SAM text and its per-read offsets in pageable host memory, one call + bwams_sam_fetch"}
    del sam_h, off_h
    # ... and in PAGE-LOCKED host memory (bwams_host_alloc), which is what the compiled caller at the reference's boundary stages
    # through (bwa-mem-scale_amd/host/mem_process_seqs_hip.cpp): never `value`
    fq_pin = capi.pinned_array(len(fq_text))
    fq_pin[:] = np.frombuffer(fq_text, np.uint8)
    sam_pin = capi.pinned_array(int(sam_bytes) + 16)
    off_pin = capi.pinned_array(n_seq_ + 1, np.int64)
    pin_ms = []
    for _ in range(2):
        t0 = time.perf_counter()
        hb_ = batch.process_chunk((fq_pin.ctypes.data, len(fq_text)), seed_opt=seed_opt, opt=mem_opt, sopt=sopt_, n_processed=first, fetch=False)
        capi._chk(capi.lib().bwams_sam_fetch(batch.h, capi._p(sam_pin), len(sam_pin), capi._p(off_pin), None, 0), "bwams_sam_fetch")
        pin_ms.append((time.perf_counter() - t0) * 1e3)
    sam_side["fastq_to_sam"]["pcie_inclusive_pinned"] = {
        "ms_per_chunk": round(min(pin_ms), 2), "Mreads_per_s": round(n_seq_ / (min(pin_ms) * 1e-3) / 1e6, 3),
        "ratio_to_resident": round(min(pin_ms) / e2e_ms, 3),
        "note": "the same call with both texts in page-locked host memory (bwams_host_alloc): what the compiled mem_process_seqs() pays per chunk"}
    # ... and the chunk over TWO batches on this GPU behind one call (host/chunk_multi.cpp: the N-GPU form of the outer boundary with
    # both shards on one device): parsed records in page-locked memory -> SAM text in page-locked memory, in read order
    try:
        b2 = [capi.Batch(ix, n_seq_, n_seq_ * RL) for _ in range(2)]
        multi = capi.Multi(b2)
        nm_pin = capi.pinned_array(9 * n_seq_)
        nm_pin[:] = np.frombuffer(b"".join(b"r%08d" % (first + i) for i in range(n_seq_)), np.uint8)
        noff_ = (np.arange(n_seq_ + 1, dtype=np.int64) * 9)
        enc_pin = capi.pinned_array(n_seq_ * RL)
        enc_pin[:] = rd_.reshape(-1)
        q_pin = capi.pinned_array(n_seq_ * RL)
        q_pin[:] = ord("I")
        mt = []
        for _ in range(2):
            t0 = time.perf_counter()
            mtext, moff = multi.process_reads(enc_pin, cum_, nm_pin, noff_, quals=q_pin, seed_opt=seed_opt, opt=mem_opt, sopt=sopt_, n_processed=first)
            mt.append((time.perf_counter() - t0) * 1e3)
        sam_side["fastq_to_sam"]["two_batches_one_call"] = {
            "ms_per_chunk": round(min(mt), 2), "Mreads_per_s": round(n_seq_ / (min(mt) * 1e-3) / 1e6, 3), "same_text": bool(len(mtext) == sam_bytes),
            "note": "bwams_multi_process_reads + bwams_multi_fetch with 2 batches on ONE GPU (parsed records up, text down, page-locked; the fetch "
                    "lands in a pageable numpy buffer here): the code path `N GPUs behind one C call` takes, byte-identical to one batch "
                    "(tests/test_host_boundary.py); no N > 1 hardware run exists for it"}
        multi.close()
        for b_ in b2:
            b_.close()
        for a_ in (nm_pin, enc_pin, q_pin):
            capi.pinned_free(a_)
        del b2, multi, nm_pin, enc_pin, q_pin, mtext, moff
    except capi.BwamsError as e:
        sam_side["fastq_to_sam"]["two_batches_one_call"] = {"error": str(e)}
    for a_ in (fq_pin, sam_pin, off_pin):
        capi.pinned_free(a_)
    del fq_pin, sam_pin, off_pin
    del row, fq_text, d_fq, got_
    del aln_, cig_, md_, text_, tb, mq_, aln_r_

    # ---------------- paired-end leg (every rank: the pestat exchange is a collective) ----------------
    pe_out = None
    if not args.no_pe:
        t0 = time.time()
        n_pairs = chunk_sizes[0] // 2
        pr = simulate.make_read_pairs_bulk(genome, n_pairs, seed=4242 + rank, contig_bounds=cb)
        penc, pcum = simulate.flatten_reads(pr)
        log(f"{n_pairs} read pairs generated in {time.time()-t0:.1f}s")
        batch.seed_upload(penc, pcum)

        def pe_step():
            if ert_h is not None:
                batch.seed_run_ert(ert_h, seed_opt, with_sa=True)
            else:
                batch.seed_run(seed_opt, with_sa=True)
            batch.chain_run(mem_opt)
            batch.extend_run(mem_opt)
            batch.dedup_run(mem_opt)
            # mem_pestat is a statistic of the whole chunk: sharded pairs all-gather their 8-byte keys (RCCL), the one
            # exchange step of the paired-end path
            pes_ = shard.pestat_sharded(batch.pestat_keys(mem_opt), dist if world > 1 else None)
            n_, nt_ = batch.pair_run(pes_, mem_opt, id_base=rank * n_pairs)
            return pes_, n_, nt_

        pe_step()
        batch.sync()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(2):
            pes_, n_pe, nt_pe = pe_step()
        batch.sync()
        if world > 1:
            dist.barrier()
        dt = (time.perf_counter() - t0) / 2
        if world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        pst = batch.stats()
        _, _, prs = batch.pair_fetch()
        pe_out = {
            "value": round(2 * n_pairs * world / dt / 1e6, 4), "unit": "Mreads/s", "ms_per_batch": round(dt * 1e3, 2),
            "pairs_per_gpu": n_pairs, "n_gpus": world,
            "ms_pestat_plus_pair": round(dt * 1e3 - float(pst.ms_seed_total + pst.ms_chain + pst.ms_ext_total + pst.ms_dedup), 2),
            "stage_ms": {"seed_total": round(float(pst.ms_seed_total), 2), "chain": round(float(pst.ms_chain), 2), "ext_total": round(float(pst.ms_ext_total), 2), "dedup": round(float(pst.ms_dedup), 2)},
            "seeding": "ert" if ert_h is not None else "fm-index",
            "ms_pair_run": round(float(pst.ms_pair), 3), "rescue_alignments": int(nt_pe), "reads_redone": int(pst.n_pair_redone),
            "regions_after_rescue": int(n_pe), "proper_pairs": round(float((prs["score"] > 0).mean()), 4),
            "orientations_failed": [int(x) for x in pes_["failed"]], "insert_avg_std": [round(float(pes_["avg"][1]), 2), round(float(pes_["std"][1]), 2)],
            "note": "2x150bp FR pairs (insert 400 +- 40, 5 % with a damaged end, 2 % discordant): SE step + mem_pestat (keys all-gathered over "
                    "RCCL when sharded) + mate rescue (ksw_align2 on the GPU) + mem_mark_primary_se + mem_pair; regions and pairing decisions "
                    "stay on the device; rank 0's counts",
        }
        if rank == 0:
            # the paired-end SAM side beside: mem_reg2aln of the regions after rescue, then mem_sam_pe's text (never part of the timed batches)
            batch.reg2aln_sam(mem_opt, capi.default_sam_opt(), pes=pes_, fetch=False)
            t0 = time.perf_counter()
            pe_n_aln, pe_n_need = batch.reg2aln_sam(mem_opt, capi.default_sam_opt(), pes=pes_, fetch=False)    # only what mem_sam_pe reads
            pe_aln_ms = (time.perf_counter() - t0) * 1e3
            pnames = [b"pair%d" % (rank * n_pairs + i // 2) for i in range(2 * n_pairs)]
            batch.sam_upload(pnames, np.full(int(pcum[-1]), ord("I"), np.uint8))
            batch.sam_run_pe(pes_, mem_opt, capi.default_sam_opt())
            t0 = time.perf_counter()
            pe_sam_bytes = batch.sam_run_pe(pes_, mem_opt, capi.default_sam_opt())
            pe_sam_ms = (time.perf_counter() - t0) * 1e3
            ptext, proff, _ = batch.sam_fetch()
            ptb = np.frombuffer(ptext, np.uint8)
            first_flags = [int(ptext[proff[r]:proff[r + 1]].split(b"\t", 2)[1]) for r in range(0, min(2 * n_pairs, 20000))]
            pe_checks = {"one_block_per_read": bool(proff[-1] == pe_sam_bytes and np.all(np.diff(proff) > 0)),
                         "every_block_ends_a_line": bool(np.all(ptb[proff[1:] - 1] == 10)),
                         "first_in_pair_then_second": all((f & 0x41) == 0x41 if r % 2 == 0 else (f & 0x81) == 0x81 for r, f in enumerate(first_flags)),
                         "proper_flag_symmetric": all((first_flags[r] & 2) == (first_flags[r + 1] & 2) for r in range(0, len(first_flags) - 1, 2))}
            pe_out["sam_text"] = {"bytes": int(pe_sam_bytes), "lines": int((ptb == 10).sum()), "ms_reg2aln": round(pe_aln_ms, 2),
                                  "regions_aligned": int(pe_n_need), "regions": int(pe_n_aln), "ms_run": round(pe_sam_ms, 2),
                                  "Mreads_per_s": round(2 * n_pairs / (pe_sam_ms * 1e-3) / 1e6, 2) if pe_sam_ms > 0 else None,
                                  "proper_pair_records": int(sum(1 for f in first_flags if f & 2)), "checks": pe_checks,
                                  "note": "mem_sam_pe from mem_pair's result on (q_pe / q_se, region edits, mate fields, MC / XA / SA) on the device; "
                                          "byte-equal to the oracle in tests/test_gpu_sam.py; first 20 000 reads' flags checked here"}
            if not all(pe_checks.values()):
                raise SystemExit(f"[bench] paired-end SAM text property check failed: {pe_checks}")
            del ptext, ptb
            # the whole paired-end path in one call, text to text (bwams_process_chunk): the two ends of a pair interleaved
            PRL = pr.shape[-1]
            prd = np.asarray(pr).reshape(-1, PRL)
            prow = np.empty((2 * n_pairs, 1 + 9 + 2 + 1 + PRL + 3 + PRL + 1), np.uint8)
            prow[:, 0] = ord("@"); prow[:, 1] = ord("p")
            pids = (np.arange(2 * n_pairs, dtype=np.int64) >> 1) + rank * n_pairs
            for d_ in range(8):
                prow[:, 2 + d_] = ord("0") + (pids // 10 ** (7 - d_)) % 10
            prow[:, 10] = ord("/"); prow[:, 11] = ord("1") + (np.arange(2 * n_pairs) & 1)
            prow[:, 12] = 10
            prow[:, 13:13 + PRL] = np.frombuffer(b"ACGTN", np.uint8)[prd]
            prow[:, 13 + PRL:16 + PRL] = np.frombuffer(b"\n+\n", np.uint8)
            prow[:, 16 + PRL:16 + 2 * PRL] = ord("I")
            prow[:, 16 + 2 * PRL] = 10
            d_pfq = torch.from_numpy(prow.reshape(-1)).to(dev)
            pe_args = dict(paired=True, seed_opt=seed_opt, opt=mem_opt, sopt=capi.default_sam_opt(), n_processed=2 * rank * n_pairs, fetch=False)
            batch.process_chunk((d_pfq.data_ptr(), prow.size), **pe_args)
            batch.sync()
            t0 = time.perf_counter()
            for _ in range(2):
                pe_e2e_bytes = batch.process_chunk((d_pfq.data_ptr(), prow.size), **pe_args)
            batch.sync()
            pe_e2e_ms = (time.perf_counter() - t0) / 2 * 1e3
            pe_out["fastq_to_sam"] = {"ms_per_chunk": round(pe_e2e_ms, 2), "Mreads_per_s": round(2 * n_pairs / (pe_e2e_ms * 1e-3) / 1e6, 3), "sam_bytes": int(pe_e2e_bytes),
                                      "note": "bwams_process_chunk, paired-end, one GPU: interleaved FASTQ text in HBM -> SAM text in HBM (decode, seed .. dedup, "
                                              "mem_pestat inferred from the chunk, mate rescue, pairing, mem_reg2aln of what is printed, mem_sam_pe's text)"}
            del prow, d_pfq

    # ---------------- the same steps on the harder genome (reported beside, never `value`; N = 1 only) ----------------
    hard_side = None
    if world == 1 and not args.no_hard_genome and not (args.fma or args.emf or args.ert):
        t0 = time.time()
        genome_h = simulate.make_genome(G, seed=2024, profile="grch38_like")
        ix_h = capi.Index.build(genome_h, local)
        bst_h = ix_h.build_stats
        if contigs is not None:
            ix_h.set_contigs(contigs)
        reads_h = simulate.make_reads(genome_h, len(reads_l[0]), seed=12345 + rank * 1000, contig_bounds=cb)[0]
        d_reads_h = torch.from_numpy(reads_h.reshape(-1)).to(dev)
        cum_h = np.arange(len(reads_h) + 1, dtype=np.int64) * RL
        batch_h = capi.Batch(ix_h, len(reads_h), len(reads_h) * RL, max_smem=32 * len(reads_h), max_sa=128 * len(reads_h))
        log(f"grch38_like genome, index ({bst_h.rounds} doubling rounds) and reads in {time.time()-t0:.1f}s")

        def step_h(collect=None):
            batch_h.seed_upload_device(d_reads_h.data_ptr(), cum_h)
            batch_h.seed_run(seed_opt, with_sa=True)
            batch_h.chain_run(mem_opt)
            batch_h.extend_run(mem_opt)
            batch_h.dedup_run(mem_opt)
            st_ = batch_h.stats()
            if collect is not None:
                collect.append(st_)

        step_h()
        batch_h.sync()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pc_h = []
        for _ in range(args.steps):
            step_h(pc_h)
        batch_h.sync()
        torch.cuda.synchronize()
        el_h = time.perf_counter() - t0
        mean_h = lambda f: float(np.mean([getattr(s_, f) for s_ in pc_h]))       # noqa: E731
        sth = pc_h[-1]
        hard_side = {
            "profile": "grch38_like", "value": round(len(reads_h) * args.steps / el_h / 1e6, 4), "unit": "Mreads/s",
            "ms_per_step": round(el_h / args.steps * 1e3, 3),
            "stage_ms": {"smem_round1": round(mean_h("ms_smem_r1"), 3), "smem_round2": round(mean_h("ms_smem_r2"), 3), "smem_round3": round(mean_h("ms_smem_r3"), 3),
                         "sa_lookup": round(mean_h("ms_sal"), 3), "seed_total": round(mean_h("ms_seed_total"), 3), "chain": round(mean_h("ms_chain"), 3),
                         "ext_total": round(mean_h("ms_ext_total"), 3), "dedup": round(mean_h("ms_dedup"), 3)},
            "events_per_read": {"backward_ext": round(sth.n_ext / len(reads_h), 2), "smems": round(sum(sth.n_smem) / len(reads_h), 2),
                                "sa_lookups": round(sth.n_sa_lookups / len(reads_h), 2), "bsw_cells": round(sth.bsw_cells / len(reads_h), 1)},
            "chains": int(sth.n_chains), "bsw_tasks": int(sth.n_left + sth.n_right), "final_regions": int(sth.n_final_regs),
            "index_build_s": {"first_pass": round(bst_h.ms_first_pass / 1e3, 2), "outputs": round(bst_h.ms_outputs / 1e3, 2), "rounds": int(bst_h.rounds)},
            "note": "the headline steps (FM-index seeding .. dedup, 1 GPU, same read recipe) on simulate.make_genome(profile='grch38_like'): Alu- and "
                    "L1-like interspersed repeats, 171-bp satellite arrays of 10^4 monomers, microsatellites, poly-A runs, four exact 150-kb segmental "
                    "duplications (one inverted), N holes filled as bns_fasta2bntseq does.  The headline genome is uniform random with 10 % of 300-bp repeats",
        }
        batch_h.close()
        ix_h.close()
        del genome_h, reads_h, d_reads_h

    # ---------------- report ----------------
    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        value = total_reads * args.steps / elapsed / 1e6
        st = per_chunk[-1]
        mean = lambda f: float(np.mean([getattr(s, f) for s in per_chunk]))      # noqa: E731  (per chunk)
        r1_ms = mean("ms_smem_r1")
        n_bases = int(cums[-1][-1])
        # algorithmic bytes of the round-1 search kernel per launch:
        # 64 B per CP_OCC block an extension touches + reads in (1 B/base) + SMEMs out (40 B)
        r1_bytes = 64 * st.n_blk_round[0] + n_bases + 40 * st.n_smem[0]
        achieved = r1_bytes / (r1_ms * 1e-3) / 1e9
        r2_bytes = 64 * st.n_blk_round[1] + 40 * st.n_smem[1]
        r3_bytes = 64 * st.n_blk_round[2] + n_bases + 40 * st.n_smem[2]
        all_bytes = (64 * st.n_ext_blocks + n_bases * 3 + 40 * sum(st.n_smem) + 64 * st.n_lf_steps + 13 * st.n_sa_lookups)
        # counter-measured figures of the same workload come from the committed PMC passes (separate rocprofv3 runs cannot
        # be taken inside this one); the file is stamped with the commit and workload it was measured on
        traffic = pmc = None
        tfile = PMC_SUMMARY
        if os.path.exists(tfile):
            try:
                tj = json.load(open(tfile))
                if abs(tj.get("genome_mbp", 0) - args.genome_mbp) < 1 and tj.get("reads") == len(reads_l[-1]) and not (args.fma or args.emf or args.ert):
                    traffic = tj.get("smem_round1_hbm_bytes_per_launch")
                    pmc = tj
            except Exception:
                traffic = pmc = None
        CHn = len(reads_l[-1])
        out = {
            # BASELINE.json: "Mreads/sec aligned, GRCh38 150bp SE, at 1/2/4/8 MI355X; % HBM roofline".  What is timed is the GPU hot
            # path (seeding through mem_sort_dedup_patch), not the SAM side: see not_included
            "metric": f"Mreads/sec through the GPU hot path (seed -> chain -> extend -> dedup, reads and index resident), 150bp SE vs a synthetic "
                      f"genome of {'GRCh38 size' if abs(G - GRCH38_L_PAC) < 1000 else f'{args.genome_mbp:.0f} Mbp'}, at {world} MI355X; % HBM roofline in roofline.frac",
            "value": round(value, 4),
            "unit": "Mreads/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_step, 3),
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "int64 intervals / int16 DP (two columns per register; int32 for queries beyond 191 bases or scores beyond 2^14)",
            "data": "synthetic",
            "not_included": ["mem_mark_primary_se + mem_reg2aln / ksw_global2 traceback / CIGAR + XA + SAM text (timed beside: sam_side)", "FASTQ decode and host I/O",
                             "PCIe transfers (see pcie_inclusive with --pcie)"],
            "config": {
                "workload": f"{total_reads} synthetic 150bp SE reads ({R} per GPU, {n_chunks} resident chunk(s) of <= {CH}) vs a synthetic {G} bp genome "
                            f"({'= GRCh38 l_pac; ' if abs(G - GRCH38_L_PAC) < 1000 else ''}{2 * G + 1} index rows, "
                            f"{len(contigs) if contigs is not None else 1} sequences; GRCh38 itself unavailable offline), "
                            f"FM-index{' + ERT (seeding runs over the ERT)' if args.ert else ''}{' + FMA tables' if args.fma else ''}{' + EMF (L=150)' if args.emf else ''}"
                            f"{'' if (args.fma or args.emf or args.ert) else ' only (no ERT/FMA/EMF)'}; step = per chunk: {'EMF probe, ' if args.emf else ''}"
                            f"{'ERT walk of every read position, the three seeding rounds over the match profiles, sort, hit listing' if args.ert else 'pack reads, SMEM r1-r3, sort, SA lookup'}, "
                            f"chaining + chain filter, extension tasks of the seeds of the kept chains, "
                            f"banded-SW left then right (w=100, retry at 200), region bookkeeping + purge (seeds the reference would extend "
                            f"and then discard are not extended), mem_sort_dedup_patch; everything on the GPU",
                "genome_mbp": round(G / 1e6, 3),
                "index_rows": 2 * G + 1,
                "index_bytes": ix.nbytes,
                "index_build_s": {"first_pass": round(bst.ms_first_pass / 1e3, 2), "outputs": round(bst.ms_outputs / 1e3, 2), "rounds": int(bst.rounds)},
                "reads_per_gpu": R,
                "chains": int(st.n_chains), "regions": int(st.n_chain_seeds),
                "bsw_tasks": int(st.n_left + st.n_right), "bsw_retries": int(st.n_retry_left + st.n_retry_right),
                "ext_rounds": int(st.n_ext_rounds), "final_regions": int(st.n_final_regs),
                "parallelism": f"reads sharded x{world}, index replicated, no collective on the single-end path",
            },
            "stage_ms": {
                "smem_round1": round(r1_ms, 3),
                "smem_round2": round(mean("ms_smem_r2"), 3),
                "smem_round3": round(mean("ms_smem_r3"), 3),
                "sort": round(mean("ms_sort"), 3),
                "sa_lookup": round(mean("ms_sal"), 3),
                "seed_total": round(mean("ms_seed_total"), 3),
                "emf": round(mean("ms_emf"), 3),
                "chain": round(mean("ms_chain"), 3),
                "ext_tasks": round(mean("ms_ext_plan"), 3),
                "ext_left": round(mean("ms_ext_left"), 3),
                "ext_right": round(mean("ms_ext_right"), 3),
                "ext_select": round(mean("ms_ext_purge"), 3),
                "ext_total": round(mean("ms_ext_total"), 3),
                "dedup": round(mean("ms_dedup"), 3),
                "note": "per chunk of reads, HIP events on the batch's stream; ext_tasks/left/right/select are the first extension round, ext_total covers all rounds; "
                        "SMEM round 3 runs beside round 2 on a stream of its own (it fills round 2's tail): its bracket overlaps round 2's, seed_total is the wall time",
            },
            "events_per_read": {
                "backward_ext": round(st.n_ext / CHn, 2),
                "backward_ext_by_round": [round(x / CHn, 2) for x in st.n_ext_round],
                "cp_occ_blocks_by_round": [round(x / CHn, 2) for x in st.n_blk_round],
                "cp_occ_blocks": round(st.n_ext_blocks / CHn, 2),
                "smems": round(sum(st.n_smem) / CHn, 2),
                "sa_lookups": round(st.n_sa_lookups / CHn, 2),
                "lf_steps": round(st.n_lf_steps / CHn, 2),
                "bsw_cells": round(st.bsw_cells / CHn, 1),
                "algorithmic_bytes": round(all_bytes / CHn, 1),
            },
            "roofline": {
                "kernel": "smem_search_kernel<ALL_POS, TAB> + smem_bwd_kernel<TAB> (SMEM round 1: the lane-per-read search and the launch behind it for the backward phases that left their lanes — a wavefront per pivot with a long list, sixteen lanes per pivot otherwise; launch_ms brackets both.  TAB = 2: the kernels read the resident interleaved form of CP_OCC — piece b = count and string of base b — half a block per end of an extension, fetched by a pair of lanes; BWAMS_CP2=0 reads CP_OCC itself)",
                "bound": "hbm",
                "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "traffic_source": None if pmc is None else {"file": os.path.relpath(PMC_SUMMARY, ROOT), "commit": pmc.get("commit"),
                                                            "frac_of_peak_measured_bytes": pmc.get("smem_round1_measured_frac")},
                "bytes_per_launch": int(r1_bytes),
                "bytes_note": "ALGORITHMIC bytes (SURVEY 8d): 64 B x every CP_OCC block an extension touches (1 if k and k + s share a block, else 2), "
                              "counted by the kernels as the oracle counts them, + 1 B per base in + 40 B per SMEM out.  Blocks served from the "
                              "lane's register cache (about 28 % of them) are INCLUDED: they cost no request, and a touched block counts 64 B although the kernels "
                              "fetch only the half that holds the extension's base (32 B of a 128-B line either way).  `traffic` is what the "
                              "counters saw move (2 x FETCH_SIZE + WRITE_SIZE of the committed PMC pass)",
                "launch_ms": round(r1_ms, 3),
                "other_rounds": {"round2_frac": round(r2_bytes / (mean("ms_smem_r2") * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                 "round3_frac": round(r3_bytes / (mean("ms_smem_r3") * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                 "seed_stage_frac": round(all_bytes / (mean("ms_seed_total") * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
            },
        }
        ext_ms = mean("ms_ext_total")
        out["extension"] = {
            "kernels": "bsw_pk_kernel (banded SW, sixteen tasks per wavefront, four lanes x eight 16-bit columns each, packed v_pk_* arithmetic: bound by integer VALU issue, see issue_roof; neither of the contract's two roofs)",
            "tasks": int(st.n_left + st.n_right), "dp_cells": int(st.bsw_cells), "ms_all_rounds": round(ext_ms, 3),
            "Gcells_per_s": round(st.bsw_cells / (ext_ms * 1e-3) / 1e9, 2) if ext_ms > 0 else None,
            "Mtasks_per_s": round((st.n_left + st.n_right) / (ext_ms * 1e-3) / 1e6, 2) if ext_ms > 0 else None,
        }
        if pmc is not None and pmc.get("bsw_valu_insts") and ext_ms > 0:
            # issue roofs of the banded-SW kernels: wave-instructions per step (PMC pass) / the chip's issue rate / time
            v, s_ = pmc["bsw_valu_insts"], pmc["bsw_salu_insts"]
            out["extension"]["issue_roof"] = {
                "valu_wave_insts": v, "salu_wave_insts": s_,
                "valu_frac": round(v / (VALU_PEAK_GINST * 1e9) / (ext_ms * 1e-3), 4),
                "salu_frac": round(s_ / (SALU_PEAK_GINST * 1e9) / (ext_ms * 1e-3), 4),
                "peaks": {"valu_Ginst_s": VALU_PEAK_GINST, "salu_Ginst_s": SALU_PEAK_GINST},
                "note": "fraction of the extension stage's wall time that the counted instructions need at the chip's issue rates: "
                        "integer VALU as measured by tools/ubench_valu.hip (4.4 cycles per wave-instruction per SIMD, add/sub 2.7, "
                        "blended for the kernel's mix), 1 SALU / cycle / CU at 2.4 GHz; counts from the committed PMC pass",
            }
        if args.ert:
            stg, roof, einfo = ert_report(st, mean, CHn, n_bases, ert_info, load_pmc_summary(args.genome_mbp, CHn))
            for k_ in ("smem_round1", "smem_round2", "smem_round3", "sa_lookup"):
                out["stage_ms"].pop(k_, None)
            out["stage_ms"].update(stg)
            out["roofline"] = roof
            out["ert"] = einfo
            for k_ in ("backward_ext", "backward_ext_by_round", "cp_occ_blocks_by_round", "cp_occ_blocks", "lf_steps", "algorithmic_bytes"):
                out["events_per_read"].pop(k_, None)
        elif ert_side is not None:
            out["ert_mode"] = ert_side
            # configs[2] end to end beside `value` (the driver's record keeps the top level)
            out["configs2_fastq_to_sam"] = ert_side["with_emf"].get("fastq_to_sam")
        if hard_side is not None:
            out["hard_genome"] = hard_side
        # the other legs' headline figures where the driver's record keeps them (it keeps `config`, `roofline`, `cpu_baseline` whole and
        # only the NAMES of other top-level keys); the full records stay at the top level
        legs = {"stage_ms": {k_: out["stage_ms"][k_] for k_ in ("seed_total", "chain", "ext_total", "dedup") if k_ in out["stage_ms"]}}
        if hard_side is not None:
            legs["hard_genome"] = {"Mreads_per_s": hard_side.get("value"), "ms_per_step": hard_side.get("ms_per_step"),
                                   "stage_ms": {k_: hard_side["stage_ms"][k_] for k_ in ("seed_total", "chain", "ext_total", "dedup")}}
        if ert_side is not None and not args.ert:
            we_ = ert_side.get("with_emf", {})
            legs["ert_mode_Mreads_per_s"] = ert_side.get("value")
            legs["configs2_step_Mreads_per_s"] = we_.get("value")
            legs["configs2_fastq_to_sam_Mreads_per_s"] = (we_.get("fastq_to_sam") or {}).get("Mreads_per_s")
            legs["configs2_stream"] = {k_: (we_.get("configs2_stream") or {}).get(k_) for k_ in ("Mreads_per_s", "ratio_to_resident", "three_in_flight_Mreads_per_s", "no_overlap_Mreads_per_s", "chunks", "error")
                                       if (we_.get("configs2_stream") or {}).get(k_) is not None}
            legs["ert_walk_roofline"] = {k_: ert_side["roofline"].get(k_) for k_ in ("frac", "launch_ms", "traffic", "bytes_per_launch")}
        if pe_out is not None:
            legs["paired_end_Mreads_per_s"] = pe_out.get("value")
            legs["paired_end_fastq_to_sam_Mreads_per_s"] = (pe_out.get("fastq_to_sam") or {}).get("Mreads_per_s")
        legs["fastq_to_sam_Mreads_per_s"] = (sam_side.get("fastq_to_sam") or {}).get("Mreads_per_s")
        out["config"]["legs"] = legs
        if emf_h is not None:
            _, codes = batch.emf_fetch(CHn)
            emf_ms = mean("ms_emf")
            emf_bytes = 16 * st.emf_nodes + st.emf_cmp_bytes + n_bases
            out["emf"] = {"table_entries": emf_info["num_seed_entry"], "table_bytes": emf_info["num_seed_entry"] * 16,
                          "distinct_lmers": emf_info["n_used"], "build_s": round(emf_info["build_ms"] / 1e3, 2),
                          "resolved_fraction": round(float(((codes == 3) | (codes == 4)).mean()), 4),
                          "launch_ms": round(emf_ms, 3), "nodes_per_read": round(st.emf_nodes / CHn, 3),
                          "algorithmic_bytes": int(emf_bytes),
                          "achieved_GBps": round(emf_bytes / (emf_ms * 1e-3) / 1e9, 1) if emf_ms > 0 else None}
        out["sam_side"] = sam_side
        if pe_out is not None:
            out["paired_end"] = pe_out
        if args.pcie:
            # host buffers in, host buffers out (bwams_seed_fmi + bwams_bsw_extend): never `value`
            enc, cum = simulate.flatten_reads(reads)
            t0 = time.perf_counter()
            for _ in range(2):
                batch.seed(enc, cum, seed_opt)              # upload reads, run, download SMEMs + SA coordinates
                batch.chain_run(mem_opt)
                batch.chain_fetch()                         # download chains
                batch.extend_run(mem_opt)
                batch.dedup_run(mem_opt)
                batch.dedup_fetch()                         # download the final regions
            dt = (time.perf_counter() - t0) / 2
            out["pcie_inclusive"] = {"value": round(len(reads) / dt / 1e6, 4), "unit": "Mreads/s", "ms_per_batch": round(dt * 1e3, 2),
                                     "note": "pageable host buffers: reads up; SMEMs, SA coordinates, chains and final regions down; includes numpy copies"}
        if not args.no_cpu_baseline and world == 1:
            log("timing the CPU oracle on a sample (cpu_baseline)...")
            threads = min(16, os.cpu_count() or 1)
            n_s = min(args.cpu_sample, len(reads))
            host = ix.fetch()                               # host copy of the index for the oracle
            v, dt, n_done = cpu_baseline(host, reads, n_s, threads, contigs)
            out["cpu_baseline"] = {
                "value": round(v, 5), "unit": "Mreads/s", "cores": threads, "kind": "port",
                "sample": f"{n_done} reads of the same batch, same index; oracle seeding+SA+chaining+chain2aln+dedup, "
                          f"{dt:.1f}s wall on {threads} threads (time-boxed)",
                "reference_measured": REFERENCE_MEASURED,
            }
        print(json.dumps(out), flush=True)
    batch.close()
    ix.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
