#!/usr/bin/env python3
"""Turns the rocprofv3 CSVs of profiles/run_profiles.sh (+ a bench.py JSON line) into the committed
round summary: profiles/<round>_summary.md, <round>_kernel_stats.csv, <round>_pmc_summary.json.

    python profiles/make_summary.py r01 gpurun_out/prof_r01f gpurun_out/bench_r01.json
"""
import collections
import csv
import glob
import json
import shutil
import sys

rnd, src, bench_path = sys.argv[1], sys.argv[2], sys.argv[3]
bench = json.loads(open(bench_path).read().strip().splitlines()[-1])


def short(k):
    for pat, name in (("smem_search_kernel<true", "smem_search_kernel<true> (SMEM round 1)"),
                      ("smem_search_kernel<false", "smem_search_kernel<false> (SMEM round 2)"),
                      ("smem_bwd_wave", "smem_bwd_wave_kernel (rounds 1 and 2: backward phases with long interval lists, wave per pivot; one launch behind each search kernel)"),
                      ("smem_bwd_kernel", "smem_bwd_kernel (rounds 1 and 2: the backward phases that left their lanes — a wavefront per pivot with a long list, sixteen lanes per pivot otherwise; one launch behind each search kernel)"),
                      ("seed_strategy", "seed_strategy_kernel (SMEM round 3; runs beside round 2 on a stream of its own: the durations overlap, 3.3 ms alone)"), ("sa_lookup", "sa_lookup_kernel"),
                      ("bsw_pk_kernel", "bsw_pk_kernel (banded SW, 16 tasks per wave, packed 16-bit columns; 5 query-length classes)"),
                      ("bsw_qwin_kernel", "bsw_qwin_kernel (banded SW, 8 tasks per wave, 32-bit: scoring the packed kernel does not take)"),
                      ("bsw_classify", "bsw_classify_kernel"), ("bsw_kernel", "bsw_kernel (one task per wave, LDS: queries > 191)"),
                      ("aln_dp_wave", "aln_dp_wave_kernel (mem_reg2aln: bands beyond 32 columns, wave per region)"), ("aln_dp_kernel", "aln_dp_kernel (mem_reg2aln: banded global alignment + traceback, lane per region)"),
                      ("sam_need", "sam_need_kernel (which regions the SAM text reads)"),
                      ("aln_simple", "aln_simple_kernel (mem_reg2aln: gap-free regions)"), ("aln_gather", "aln_gather_kernel"),
                      ("aln_plan", "aln_plan_kernel"),
                      ("sam_text_kernel", "sam_text_kernel (single-end SAM text: count pass / write pass)"), ("sam_mapq", "sam_mapq_kernel (mem_approx_mapq_se)"),
                      ("fastq_emit", "fastq_emit_kernel (FASTQ decode: copy / encode, wave per record)"), ("fastq_measure", "fastq_measure_kernel (FASTQ decode: validate + measure, lane per record)"),
                      ("fastq_count", "fastq_count_kernel (FASTQ decode: line ends)"),
                      ("key_collect", "key_collect_kernel (index build: MSD chunk collection)"),
                      ("chunk_finish", "chunk_finish_kernel (index build)"), ("bwt_block", "bwt_block_kernel (index build: BWT -> CP_OCC)"),
                      ("round_keys", "round_keys_kernel (index build: doubling round)"), ("round_finish", "round_finish_kernel (index build)"),
                      ("key_hist", "key_hist_kernel (index build)"), ("sa_sample", "sa_sample_kernel (index build)"),
                      ("chain_count", "chain_count_kernel"), ("chain_wave", "chain_wave_kernel (wave per read, LDS state; 6 size classes)"),
                      ("chain_heavy", "chain_heavy_kernel (sort + filter of many-chain reads)"),
                      ("chain_emit", "chain_emit_kernel"), ("chain_kernel", "chain_kernel (lane per read)"),
                      ("ext_plan", "ext_plan_kernel"), ("ext_build", "ext_build_kernel (task construction)"),
                      ("ext_post", "ext_post_kernel"), ("ext_select_wave", "ext_select_wave_kernel (selection / purge, wave per read)"),
                      ("ext_select", "ext_select_kernel (selection / purge, lane per read)"),
                      ("dedup_triage", "dedup_triage_kernel"), ("dedup_wave", "dedup_wave_kernel (mem_sort_dedup_patch, wave per read)"),
                      ("dedup_gather", "dedup_gather_kernel"), ("dedup_kernel", "dedup_kernel (mem_sort_dedup_patch, lane per read)"),
                      ("seedsw_", "seedsw kernels (long reads only)"),
                      ("pair_post_wave", "pair_post_wave_kernel (mate rescue into long region lists, wave per read)"),
                      ("pair_post", "pair_post_kernel (mate rescue, lane per read)"),
                      ("pair_mark_wave", "pair_mark_wave_kernel (mem_mark_primary_se, wave per read)"),
                      ("pair_mark", "pair_mark_kernel (mem_mark_primary_se, lane per read)"),
                      ("pair_pair", "pair_pair_kernel (mem_pair)"), ("pair_plan", "pair_plan_kernel (rescue windows)"),
                      ("pair_build", "pair_build_kernel (rescue tasks)"), ("pair_gather", "pair_gather_kernel"),
                      ("pair_count", "pair_count_kernel"), ("pestat", "pestat_kernel"),
                      ("ext_heavy_list", "ext_heavy_list_kernel"),
                      ("pack_reads", "pack_reads_kernel"), ("round2_work", "round2_work_kernel"),
                      ("make_keys", "make_keys_kernel"), ("gather_sorted", "gather_sorted_kernel"),
                      ("plan_kernel", "plan_kernel (task construction)"), ("build_kernel", "build_kernel (task construction)"),
                      ("ert_profile", "ert_profile_kernel (ERT: one forward walk per read position)"),
                      ("ert_select", "ert_select_kernel (ERT: the three seeding rounds over the match profiles)"),
                      ("ert_locate", "ert_locate_kernel (ERT: where a seed's hits are, their number)"),
                      ("ert_hits", "ert_hits_kernel (ERT: hits by rank descent, lane per sampled hit)"),
                      ("ert_gather", "ert_gather_kernel (ERT: serial leaf walk, fallback)"),
                      ("ert_size", "ert_size_kernel (ERT build: sizes and pointer widths)"),
                      ("ert_emit", "ert_emit_kernel (ERT build: bytes)"),
                      ("ert_count", "ert_count_kernel"), ("ert_clear", "ert_clear_kernel"),
                      ("emf_probe", "emf_probe_kernel"), ("ksw_kernel", "ksw_kernel")):
        if pat in k:
            return name
    return None


import os


def newest(pattern):
    fs = sorted(glob.glob(pattern), key=os.path.getmtime)
    return fs[-1:] 


BWD1 = "smem_bwd_kernel, the launch behind round 1"
BWD2 = "smem_bwd_kernel, the launch behind round 2"
ks = newest(src + "/trace/*/*_kernel_stats.csv")[0]
shutil.copy(ks, f"profiles/{rnd}_kernel_stats.csv")
rows = list(csv.DictReader(open(ks)))
P = {}
N = {}          # launches per run (3 steps)
for p in ("pmc_fetch", "pmc_write", "pmc_l2", "pmc_sq"):
    fs = newest(f"{src}/{p}/*/*_counter_collection.csv")
    if not fs:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    rws = list(csv.DictReader(open(fs[0])))
    if rws and "Dispatch_Id" in rws[0]:
        rws.sort(key=lambda r: int(r["Dispatch_Id"]))
    for r in rws:
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        s = short(k)
        if s:
            for c, x in v.items():
                P.setdefault(s, {})[c] = sum(x) / len(x)
                N[s] = len(x)
                if "smem_bwd_wave" in k or "smem_bwd_kernel" in k:          # launches alternate: behind round 1, behind round 2
                    P.setdefault(BWD1, {})[c] = sum(x[0::2]) / max(len(x[0::2]), 1)
                    P.setdefault(BWD2, {})[c] = sum(x[1::2]) / max(len(x[1::2]), 1)
                    N[BWD1], N[BWD2] = len(x[0::2]), len(x[1::2])

E = {}          # the ERT walk kernel's counters (passes over bench.py --ert)
for p in ("pmc_fetch_ert", "pmc_write_ert", "pmc_sq_ert"):
    fs = newest(f"{src}/{p}/*/*_counter_collection.csv")
    if not fs:
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "ert_profile_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, x in acc.items():
        E[c] = sum(x) / len(x)

r1 = P["smem_search_kernel<true> (SMEM round 1)"]
# passes of the hot path in one run = launches of SMEM round 1: the bench's steps (1 warm-up + 2 timed) and the three text-to-text
# calls of its sam_side.fastq_to_sam leg (bwams_process_chunk runs the same kernels on the same reads)
n_pass = N.get("smem_search_kernel<true> (SMEM round 1)", 3)
n_pass_trace = sum(int(r["Calls"]) for r in rows if "smem_search_kernel<true" in r["Name"]) or 3
# round 1 = the search kernel + the wave kernel behind it (bench.py's HIP events bracket both)
b1 = P.get(BWD1, {})
fetch, write = (r1["FETCH_SIZE"] + b1.get("FETCH_SIZE", 0)) * 1024, (r1["WRITE_SIZE"] + b1.get("WRITE_SIZE", 0)) * 1024
alg = bench["roofline"]["bytes_per_launch"]
with open(f"profiles/{rnd}_summary.md", "w") as f:
    f.write(f"# Round {rnd} — rocprofv3 summary (MI355X, gfx950, ROCm 7.2)\n\n")
    f.write(f"Collected by `bash profiles/run_profiles_{rnd}.sh <tag>` at the metric's configuration (synthetic genome of GRCh38's size, 6.4 G index rows): "
            "`rocprofv3 --kernel-trace --stats -- python3 bench.py "
            "--steps 2 --warmup 1 --no-cpu-baseline --no-pe --no-ert-leg --no-hard-genome` plus one `--pmc` pass per counter group (never combined "
            f"with tracing); summarised by `profiles/make_summary.py`.  Raw: `profiles/{rnd}_kernel_stats.csv`.\n\n")
    f.write("## Kernel time (library kernels; rocPRIM kernels omitted; the index-build kernels run once, untimed by bench.py)\n\n"
            f"{n_pass_trace} passes of the hot path per run (1 warm-up + 2 timed steps, and 3 text-to-text calls of `sam_side.fastq_to_sam`, which run "
            "the same kernels on the same reads); kernels that run once per extension round or per query-length class have "
            f"several calls per pass, so the per-step column is total / {n_pass_trace}.\n\n| kernel | calls | avg ms | ms per step |\n|---|---|---|---|\n")
    for r in rows:
        s = short(r["Name"])
        if s:
            f.write(f"| {s} | {r['Calls']} | {float(r['AverageNs'])/1e6:.3f} | {float(r['TotalDurationNs'])/n_pass_trace/1e6:.2f} |\n")
    pe = newest(src + "/trace_pe/*/*_kernel_stats.csv")
    if pe and "paired_end" in bench:
        shutil.copy(pe[0], f"profiles/{rnd}_pe_kernel_stats.csv")
        f.write("\n### Paired-end leg (`bench.py`'s `paired_end` object: 500 k pairs; 3 calls = 1 warm-up + 2 timed batches)\n\n"
                "| kernel | calls | avg ms |\n|---|---|---|\n")
        for r in csv.DictReader(open(pe[0])):
            s_ = short(r["Name"])
            if s_ and ("pair_" in s_ or "ksw" in s_ or "pestat" in s_):
                f.write(f"| {s_} | {r['Calls']} | {float(r['AverageNs'])/1e6:.3f} |\n")
    te = newest(src + "/trace_ert/*/*_kernel_stats.csv")
    if te:
        shutil.copy(te[0], f"profiles/{rnd}_ert_kernel_stats.csv")
        f.write("\n### Seeding over the ERT (`bench.py --ert --steps 2 --warmup 1 --no-cpu-baseline --no-pe`; the build kernels run once)\n\n"
                "| kernel | calls | avg ms |\n|---|---|---|\n")
        for r in csv.DictReader(open(te[0])):
            s_ = short(r["Name"])
            if s_ and "ert_" in s_:
                f.write(f"| {s_} | {r['Calls']} | {float(r['AverageNs'])/1e6:.3f} |\n")
        if E.get("FETCH_SIZE") is not None:
            ef, ew = E["FETCH_SIZE"] * 1024, E.get("WRITE_SIZE", 0) * 1024
            f.write(f"\nWalk kernel (`ert_profile_kernel`) per launch: FETCH_SIZE {ef/1e9:.2f} GB, WRITE_SIZE {ew/1e9:.2f} GB -> HBM traffic = 2 x "
                    f"{ef/1e9:.2f} + {ew/1e9:.2f} = **{(2*ef+ew)/1e9:.1f} GB**; SQ: ACTIVE_INST_ANY / WAVE_CYCLES = "
                    f"{E.get('SQ_ACTIVE_INST_ANY',0)/max(E.get('SQ_WAVE_CYCLES',1),1):.3f}, WAIT_ANY / WAVE_CYCLES = {E.get('SQ_WAIT_ANY',0)/max(E.get('SQ_WAVE_CYCLES',1),1):.3f}, "
                    f"{E.get('SQ_INSTS_VALU',0)/1e9:.2f} G VALU + {E.get('SQ_INSTS_SALU',0)/1e9:.2f} G SALU + {E.get('SQ_INSTS_VMEM_RD',0)/1e9:.3f} G VMEM-read wave-instructions.\n")
    kt = newest(src + "/trace/*/*_kernel_trace.csv")
    if kt:
        allk = sorted(csv.DictReader(open(kt[0])), key=lambda r: int(r["Start_Timestamp"]))
        dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6       # noqa: E731
        tr = [r for r in allk if "smem_bwd_wave" in r["Kernel_Name"] or "smem_bwd_kernel" in r["Kernel_Name"]]
        s1 = [r for r in allk if "smem_search_kernel<true" in r["Kernel_Name"]]
        # the run also launches the search on HALF-size chunks (sam_side's two-batches-behind-one-call leg): averaged apart
        top = max(dur(r) for r in s1) if s1 else 0.0
        full = [r for r in s1 if dur(r) > 0.75 * top]
        half = [r for r in s1 if dur(r) <= 0.75 * top]
        # the launch behind every full-size round-1 search: the first smem_bwd launch that starts after it
        behind = []
        for r in full:
            nxt = [x for x in tr if int(x["Start_Timestamp"]) >= int(r["End_Timestamp"])]
            if nxt:
                behind.append(dur(nxt[0]))
        if full and behind:
            fs, fb = sum(dur(r) for r in full) / len(full), sum(behind) / len(behind)
            f.write(f"\nRound 1 = `smem_search_kernel<true>` + the launch of `smem_bwd_kernel` behind it.  FULL-size launches only ({len(full)} of {len(s1)}: the run "
                    f"also searches half-size chunks {len(half)} times — `sam_side`'s two-batches-behind-one-call leg — which the per-kernel table above averages in: "
                    f"{(sum(dur(r) for r in half) / len(half)) if half else 0.0:.2f} ms each): rocprofv3 averages {fs:.2f} + {fb:.2f} = **{fs + fb:.2f} ms** "
                    f"(search min {min(dur(r) for r in full):.2f} / max {max(dur(r) for r in full):.2f}), "
                    f"to compare with `roofline.launch_ms` = {bench['roofline']['launch_ms']} ms, which brackets both with HIP events.\n")
    f.write(f"\n`roofline.launch_ms` measured live by `bench.py` with HIP events in the run below: {bench['roofline']['launch_ms']} ms "
            "(the rocprofv3 average above covers warm-up + timed launches of the profiled run).\n\n")
    f.write("## PMC per launch (uncorrected counter values; average over the launches of one run)\n\n"
            "| kernel | launches per run | FETCH_SIZE (GB) | WRITE_SIZE (GB) | TCC hit / miss (M req) | ACTIVE_INST_ANY / WAVE_CYCLES | WAIT_ANY / WAVE_CYCLES | VALU / SALU wave-insts (G) |\n|---|---|---|---|---|---|---|---|\n")
    for s, c in P.items():
        if "FETCH_SIZE" not in c:
            continue
        f.write(f"| {s} | {N.get(s, 0)} | {c['FETCH_SIZE']*1024/1e9:.2f} | {c.get('WRITE_SIZE',0)*1024/1e9:.2f} | {c.get('TCC_HIT_sum',0)/1e6:.0f} / "
                f"{c.get('TCC_MISS_sum',0)/1e6:.0f} | {c.get('SQ_ACTIVE_INST_ANY',0)/max(c.get('SQ_WAVE_CYCLES',1),1):.3f} | "
                f"{c.get('SQ_WAIT_ANY',0)/max(c.get('SQ_WAVE_CYCLES',1),1):.3f} | "
                f"{c.get('SQ_INSTS_VALU',0)/1e9:.2f} / {c.get('SQ_INSTS_SALU',0)/1e9:.2f} |\n")
    f.write(f"""
**Calibration of FETCH_SIZE on this access pattern** (`tools/ubench_gather.hip` under `rocprofv3 --pmc FETCH_SIZE` and
`--pmc TCC_EA0_RDREQ_sum`, known byte counts): a random 16-, 32-, 64- and 128-byte read per lane all report exactly one
`TCC_EA0_RDREQ` and FETCH_SIZE = 64 B per access — the full-line case is under-reported by 2x, as MI355X_MICROARCH.md §HBM
prescribes ("128-B requests tallied at 64 B"), and all four shapes saturate at the same ~49 G requests/s = 6.3 TB/s of
128-B lines.  Every random request moves one 128-B line: **HBM read bytes = 2 x FETCH_SIZE**; WRITE_SIZE is exact.

Round 1 (search kernel + the wave kernel behind it: `roofline.launch_ms` brackets both) per pass: algorithmic bytes {alg/1e9:.2f} GB; corrected traffic = 2 x {fetch/1e9:.2f} + {write/1e9:.2f}
= **{(2*fetch+write)/1e9:.1f} GB** = {(2*fetch+write)/alg:.2f}x algorithmic.  At {bench['roofline']['launch_ms']} ms per launch that is
{(2*fetch+write)/(bench['roofline']['launch_ms']*1e-3)/1e12:.2f} TB/s of HBM traffic ({(2*fetch+write)/(bench['roofline']['launch_ms']*1e-3)/8e12:.2f} of the 8 TB/s peak,
{(2*fetch+write)/(bench['roofline']['launch_ms']*1e-3)/6.3e12:.2f} of the 6.3 TB/s a streaming copy reaches), against `roofline.frac` = {bench['roofline']['frac']} in algorithmic bytes.
L2 misses per launch of the search kernel {r1.get('TCC_MISS_sum',0)/1e6:.0f} M = {r1.get('TCC_MISS_sum',0)/(bench['roofline']['launch_ms']*1e-3)/1e9:.1f} G lines/s; `tools/ubench_gather` (mode 1, the kernel's quad-cooperative
fetch, a 12 GiB table) tops out at 48 G random 64-byte blocks/s = 3.07 TB/s of useful bytes = 0.38 of peak when every block is its own line.

Banded-SW kernels per step: {sum(c.get("SQ_INSTS_VALU", 0) * N.get(s_, 0) for s_, c in P.items() if s_.startswith("bsw_")) / n_pass / 1e9:.1f} G vector and
{sum(c.get("SQ_INSTS_SALU", 0) * N.get(s_, 0) for s_, c in P.items() if s_.startswith("bsw_")) / n_pass / 1e9:.1f} G scalar wave-instructions (issue rates: integer VALU 555-570 G/s measured by `tools/ubench_valu.hip` = 4.4 cycles per SIMD, add/sub 910 G/s;
614.4 G SALU/s at one per cycle per CU, 2.4 GHz) in `stage_ms.ext_total` = {bench['stage_ms']['ext_total']} ms.

## Smith-Waterman kernels alone, against the real reference objects on the host cores

`python tests/bench_sw_kernels.py` (profiles/{rnd}_sw_kernels.jsonl): the CPU column is `BandedPairWiseSW::getScores16` (AVX512) and
`ksw_align2` (SSE2) of `oracle/_ref` — the reference's own code compiled from its tree — on the same tasks, 16 threads.

```
{open(f"profiles/{rnd}_sw_kernels.jsonl").read() if __import__("os").path.exists(f"profiles/{rnd}_sw_kernels.jsonl") else ""}```

## bench.py line of the same build

```json
{json.dumps(bench, indent=1)}
```
""")
json.dump({"genome_mbp": bench["config"]["genome_mbp"], "reads": bench["config"]["reads_per_gpu"],
           "smem_round1_hbm_bytes_per_launch": int(2 * fetch + write),
           "smem_round1_fetch_size_bytes": int(fetch), "smem_round1_write_size_bytes": int(write),
           "correction": "HBM read bytes = 2 x FETCH_SIZE (calibrated with tools/ubench_gather: one 128-B line per random request), WRITE_SIZE exact",
           "smem_round1_measured_frac": round((2 * fetch + write) / (bench["roofline"]["launch_ms"] * 1e-3) / 8e12, 4),
           "smem_round1_l2_hit_miss": [int(r1.get("TCC_HIT_sum", 0)), int(r1.get("TCC_MISS_sum", 0))],
           "bsw_valu_insts": int(sum(c.get("SQ_INSTS_VALU", 0) * N.get(s_, 0) for s_, c in P.items() if s_.startswith("bsw_")) / n_pass),
           "bsw_salu_insts": int(sum(c.get("SQ_INSTS_SALU", 0) * N.get(s_, 0) for s_, c in P.items() if s_.startswith("bsw_")) / n_pass),
           "bsw_valu_per_cell": round(sum(c.get("SQ_INSTS_VALU", 0) * N.get(s_, 0) for s_, c in P.items() if s_.startswith("bsw_")) / n_pass / max(bench["extension"]["dp_cells"], 1), 3),
           "ert_walk_hbm_bytes_per_launch": int(2 * E["FETCH_SIZE"] * 1024 + E.get("WRITE_SIZE", 0) * 1024) if "FETCH_SIZE" in E else None,
           "ert_walk_fetch_write_bytes": [int(E["FETCH_SIZE"] * 1024), int(E.get("WRITE_SIZE", 0) * 1024)] if "FETCH_SIZE" in E else None,
           "ert_walk_valu_salu_vmem_insts": [int(E.get("SQ_INSTS_VALU", 0)), int(E.get("SQ_INSTS_SALU", 0)), int(E.get("SQ_INSTS_VMEM_RD", 0))] if E else None,
           "commit": os.popen("git rev-parse --short HEAD").read().strip(),
           "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_INSTS_* (separate passes), profiles/run_profiles_" + rnd + ".sh; bsw_* = wave-instructions per step"},
          open(f"profiles/{rnd}_pmc_summary.json", "w"), indent=1)
print(open(f"profiles/{rnd}_summary.md").read()[:2500])
