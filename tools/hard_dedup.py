"""Where de-duplication's time goes on the grch38_like genome: regions per read going into mem_sort_dedup_patch, neighbours within
max_chain_gap, candidate pairs that pass mem_patch_reg's coordinate tests."""
import sys, time
import numpy as np
sys.path.insert(0, "bwa-mem-scale_amd"); sys.path.insert(0, ".")
from bwams import capi, simulate
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 3_209_286_105
N = 1_000_000
g = simulate.make_genome(n, seed=2024, profile="grch38_like")
ix = capi.Index.build(g, 0)
contigs = simulate.chromosomes(n) if n >= 2 ** 31 else None
if contigs is not None:
    ix.set_contigs(contigs)
reads = simulate.make_reads(g, N, seed=12345, contig_bounds=None if contigs is None else simulate.contig_bounds(contigs))[0]
enc, cum = simulate.flatten_reads(reads)
b = capi.Batch(ix, N, int(cum[-1]), max_smem=32 * N, max_sa=128 * N)
b.seed_upload(enc, cum)
opt = capi.default_mem_opt()
b.seed_run(capi.default_seed_opt(), with_sa=True); b.chain_run(opt); b.extend_run(opt)
regs, off, _ = b.extend_fetch()
alive = regs["qe"] > regs["qb"]
per = np.add.reduceat(alive.astype(np.int64), off[:-1].clip(max=len(regs) - 1)) * (np.diff(off) > 0)
slots = np.diff(off)
print("reads", N, "region slots", len(regs), "alive", int(alive.sum()))
for lo, hi in ((0, 1), (2, 32), (33, 128), (129, 512), (513, 2048), (2049, 10 ** 9)):
    m = (slots >= lo) & (slots <= hi)
    print(f"  slots {lo}..{hi}: {int(m.sum())} reads, {int(slots[m].sum())} slots, alive {int(per[m].sum())}")
# neighbours: for reads in the upper tiers, pairs (i, j) the serial loop visits
rng = np.random.default_rng(1)
big = np.flatnonzero(slots > 512)
tot_pairs = tot_geom = tot_red = 0
for r in big[:200]:
    a = regs[off[r]:off[r + 1]]
    a = a[a["qe"] > a["qb"]]
    a = a[np.argsort(a["re"], kind="stable")]
    pairs = 0
    rb, re, qb, qe, sc = (a[f].astype(np.int64) for f in ("rb", "re", "qb", "qe", "score"))
    for i in range(1, len(a)):
        j = i - 1
        while j >= 0 and a["rid"][j] == a["rid"][i] and rb[i] < re[j] + 10000:
            pairs += 1
            or_ = re[j] - rb[i]; oq = (qe[j] - qb[i]) if qb[j] < qb[i] else (qe[i] - qb[j])
            mr = min(re[j] - rb[j], re[i] - rb[i]); mq = min(qe[j] - qb[j], qe[i] - qb[i])
            red = or_ > 0.95 * mr and oq > 0.95 * mq
            if red: tot_red += 1
            elif rb[j] < rb[i] and not (qb[j] >= qb[i] or qe[j] >= qe[i] or re[j] >= re[i]):
                w = abs((re[j] - rb[i]) - (qe[j] - qb[i]))
                rr = abs((re[j] - rb[i]) / (re[i] - rb[j]) - (qe[j] - qb[i]) / (qe[i] - qb[j]))
                if re[j] < rb[i] or qe[j] < qb[i]:
                    ok = not (w > 200 or rr >= 0.05)
                else:
                    ok = not (w > 400 or rr >= 0.1)
                if ok: tot_geom += 1
            j -= 1
    tot_pairs += pairs
    if r in big[:5]:
        print("   read", r, "alive", len(a), "pairs", pairs, "scores", np.unique(a["score"])[-5:], "qb/qe", a["qb"][:5], a["qe"][:5], "rb gaps", np.diff(np.sort(a["rb"]))[:8])
print("sampled", min(200, len(big)), "big reads: pairs visited", tot_pairs, "redundant", tot_red, "pass mem_patch_reg geometry (-> alignment)", tot_geom)
t = time.time(); nn = b.dedup_run(opt); b.sync(); print("dedup_run", time.time() - t, nn)

import re as _re
