"""The stage sequence of tests/test_gpu_fullsize.py::test_configs2_index_set_ert_plus_emf_sampled_parity on a grch38_like genome, with a
time stamp after every step (to find the step that is slow on the harder genome)."""
import sys, time
import numpy as np
sys.path.insert(0, "bwa-mem-scale_amd"); sys.path.insert(0, "."); sys.path.insert(0, "tests")
from bwams import capi, simulate
from oracle import loader
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3_209_286_105
profile = sys.argv[2] if len(sys.argv) > 2 else "grch38_like"
N_READS = 1_000_000
T0 = time.time()
def stamp(what):
    print(f"{time.time() - T0:8.1f} s  {what}", flush=True)
g = simulate.make_genome(n, seed=77, profile=None if profile == "none" else profile); stamp("genome")
ix = capi.Index.build(g, 0); stamp("index")
contigs = simulate.chromosomes(n) if n >= 2 ** 31 else None
if contigs is not None:
    ix.set_contigs(contigs)
host = ix.fetch(); stamp("fetch")
reads, _, _ = simulate.make_reads(g, N_READS, seed=99, contig_bounds=None if contigs is None else simulate.contig_bounds(contigs))
enc, cum = simulate.flatten_reads(reads); stamp("reads")
ert = capi.Ert.build(ix); stamp("ert build " + str(ert.info()))
emf = capi.Emf.build(ix, seed_len=150, slack=1.1); stamp("emf build")
b = capi.Batch(ix, N_READS, int(cum[-1]), max_smem=32 * N_READS, max_sa=128 * N_READS)
b.seed_upload(enc, cum)
opt, sopt = capi.default_mem_opt(), capi.default_seed_opt()
b.emf_run(emf); perfect, code = b.emf_fetch(N_READS); stamp("emf probe")
eregs, eoff, erev = b.emf_regs(emf, opt); stamp("emf regs")
b.seed_run_ert(ert, sopt, with_sa=True); b.sync(); stamp("seed_run_ert")
sm, coord, sa_off = b.seed_fetch(); stamp(f"seed fetch: {len(sm)} smems, {len(coord)} coords, max s {int(sm['s'].max())}")
b.chain_run(opt); b.sync(); stamp("chain")
b.extend_run(opt); b.sync(); stamp("extend")
nn = b.dedup_run(opt); b.sync(); stamp(f"dedup {nn}")
st = b.stats(); stamp("stats " + ", ".join(f"{k}={getattr(st, k)}" for k, _ in st._fields_ if "time" not in k)[:1500])
b.seed_run(sopt, with_sa=True); b.sync(); stamp("seed_run FM")
resolved = (code == 3) | (code == 4)
rng = np.random.default_rng(8)
heavy = np.unique(sm["rid"][sm["s"] > 500])
stamp(f"heavy reads {len(heavy)}")
pick = np.unique(np.concatenate([rng.choice(N_READS, size=2300, replace=False), rng.choice(heavy, size=min(200, len(heavy)), replace=False)]))
kt, mt = ert.fetch(pad=16); stamp("ert fetch")
e = loader.OracleERT.from_tables(kt, mt, host.ref_0123)
sub = pick[~resolved[pick]]
sub_enc, sub_cum = simulate.flatten_reads(reads[sub])
oo = loader.default_seed_opt()
ref_sm, ref_coord, ref_off, cls, flags = e.walk_collect(sub_enc, sub_cum, oo); stamp(f"oracle walk_collect {len(sub)} reads flags {flags}")
mems, mem_off, hits, hit_off, flags = e.walk(sub_enc, sub_cum, oo); stamp("oracle walk")
ch, sd, choff = loader.chain_new_ert(mems, mem_off, hits, hit_off, sub_cum, len(g), contigs=contigs); stamp(f"oracle chain {len(ch)} chains {len(sd)} seeds")
regs, reg_off, _ = loader.chain2aln(ch, sd, choff, sub_enc, sub_cum, host.ref_0123, len(g), contigs=contigs); stamp(f"oracle chain2aln {len(regs)}")
wfin, wfin_off = loader.regs_finish(regs, reg_off, sub_enc, sub_cum, host.ref_0123, len(g), contigs=contigs); stamp("oracle finish")
