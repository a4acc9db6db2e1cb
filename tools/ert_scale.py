"""ERT at genome scale: build on the GPU, seed one chunk through the ERT and through the FM-index, compare, time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bwa-mem-scale_amd")]
import numpy as np
import torch
from bwams import capi, simulate

def P(*a):
    print(*a, flush=True)

G = int(float(sys.argv[1]) * 1e6) if len(sys.argv) > 1 else 3_209_286_105
NR = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
torch.cuda.init()
t = time.time()
genome = simulate.make_genome(G, seed=2024)
contigs = simulate.chromosomes(G) if G >= 2 ** 31 else None
cb = None if contigs is None else simulate.contig_bounds(contigs)
P(f"genome {G/1e6:.0f} Mbp in {time.time()-t:.1f}s")
t = time.time()
ix = capi.Index.build(genome, 0)
if contigs is not None:
    ix.set_contigs(contigs)
P(f"FM-index in {time.time()-t:.1f}s, {ix.nbytes/2**30:.1f} GiB")
os.environ["BWAMS_VERBOSE"] = "1"
t = time.time()
ert = capi.Ert.build(ix)
info = ert.info()
P(f"ERT in {time.time()-t:.1f}s: k-mer table 8 GiB + trees {info['mlt_bytes']/2**30:.2f} GiB; kernel ms {info['build_ms']}")
os.environ["BWAMS_VERBOSE"] = "0"
reads = simulate.make_reads(genome, NR, seed=12345, contig_bounds=cb)[0]
enc, cum = simulate.flatten_reads(reads)
b = capi.Batch(ix, NR, int(cum[-1]), max_smem=32 * NR, max_sa=128 * NR)
b.seed_upload(enc, cum)
opt = capi.default_seed_opt()
for rep in range(3):
    b.seed_run_ert(ert, opt); b.sync()
    st = b.stats()
    P("ERT  ms profile %.3f select %.3f locate %.3f sort %.3f hits+locate %.3f total %.3f" % (st.ms_smem_r1, st.ms_smem_r2, st.ms_smem_r3, st.ms_sort, st.ms_sal, st.ms_seed_total))
sm_e, co_e, off_e = b.seed_fetch()
for rep in range(2):
    b.seed_run(opt); b.sync()
    st = b.stats()
    P("FM   ms r1 %.3f r2 %.3f r3 %.3f sort %.3f sal %.3f total %.3f" % (st.ms_smem_r1, st.ms_smem_r2, st.ms_smem_r3, st.ms_sort, st.ms_sal, st.ms_seed_total))
sm_f, co_f, off_f = b.seed_fetch()
P("seeds", len(sm_e), len(sm_f), "coords", len(co_e), len(co_f))
ok = len(sm_e) == len(sm_f) and all(np.array_equal(sm_e[f], sm_f[f]) for f in ("rid", "m", "n", "s")) and np.array_equal(off_e, off_f)
okc = ok and bool(np.all((co_e == co_f) | ((co_f == 0) & (co_e < 128))))
P("ERT seeds == FM seeds:", ok, " coordinates:", okc)
