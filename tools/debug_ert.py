"""step-by-step ERT seeding run with progress lines (debug aid for the GPU box)"""
import faulthandler, sys, time, os
faulthandler.dump_traceback_later(150, exit=True)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bwa-mem-scale_amd"), os.path.join(ROOT, "tests")]
import numpy as np
def P(*a):
    print(*a, flush=True)
from bwams import capi, fmindex, simulate
from oracle import loader
import test_gpu_ert as T
P("imports done")
g, idx, o, e, ix, ert = T._make(100000, 1, 8, 2, 16)
P("index + ert resident", ert.nbytes())
enc, cum = T._reads(g, int(sys.argv[1]) if len(sys.argv) > 1 else 1500, 1)
oo, go = T._opts()
want, wcoord, woff = e.collect(enc, cum, oo)
P("oracle", len(want), len(wcoord))
b = capi.Batch(ix, len(cum) - 1, int(cum[-1]))
b.seed_upload(enc, cum)
P("uploaded")
t = time.time()
b.seed_run_ert(ert, go, with_sa=False)
b.sync()
P("run (no sa)", time.time() - t)
ns, na = b.seed_counts()
P("counts", ns, na)
b.seed_run_ert(ert, go, with_sa=True)
b.sync()
P("run (sa)")
got, coord, off = b.seed_fetch()
P("fetched", len(got), len(coord))
for f in ("rid", "m", "n", "s"):
    P(f, np.array_equal(got[f], want[f]))
P("off", np.array_equal(off, woff), "coord", np.array_equal(coord, wcoord))
st = b.stats()
P("ms profile %.3f select %.3f locate %.3f sort %.3f hits %.3f total %.3f" % (st.ms_smem_r1, st.ms_smem_r2, st.ms_smem_r3, st.ms_sort, st.ms_sal, st.ms_seed_total))
