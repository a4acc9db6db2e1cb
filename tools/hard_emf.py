"""EMF (exact-match filter) build on the grch38_like genome at full size: what the builder reports, and the text at the windows it names."""
import re, sys, time
import numpy as np
sys.path.insert(0, "bwa-mem-scale_amd")
from bwams import capi, simulate
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3_209_286_105
t = time.time(); g = simulate.make_genome(n, seed=77, profile="grch38_like"); print("genome", time.time() - t, flush=True)
t = time.time(); ix = capi.Index.build(g, 0); print("index", time.time() - t, flush=True)
try:
    t = time.time(); e = capi.Emf.build(ix, seed_len=150, slack=1.1); print("emf ok", time.time() - t, e.info())
except capi.BwamsError as ex:
    print(ex)
    m = re.search(r"first two at (\d+) and (\d+)", str(ex))
    for p in (int(m.group(1)), int(m.group(2))):
        print(p, "".join("ACGT"[b] for b in g[p:p + 150]))
