"""Bisect index-size problems: build on the GPU at several sizes, seed 2000 reads on GPU and oracle, compare."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bwa-mem-scale_amd")]
import numpy as np
import torch
torch.cuda.init()
from bwams import capi, fmindex, simulate
from oracle import loader

sizes = [float(x) for x in sys.argv[1:]] or [1.0e9, 1.2e9, 2.2e9]
for n in sizes:
    n = int(n)
    t0 = time.time()
    g = simulate.make_genome(n, seed=77)
    t1 = time.time()
    ix = capi.Index.build(g, 0)
    t2 = time.time()
    host = ix.fetch()
    t3 = time.time()
    st = ix.build_stats
    print(f"n={n}: genome {t1-t0:.1f}s build {t2-t1:.1f}s (first pass {st.ms_first_pass:.0f} ms, outputs {st.ms_outputs:.0f} ms, chunks {st.chunks}, rounds {st.rounds}) fetch {t3-t2:.1f}s", flush=True)
    L = host.ref_seq_len
    # invariants of the index itself
    cp = host.cp_occ
    last = (L - 1) >> 6
    tot = cp[last, :4].astype(np.int64) + np.array([bin(int(x)).count("1") for x in cp[last, 4:]])
    print("  totals from last block", tot, "count diffs", np.diff(host.count), "sentinel", host.sentinel_index, flush=True)
    reads, _, _ = simulate.make_reads(g, 2000, seed=5)
    enc, cum = simulate.flatten_reads(reads)
    o = loader.OracleFMI(host)
    want = o.collect_smem(enc, cum)
    wcoord, woff = o.sa_lookup(want)
    ok = 0
    for i in range(min(len(want), 3000)):
        if woff[i + 1] > woff[i]:
            c = wcoord[woff[i]]; r = reads[want["rid"][i]]; m, e = want["m"][i], want["n"][i] + 1
            ok += np.array_equal(host.ref_0123[c:c + e - m], r[m:e])
    print(f"  oracle: {len(want)} smems, {len(wcoord)} coords, {ok} of first 3000 spell their seed", flush=True)
    b = capi.Batch(ix, len(reads), int(cum[-1]))
    sm, coord, off = b.seed(enc, cum)
    print(f"  gpu: {len(sm)} smems {len(coord)} coords; equal smems: {len(sm)==len(want) and all(np.array_equal(sm[f], want[f]) for f in ('rid','m','n','k','l','s'))}"
          f" equal coords: {np.array_equal(coord, wcoord)}", flush=True)
    if n <= 1_000_000_000:
        idx_dev = fmindex.build_fmindex(g, device="cuda:0", keep_ref=False)
        print("  vs torch builder: cp", bool((idx_dev.cp_occ.cpu().numpy().view(np.uint64) == host.cp_occ).all()),
              "ms", bool((idx_dev.sa_ms_byte.cpu().numpy() == host.sa_ms_byte).all()),
              "ls", bool((idx_dev.sa_ls_word.cpu().numpy().view(np.uint32) == host.sa_ls_word).all()),
              "sent", idx_dev.sentinel_index == host.sentinel_index, flush=True)
        del idx_dev
        torch.cuda.empty_cache()
    b.close(); ix.close()
    del g, host
