import sys, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p_ in (ROOT, os.path.join(ROOT, 'bwa-mem-scale_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p_)
import numpy as np
from bwams import simulate, fmindex
from oracle import loader
def make(n_mono=3000, div=0.02, L=30000, err=0.01, seed=5):
    rng = np.random.default_rng(seed)
    cons = rng.integers(0, 4, size=171, dtype=np.uint8)
    arr = np.tile(cons, n_mono)
    mut = rng.random(arr.shape) < div
    arr[mut] = (arr[mut] + rng.integers(1, 4, size=int(mut.sum()), dtype=np.uint8)) & 3
    g = np.concatenate([rng.integers(0, 4, size=50000, dtype=np.uint8), arr, rng.integers(0, 4, size=50000, dtype=np.uint8)])
    st = 50000 + 171 * 100 + 17
    read = g[st:st + L].copy()
    e = rng.random(L) < err
    read[e] = (read[e] + rng.integers(1, 4, size=int(e.sum()), dtype=np.uint8)) & 3
    return g, read
def gpu(g, read, idx):
    import torch
    torch.cuda.init()
    from bwams import capi
    ix = capi.Index.from_host(idx, 0)
    c = np.zeros(1, capi.CONTIG_DTYPE); c["len"] = len(g); ix.set_contigs(c)
    enc, cum = simulate.flatten_reads([read])
    b = capi.Batch(ix, 1, len(read), max_smem=100000, max_sa=1 << 20)
    b.seed_upload(enc, cum)
    t0 = time.time(); b.seed_run(capi.default_seed_opt(), with_sa=True); b.sync(); print('gpu seed', time.time() - t0, flush=True)
    t0 = time.time(); nc, ns = b.chain_run(capi.default_mem_opt()); b.sync(); print('gpu chain', nc, ns, time.time() - t0, flush=True)
    return b.chain_fetch()


if __name__ == "__main__":
    g, read = make(n_mono=int(sys.argv[1]), div=float(sys.argv[2]), L=int(sys.argv[3]), err=float(sys.argv[4]))
    t0=time.time(); idx = fmindex.build_fmindex(g); print('index', time.time()-t0)
    enc, cum = simulate.flatten_reads([read])
    o = loader.OracleFMI(idx)
    so = loader.default_seed_opt()
    if os.environ.get('LONGREAD_GPU'):
        gch, gsd, goff = gpu(g, read, idx)
    t0=time.time(); sm = o.collect_smem(enc, cum, so); coord, off = o.sa_lookup(sm, so.max_occ); print('seeds', len(sm), len(coord), time.time()-t0)
    t0=time.time(); ref = np.concatenate([g, (3 - g[::-1]).astype(np.uint8)]); mo = loader.default_mem_opt(); ch, sd, choff = loader.chain_seeds(sm, coord, off, cum, len(g), ref_string=ref, enc=enc, opt=mo); print('chains', len(ch), len(sd), time.time()-t0)
    if os.environ.get('LONGREAD_GPU'):
        print('equal', len(gch) == len(ch), all(np.array_equal(gch[f], ch[f]) for f in ('pos', 'n', 'rid', 'w_kept_alt', 'first')), np.array_equal(goff, choff))
