#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz (small committed vectors).

Inputs are seeded; expected outputs come from the CPU oracle.  For the banded-SW vectors
the script additionally REQUIRES agreement with the real reference objects in oracle/_ref
(built by oracle/Makefile from /root/reference) when they are present, and records in the
file whether that cross-check was made.  No reference source text is stored: only inputs
and expected integers.

    python tests/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "bwa-mem-scale_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

from bwams import fmindex, simulate  # noqa: E402
from oracle import loader  # noqa: E402
from util import OUT_FIELDS, make_pairs  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)

# ---- seeding: 30 kb genome, 160 reads -------------------------------------------------
g = simulate.make_genome(30000, seed=101, repeat_frac=0.2, repeat_len=150, n_families=3)
idx = fmindex.build_fmindex(g)
reads, pos, rev = simulate.make_reads(g, 160, seed=202)
enc, cum = simulate.flatten_reads(reads)
o = loader.OracleFMI(idx)
ctr = loader.Counters()
sm = o.collect_smem(enc, cum, counters=ctr)
coord, off = o.sa_lookup(sm, 500, counters=ctr)
np.savez_compressed(
    os.path.join(OUT, "seed_toy.npz"),
    genome=g, reads=reads,
    smem_rid=sm["rid"], smem_m=sm["m"], smem_n=sm["n"], smem_k=sm["k"], smem_l=sm["l"], smem_s=sm["s"],
    sa_coord=coord, sa_off=off,
    counters=np.array([ctr.n_ext, ctr.n_ext_blocks, ctr.n_sa_lookups, ctr.n_lf_steps] + list(ctr.n_smem)),
    count=idx.count, sentinel=np.int64(idx.sentinel_index), ref_seq_len=np.int64(idx.ref_seq_len))

# ---- banded SW: 700 tasks x 2 band widths ---------------------------------------------
pairs, ref, qer = make_pairs(700, seed=303)
REF = loader.ref_lib()
res = {}
for w in (100, 200):
    ours, cells = loader.bsw_pairs(pairs, ref, qer, w)
    if REF is not None:
        theirs = loader.ref_bsw(REF, "scalar", pairs, ref, qer, w)
        for f in OUT_FIELDS:
            assert np.array_equal(ours[f], theirs[f]), (w, f)
    res[w] = (ours, cells)
np.savez_compressed(
    os.path.join(OUT, "bsw_tasks.npz"),
    pairs=pairs.view(np.int32).reshape(len(pairs), -1), ref=ref, qer=qer,
    out_w100=np.stack([res[100][0][f] for f in OUT_FIELDS], axis=1),
    out_w200=np.stack([res[200][0][f] for f in OUT_FIELDS], axis=1),
    cells=np.array([res[100][1], res[200][1]]),
    checked_against_reference=np.array([REF is not None]))
print("golden vectors written to", OUT, "| reference cross-check:", REF is not None)
