#!/usr/bin/env python3
"""Kernel-level throughput of the two Smith-Waterman kernels, with the REAL reference objects
(oracle/_ref: bandedSWA.cpp / ksw.cpp compiled from the reference tree) timed beside them on the
host cores as cpu_baseline kind "reference".  Prints one JSON line per kernel.

    python tests/bench_sw_kernels.py [--tasks 400000]
"""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "bwa-mem-scale_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402

from bwams import capi, fmindex, simulate  # noqa: E402
from oracle import loader  # noqa: E402


def ext_tasks(n, seed):
    """extension tasks shaped like 150-bp read extensions: query 5..130, target = query + gap window"""
    rng = np.random.default_rng(seed)
    ql = rng.integers(5, 131, size=n)
    tl = ql + np.minimum(np.maximum(ql - 5, 1), 200)
    qo = np.r_[0, np.cumsum(ql)]; to = np.r_[0, np.cumsum(tl)]
    qer = rng.integers(0, 4, size=int(qo[-1]), dtype=np.uint8)
    ref = rng.integers(0, 4, size=int(to[-1]), dtype=np.uint8)
    # target = mutated query followed by random bases
    for i in range(n):
        q = qer[qo[i]:qo[i + 1]].copy()
        m = rng.random(len(q)) < 0.03
        q[m] = (q[m] + 1) & 3
        ref[to[i]:to[i] + len(q)] = q
    pairs = np.zeros(n, dtype=capi.SEQPAIR_DTYPE)
    pairs["idr"], pairs["idq"], pairs["len1"], pairs["len2"] = to[:-1], qo[:-1], tl, ql
    pairs["h0"] = rng.integers(19, 120, size=n)
    pairs["id"] = np.arange(n)
    return pairs, ref, qer


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tasks", type=int, default=400_000)
    ap.add_argument("--local-tasks", type=int, default=60_000)
    args = ap.parse_args()
    capi.lib()
    g = simulate.make_genome(100000, seed=1)
    ix = capi.Index.from_host(fmindex.build_fmindex(g), 0)
    b = capi.Batch(ix, 8, 1200)
    REF = loader.ref_lib()
    threads = min(16, os.cpu_count() or 1)

    # ---- banded SW extension ----
    pairs, ref, qer = ext_tasks(args.tasks, 7)
    b.bsw_upload(pairs, ref, qer)
    b.bsw_run(100); b.sync()
    ms = []
    for _ in range(5):
        b.bsw_run(100); b.sync(); ms.append(b.stats().ms_bsw)
    cells = b.stats().bsw_cells
    line = {"kernel": "banded SW extension (bwams_bsw_extend)", "tasks": args.tasks, "ms": round(float(np.mean(ms)), 3),
            "Mtasks_per_s": round(args.tasks / np.mean(ms) / 1e3, 2), "Gcells_per_s": round(cells / np.mean(ms) / 1e6, 2)}
    if REF is not None:
        ns = min(args.tasks, 100_000)
        chunks = np.array_split(np.arange(ns), threads)

        def work(ixs):
            loader.ref_bsw(REF, "vec16", pairs[ixs], ref, qer, 100)
        t0 = time.perf_counter()
        with ThreadPoolExecutor(threads) as ex:
            list(ex.map(work, chunks))
        dt = time.perf_counter() - t0
        line["cpu_baseline"] = {"value": round(ns / dt / 1e6, 4), "unit": "Mtasks/s", "cores": threads, "kind": "reference",
                                "sample": f"BandedPairWiseSW::getScores16 ({REF.isa}) of oracle/_ref on {ns} of the same tasks"}
    print(json.dumps(line), flush=True)

    # ---- mate-rescue local SW ----
    from util import make_local_cases
    base = make_local_cases(2000, seed=3, qmax=151, tmax=900)
    reps = (args.local_tasks + len(base) - 1) // len(base)
    cases = (base * reps)[:args.local_tasks]
    xtra = loader.KSW_XSUBO | loader.KSW_XSTART | loader.KSW_XBYTE | 19
    kp = np.zeros(len(cases), dtype=capi.SEQPAIR_DTYPE)
    ql = np.array([len(q) for q, _ in cases]); tl = np.array([len(t) for _, t in cases])
    kp["idr"], kp["idq"], kp["len1"], kp["len2"], kp["h0"] = np.r_[0, np.cumsum(tl)][:-1], np.r_[0, np.cumsum(ql)][:-1], tl, ql, xtra
    kr = np.concatenate([t for _, t in cases]); kq = np.concatenate([q for q, _ in cases])
    b.ksw_align(kp, kr, kq)
    ms = []
    for _ in range(3):
        b.ksw_align(kp, kr, kq); ms.append(b.stats().ms_ksw)
    cells = int((((ql + 15) // 16 * 16) * tl).sum())
    line = {"kernel": "mate-rescue local SW (bwams_ksw_align, both passes)", "tasks": len(cases), "ms": round(float(np.mean(ms)), 3),
            "Mtasks_per_s": round(len(cases) / np.mean(ms) / 1e3, 3), "Gcells_per_s_first_pass": round(cells / np.mean(ms) / 1e6, 2)}
    if REF is not None:
        ns = min(len(cases), 8000)

        def workk(ixs):
            for i in ixs:
                loader.ref_ksw_align2(REF, cases[i][0], cases[i][1], xtra)
        t0 = time.perf_counter()
        with ThreadPoolExecutor(threads) as ex:
            list(ex.map(workk, np.array_split(np.arange(ns), threads)))
        dt = time.perf_counter() - t0
        line["cpu_baseline"] = {"value": round(ns / dt / 1e6, 5), "unit": "Mtasks/s", "cores": threads, "kind": "reference",
                                "sample": f"ksw_align2 (SSE2, oracle/_ref) on {ns} of the same tasks; python call overhead included"}
    print(json.dumps(line), flush=True)
    b.close(); ix.close()


if __name__ == "__main__":
    main()
