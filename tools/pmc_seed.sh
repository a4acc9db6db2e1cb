#!/bin/bash
# Issue counters of the seeding kernels for one library build / switch set (round 4 lab).
#   bash tools/pmc_seed.sh <tag> [ENV=VAL ...]      (BWAMS_LIB selects another build)
# One rocprofv3 --pmc pass over tools/seed_lab.py (1 warm-up + 1 seeding pass), then per kernel: wave-instructions per launch.
TAG=$1; shift
export TMPDIR=/tmp
OUT=gpurun_out/pmcs_$TAG
rm -rf $OUT; mkdir -p $OUT
for kv in "$@"; do export "$kv"; done
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/p -- python3 tools/seed_lab.py --steps 1 --warmup 1 $TAG: > $OUT/log.txt 2>&1
grep "\[lab\]" $OUT/log.txt | tail -1
python3 - $OUT <<'PY'
import csv, glob, collections, sys
for f in glob.glob(sys.argv[1] + "/p/*/*_counter_collection.csv"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if "smem_" in k or "seed_strategy" in k:
            name = k.replace("bwams::(anonymous namespace)::", "").replace("void ", "")[:34]
            last = {c: x[-1] for c, x in v.items()}          # the timed pass
            print(f"{name:34s} " + "  ".join(f"{c[3:]}={last[c]/1e9:.3f}G" for c in sorted(last)))
PY
rm -rf $OUT/p
