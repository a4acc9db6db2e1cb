#!/bin/bash
# kernel trace + SQ counters of one bench step for a given library build; prints the banded-SW kernels' lines
# usage: bash tools/prof_bsw.sh <tag> [env assignments...]
TAG=$1; shift
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
for kv in "$@"; do export "$kv"; done
ARGS="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pe --no-ert-leg"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
echo "== $TAG kernel stats (bsw / ext)"; head -1 $f; grep -E "bsw_|ext_" $f | head -12
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq -- python3 $ARGS > $OUT/pmc_sq.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/pmc_sq/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for fn in f:
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"]
        if "bsw_" in k:
            agg[k.split("(")[0][:60]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in agg.items():
    print(k, {c: f"{x:.3g}" for c, x in v.items()})
PY
