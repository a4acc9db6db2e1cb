#!/bin/bash
# PMC passes over the ERT seeding kernels at GRCh38 size (tools/ert_scale.py); separate passes, never with a trace domain.
export TMPDIR=/tmp
OUT=gpurun_out/pmc_ert
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES --output-format csv -d $OUT/sq -- python3 tools/ert_scale.py > $OUT/sq.log 2>&1
echo "sq done" > $OUT/progress.txt
rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/mem -- python3 tools/ert_scale.py > $OUT/mem.log 2>&1
echo "mem done" >> $OUT/progress.txt
python3 - <<'PY'
import csv, glob, collections
for d in ("sq", "mem"):
    files = glob.glob(f"gpurun_out/pmc_ert/{d}/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][-60:]
            if "ert_" not in k and "smem_search" not in k: continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
            n[(k, r["Counter_Name"])] += 1
    for k, v in agg.items():
        print(d, k, {c: round(x / n[(k, c)], 1) for c, x in v.items()}, "launches", max(n[(k, c)] for c in v))
PY
