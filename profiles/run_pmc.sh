#!/bin/bash
# Extra PMC passes for diagnosis.  Usage: bash profiles/run_pmc.sh <tag> <genome_mbp>
TAG=${1:-r01b}
GMBP=${2:-512}
export TMPDIR=/tmp
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
ARGS="bench.py --genome-mbp $GMBP --steps 1 --warmup 0 --no-cpu-baseline"
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  timeout -k 5 240 rocprofv3 --pmc $line --output-format csv -d $OUT/p$i -- python3 $ARGS > $OUT/p$i.log 2>&1 || echo "pass $i failed: $line"
  echo "pass $i done: $line" >> $OUT/progress.txt
done <<'LIST'
TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum
TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum
TCP_TOTAL_ACCESSES_sum TCP_TOTAL_CACHE_ACCESSES_sum
TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum
TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum
SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_BRANCH
SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
TCC_REQ_sum TCC_READ_sum
GRBM_GUI_ACTIVE
LIST
python3 - <<'PY'
import csv, glob, collections, os, sys
out = os.environ.get("OUT", "")
for f in sorted(glob.glob("gpurun_out/pmc_*/p*/*/*_counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if any(x in k for x in ("smem_search", "seed_strategy", "sa_lookup", "bsw_kernel")):
            short = "R1" if "<true>" in k else "R2" if "<false>" in k else "R3" if "seed_strategy" in k else "SAL" if "sa_lookup" in k else "BSW"
            print(f.split("/")[2], short, {c: round(x[-1]) for c, x in v.items()})
PY
