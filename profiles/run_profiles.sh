#!/bin/bash
# Collects the rocprofv3 evidence for one round.  Usage (on the GPU box, from the repo root):
#   bash profiles/run_profiles.sh <tag> [genome_mbp]
# kernel-trace/stats and each PMC group run as separate passes (never combined).
set -e
TAG=${1:-r01}
GMBP=${2:-1000}
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
ARGS="bench.py --genome-mbp $GMBP --steps 2 --warmup 1 --no-cpu-baseline --no-pe"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1
# a second trace that also runs the paired-end leg (its kernels are summarised separately)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_pe -- python3 bench.py --genome-mbp $GMBP --steps 1 --warmup 0 --no-cpu-baseline > $OUT/trace_pe.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -- python3 $ARGS > $OUT/pmc_l2.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq -- python3 $ARGS > $OUT/pmc_sq.log 2>&1 || true
find $OUT -name "*.csv" | head -50
