#!/bin/bash
# Re-collects only what a change to the banded-SW kernels invalidates: the kernel trace, the SQ instruction counters and the bench line.
# The other passes (HBM bytes and L2 counters of the SMEM kernels, the paired-end and ERT traces) are taken over from an earlier run directory
# by the caller (cp -r gpurun_out/prof_<old>/{pmc_fetch,pmc_write,pmc_l2,trace_pe,trace_ert,pmc_*_ert} gpurun_out/prof_<tag>/).
# Usage (on the GPU box, from the repo root):   bash profiles/run_profiles_r02_bsw.sh <tag>
set -e
TAG=${1:-r02}
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
rm -rf $OUT
mkdir -p $OUT
ARGS="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pe --no-ert-leg"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1
echo "trace done" > $OUT/progress.txt
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq -- python3 $ARGS > $OUT/pmc_sq.log 2>&1 || true
echo "sq done" >> $OUT/progress.txt
python3 bench.py > $OUT/bench.json 2> $OUT/bench.log
echo "bench done" >> $OUT/progress.txt
