#!/bin/bash
# Collects the rocprofv3 evidence for round 4 at the metric's configuration (GRCh38-size synthetic genome, 1 M reads).
# Usage (on the GPU box, from the repo root):   bash profiles/run_profiles_r04.sh <tag>
# kernel-trace/stats and each PMC group run as separate passes (never combined).  Progress goes to $OUT/progress.txt.
set -e
TAG=${1:-r04}
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
rm -rf $OUT
mkdir -p $OUT
ARGS="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pe --no-ert-leg --no-hard-genome"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1
echo "trace done" > $OUT/progress.txt
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/pmc_fetch.log 2>&1
echo "fetch done" >> $OUT/progress.txt
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/pmc_write.log 2>&1
echo "write done" >> $OUT/progress.txt
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -- python3 $ARGS > $OUT/pmc_l2.log 2>&1
echo "l2 done" >> $OUT/progress.txt
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq -- python3 $ARGS > $OUT/pmc_sq.log 2>&1 || true
echo "sq done" >> $OUT/progress.txt
# the same with seeding over the ERT (bench.py --ert): kernel trace + the walk kernel's HBM bytes
EARGS="bench.py --ert --steps 2 --warmup 1 --no-cpu-baseline --no-pe --no-hard-genome"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_ert -- python3 $EARGS > $OUT/trace_ert.log 2>&1
echo "trace_ert done" >> $OUT/progress.txt
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_ert -- python3 $EARGS > $OUT/pmc_fetch_ert.log 2>&1
echo "fetch_ert done" >> $OUT/progress.txt
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_ert -- python3 $EARGS > $OUT/pmc_write_ert.log 2>&1
echo "write_ert done" >> $OUT/progress.txt
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq_ert -- python3 $EARGS > $OUT/pmc_sq_ert.log 2>&1 || true
echo "sq_ert done" >> $OUT/progress.txt
# Smith-Waterman kernels alone against the reference's own objects (oracle/_ref) on the host cores
python3 tests/bench_sw_kernels.py > $OUT/sw_kernels.jsonl 2> $OUT/sw_kernels.log || true
echo "sw done" >> $OUT/progress.txt
# the kernel traces are large: keep the stats and the two seeding kernels' trace rows only
for d in trace trace_ert; do
  f=$(find $OUT/$d -name '*kernel_trace.csv' | head -1)
  [ -n "$f" ] && { head -1 $f > $f.small; grep -E "smem_search_kernel|smem_bwd|ert_profile" $f >> $f.small; mv $f.small $f; }
done
find $OUT -name "*.csv" | head -50
