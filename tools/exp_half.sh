#!/bin/bash
# kernel times of the step at 500 k reads, to compare with the 1 M profile: which kernels do not shrink with the chunk
export TMPDIR=/tmp
OUT=gpurun_out/prof_half
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --reads 500000 --steps 2 --warmup 1 --no-cpu-baseline --no-pe --no-ert-leg > $OUT/trace.log 2>&1
tail -2 $OUT/trace.log | cut -c1-300
