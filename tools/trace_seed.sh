#!/bin/bash
# Durations of the seeding kernels for one switch set (round 4 lab):  bash tools/trace_seed.sh <tag> [ENV=VAL ...]
TAG=$1; shift
export TMPDIR=/tmp
OUT=gpurun_out/trs_$TAG
rm -rf $OUT; mkdir -p $OUT
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p -- python3 tools/seed_lab.py --steps 3 --warmup 1 $TAG: > $OUT/log.txt 2>&1
grep "\[lab\]" $OUT/log.txt | tail -1
python3 tools/kstats.py $(find $OUT/p -name '*kernel_stats.csv' | head -1) | grep -E "smem_|seed_strategy|sa_lookup|round2_work"
rm -rf $OUT/p
