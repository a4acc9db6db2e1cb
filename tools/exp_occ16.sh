#!/bin/bash
# SMEM search over the occ16 table: LDS ring entries x workgroups per CU -> round kernel times (libraries built beforehand as _build/var_<ring>_<blocks>.so)
for v in 8_1 6_4 5_4; do
  export BWAMS_LIB=$PWD/bwa-mem-scale_amd/_build/var_$v.so
  timeout -k 10 120 python -m pytest tests/test_golden.py -m gpu -x -q 2>&1 | tail -1
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-pe --no-ert-leg --steps 4 --warmup 2 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=j['stage_ms']; print('var $v:', 'step', j['ms_per_step'], 'r1', s['smem_round1'], 'r2', s['smem_round2'], 'r3', s['smem_round3'], 'seed', s['seed_total'], 'frac', j['roofline']['frac'])"
done
