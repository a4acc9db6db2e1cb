#!/bin/bash
# SMEM search over the compact Occ table (CpOcc2) against the reference-layout CP_OCC
for v in 1 0; do
BWAMS_CP2=$v timeout -k 10 300 python bench.py --no-cpu-baseline --no-pe --no-ert-leg --steps 3 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=j['stage_ms']; print('cp2 $v:', 'step', j['ms_per_step'], 'r1', s['smem_round1'], 'r2', s['smem_round2'], 'r3', s['smem_round3'], 'seed', s['seed_total'], 'frac', j['roofline']['frac'], 'index GB', round(j['config']['index_bytes']/2**30,2))"
done
