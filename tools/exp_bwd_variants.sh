# Experiment (profiles/r03_notes.md 86-87): hand-over thresholds of the SMEM search (list at the forward end, columns, list alive then)
set -e
for v in "40 24 8" "48 32 8" "32 24 8" "40 24 4" "40 16 12" "64 24 8" "0 0 0"; do
  set -- $v
  BWAMS_BWD_MIN_LIST=$1 BWAMS_BWD_COLS=$2 BWAMS_BWD_LATE_LIST=$3 timeout -k 10 280 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-pe --no-ert-leg --no-hard-genome > gpurun_out/bv.json 2> gpurun_out/bv.err
  python -c "
import json
d=json.loads(open('gpurun_out/bv.json').read().strip().splitlines()[-1]); s=d['stage_ms']
print('$v', d['ms_per_step'], s['smem_round1'], s['smem_round2'], s['smem_round3'], s['seed_total'], d['roofline']['frac'])"
done
