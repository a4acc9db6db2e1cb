# Experiment (profiles/r03_notes.md 96): lower hand-over thresholds once the read queue has run dry
set -e
for v in "24 8 16" "32 8 12" "24 6 14" "24 10 12" "24 8 12"; do
  set -- $v
  BWAMS_BWD_DRY_MIN_LIST=$1 BWAMS_BWD_DRY_COLS=$2 BWAMS_BWD_DRY_LATE_LIST=$3 timeout -k 10 280 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-pe --no-ert-leg --no-hard-genome > gpurun_out/bv.json 2> gpurun_out/bv.err
  python -c "
import json
d=json.loads(open('gpurun_out/bv.json').read().strip().splitlines()[-1]); s=d['stage_ms']
print('dry $v', d['ms_per_step'], s['smem_round1'], s['smem_round2'], s['seed_total'], d['roofline']['frac'])"
done
