# Experiment (profiles/r03_notes.md 89): the ERT walk's groups of read positions from a cursor instead of round robin
set -e
BWAMS_ERT_TICKET=1 timeout -k 10 200 python -m pytest tests/test_gpu_ert.py -x -q 2>&1 | tail -n 1
for v in "0 32" "1 32" "1 8" "1 12" "1 16"; do
  set -- $v
  BWAMS_ERT_TICKET=$1 BWAMS_ERT_GRID=$2 timeout -k 10 280 python bench.py --ert --steps 3 --warmup 1 --no-cpu-baseline --no-pe --no-hard-genome > gpurun_out/et.json 2> gpurun_out/et.err
  python -c "
import json
d=json.loads(open('gpurun_out/et.json').read().strip().splitlines()[-1]); s=d['stage_ms']
print('ticket $1 grid $2', d['ms_per_step'], s.get('ert_walk'), s.get('seed_total'))"
done
