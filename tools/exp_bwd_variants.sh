set -e
BWAMS_BWD_MIN_LIST=200 BWAMS_BWD_COLS=2 BWAMS_BWD_LATE_LIST=1 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q > gpurun_out/t_late.log 2>&1 || { tail -n 20 gpurun_out/t_late.log; exit 1; }
tail -n 1 gpurun_out/t_late.log
BWAMS_BWD_MIN_LIST=5 BWAMS_BWD_COLS=4 BWAMS_BWD_LATE_LIST=2 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q > gpurun_out/t_late2.log 2>&1 || { tail -n 20 gpurun_out/t_late2.log; exit 1; }
tail -n 1 gpurun_out/t_late2.log
for v in "40 16 12" "40 12 12" "40 24 8" "28 16 12" "1000 16 12" "40 10 14"; do
  set -- $v
  BWAMS_VERBOSE=1 BWAMS_BWD_MIN_LIST=$1 BWAMS_BWD_COLS=$2 BWAMS_BWD_LATE_LIST=$3 timeout -k 10 280 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pe --no-ert-leg --no-hard-genome > gpurun_out/bv.json 2> gpurun_out/bv.err
  python -c "
import json
d=json.loads(open('gpurun_out/bv.json').read().strip().splitlines()[-1]); s=d['stage_ms']
print('$v', d['ms_per_step'], s['smem_round1'], s['smem_round2'], s['smem_round3'], s['seed_total'], d['roofline']['frac'])"
  grep bwd_wave gpurun_out/bv.err | head -1 | cut -c1-200
done
