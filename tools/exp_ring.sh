#!/bin/bash
# SMEM search: LDS ring entries of the interval list x workgroups per CU (launch bounds) -> round-1 / round-2 kernel times
cd bwa-mem-scale_amd/csrc
BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -I../../include"
for v in "8 1" "6 4" "5 4" "4 4"; do
  set -- $v
  /opt/rocm/bin/hipcc $BASE -DBWAMS_PREV_LDS=$1 -DBWAMS_SEARCH_MIN_BLOCKS=$2 -c fmi_seed.hip -o ../_build/fmi_seed.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../_build/libbwams.so ../_build/*.o
  (cd ../.. && timeout -k 10 300 python bench.py --no-cpu-baseline --no-pe --no-ert-leg --steps 3 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=j['stage_ms']; print('ring $1 blocks/CU>=$2:', 'step', j['ms_per_step'], 'r1', s['smem_round1'], 'r2', s['smem_round2'], 'seed', s['seed_total'], 'frac', j['roofline']['frac'])")
done
