#!/bin/bash
# packed banded SW: waves per SIMD forced by amdgpu_waves_per_eu (libraries built beforehand as _build/var_w<N>.so; the default build has 4)
for v in default w5 w6; do
  if [ $v = default ]; then unset BWAMS_LIB; else export BWAMS_LIB=$PWD/bwa-mem-scale_amd/_build/var_$v.so; fi
  timeout -k 10 200 python tools/dbg_bsw.py 2>&1 | tail -1
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-pe --no-ert-leg --no-hard-genome --steps 4 --warmup 2 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=j['stage_ms']; print('$v:', 'step', j['ms_per_step'], {k: s[k] for k in ('ext_left','ext_right','ext_total')})"
done
