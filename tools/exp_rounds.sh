#!/bin/bash
# extension rounds before everything undecided is extended at once (BWAMS_EXT_MAX_ROUNDS): ext_total, tasks, step
for r in 1 2 3 6; do
BWAMS_EXT_MAX_ROUNDS=$r timeout -k 10 300 python bench.py --no-cpu-baseline --no-pe --no-ert-leg --steps 3 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('max_rounds',$r, 'step', j['ms_per_step'], 'ext_total', j['stage_ms']['ext_total'], 'tasks', j['config']['bsw_tasks'], 'rounds', j['config']['ext_rounds'], 'regs', j['config']['final_regions'])"
done
