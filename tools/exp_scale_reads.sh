#!/bin/bash
# stage times against the chunk size: what is fixed, what scales
for n in 125000 250000 500000 1000000 2000000; do
timeout -k 10 300 python bench.py --reads $n --chunk-reads $n --no-cpu-baseline --no-pe --no-ert-leg --steps 3 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=j['stage_ms']
print('reads',$n,'step',j['ms_per_step'],'r1',s['smem_round1'],'r2',s['smem_round2'],'r3',s['smem_round3'],'sal',s['sa_lookup'],'seed',s['seed_total'],'chain',s['chain'],'ext',s['ext_total'],'dedup',s['dedup'])"
done
