for d in 0 1; do
BWAMS_DEBUG=$d timeout -k 10 300 python bench.py --genome-mbp 512 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('debug=$d', d['stage_ms'])"
done
