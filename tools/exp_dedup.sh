timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_chain.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -2
timeout -k 10 300 python bench.py --steps 3 --no-cpu-baseline --no-pe --no-ert-leg 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['stage_ms'])"
