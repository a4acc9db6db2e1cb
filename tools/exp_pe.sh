timeout -k 10 400 python -m pytest tests/test_gpu_pair.py tests/test_gpu_aln.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -2
timeout -k 10 400 python bench.py --steps 3 --no-cpu-baseline --no-ert-leg 2>/dev/null > gpurun_out/exp_pe.json
python3 - <<'PY'
import json
j = json.loads(open("gpurun_out/exp_pe.json").read().strip().splitlines()[-1])
print("sam_side ms_run", j["sam_side"]["ms_run"], "Maln/s", j["sam_side"]["Malignments_per_s"], "mark_se", j["sam_side"]["ms_mark_primary_se"],
      "| step", j["value"], j["ms_per_step"], "| PE", j["paired_end"]["value"], j["paired_end"]["ms_per_batch"], j["paired_end"]["ms_pair_run"])
PY
