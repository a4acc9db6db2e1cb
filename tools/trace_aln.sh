export TMPDIR=/tmp
rm -rf gpurun_out/trace_aln; mkdir -p gpurun_out/trace_aln
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/trace_aln -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-pe --no-ert-leg > gpurun_out/trace_aln/log.txt 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/trace_aln/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "aln_" in r["Name"] or "sam_" in r["Name"]:
        print(r["Name"].split("(")[0][-40:], r["Calls"], "avg ms", round(float(r["AverageNs"]) / 1e6, 3))
PY
