export TMPDIR=/tmp
rm -rf gpurun_out/trace_chain; mkdir -p gpurun_out/trace_chain
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_chain -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-pe --no-ert-leg > gpurun_out/trace_chain/log.txt 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/trace_chain/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "chain_" in r["Kernel_Name"] or "dedup_wave" in r["Kernel_Name"] or "ext_select" in r["Kernel_Name"]]
t0 = min(int(r["Start_Timestamp"]) for r in rows)
for r in rows[len(rows)//2:]:
    print(r["Kernel_Name"].split("(")[0][-34:], "grid", r.get("Grid_Size", r.get("Grid_Size_X")), "lds", r.get("LDS_Block_Size"), "start %.2f dur %.2f ms" % ((int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
PY
