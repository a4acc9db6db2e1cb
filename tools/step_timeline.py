"""Kernel timeline of one timed step from a rocprofv3 --kernel-trace CSV: start, end, duration (ms from the step's first kernel).
usage: python tools/step_timeline.py <kernel_trace.csv> [from_ms] [to_ms] [which_step]"""
import csv, re, sys
f = sys.argv[1]
lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
hi = float(sys.argv[3]) if len(sys.argv) > 3 else 1e9
which = int(sys.argv[4]) if len(sys.argv) > 4 else 2
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "smem_search_kernel<true>" in r["Kernel_Name"]]
i0 = idx[which]
i1 = idx[which + 1] if which + 1 < len(idx) else len(rows)
t0 = int(rows[i0]["Start_Timestamp"])
def short(n):
    n = n.replace("bwams::", "").replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([A-Za-z0-9_:]+(<[^(]*>)?)", n)
    return (m.group(1) if m else n)[:50]
for r in rows[i0:i1]:
    s = (int(r["Start_Timestamp"]) - t0) / 1e6
    e = (int(r["End_Timestamp"]) - t0) / 1e6
    if s < lo or s > hi or e - s < 0.12:
        continue
    print("%8.2f %8.2f %7.2f  %s lds=%s grid=%s" % (s, e, e - s, short(r["Kernel_Name"]), r["LDS_Block_Size"], r["Grid_Size_X"]))
