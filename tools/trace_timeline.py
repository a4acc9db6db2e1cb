"""Kernel timeline of the LAST hot-path step in a rocprofv3 kernel trace (the launches from the last pack_reads on).

    python3 tools/trace_timeline.py <kernel_trace.csv> <out.txt> [min_ms]

One line per launch of at least min_ms (default 0.3): start, end, duration in ms from the step's first launch, queue, LDS, grid, name.
"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "pack_reads" in r["Kernel_Name"]]
last = rows[idx[-1]:] if idx else rows
t0 = int(last[0]["Start_Timestamp"])
min_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 0.3
with open(sys.argv[2], "w") as out:
    for r in last:
        s, e = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
        if e - s < min_ms:
            continue
        n = r["Kernel_Name"].replace("bwams::(anonymous namespace)::", "").replace("void ", "")[:80]
        out.write("%9.2f %9.2f %8.2f q%s lds%s g%s %s\n" % (s, e, e - s, r.get("Queue_Id", ""), r.get("LDS_Block_Size", ""), r.get("Grid_Size", ""), n))
