"""Print the bwams kernels of a rocprofv3 kernel_stats.csv (name, calls, avg/min/max ms)."""
import csv, sys, re
for row in csv.DictReader(open(sys.argv[1])):
    if 'bwams' in row['Name'] or (len(sys.argv) > 2 and sys.argv[2] in row['Name']):
        m = re.search(r'(\w+(<[^>]*>)?)\(', row['Name'].replace('(anonymous namespace)::', ''))
        n = m.group(1) if m else row['Name'][:40]
        print(f"{n:34s} calls {row['Calls']:>3s} avg {float(row['AverageNs'])/1e6:8.3f} ms  min {float(row['MinNs'])/1e6:8.3f} max {float(row['MaxNs'])/1e6:8.3f}")
