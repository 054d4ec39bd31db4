"""Top rows of a rocprofv3 kernel_stats.csv: calls, average us, name.   python tools/top_kernels.py <csv> [rows]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 10]:
    print(f"{int(r['Calls']):5d}  {float(r['AverageNs']) / 1e3:9.1f} us  {r['Name'][:80]}")
