"""Average a rocprofv3 --pmc counter per kernel from <dir>/*counter_collection.csv.
usage: python tools/pmc_summary.py <csv> [name-substring ...]"""
import csv, sys, collections
path = sys.argv[1]
subs = sys.argv[2:]
acc = collections.defaultdict(lambda: [0, 0.0])
with open(path) as f:
    for r in csv.DictReader(f):
        k = (r["Counter_Name"], r["Kernel_Name"])
        if subs and not any(s in r["Kernel_Name"] for s in subs):
            continue
        acc[k][0] += 1
        acc[k][1] += float(r["Counter_Value"])
print("counter,kernel,dispatches,avg_value")
for (c, k), (n, v) in sorted(acc.items()):
    print(f'{c},"{k[:90]}",{n},{v / n:.1f}')
