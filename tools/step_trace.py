"""Print one step's kernel sequence from a rocprofv3 kernel-trace CSV (anchor: k_adj_pack launches)."""
import csv, sys
path = sys.argv[1]
thresh_us = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_adj_pack' in r['Kernel_Name']]
a, b = idx[-3], idx[-2]
tot = 0
for r in rows[a:b]:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot += d
    if d >= thresh_us:
        print(f"{r['Kernel_Name'][:64]:64s} grid={r['Grid_Size_X']:>8s},{r['Grid_Size_Y']:>4s} wg={r['Workgroup_Size_X']:>4s} us={d:9.2f}")
print(f"{b - a} launches, {tot:.1f} us of kernel time")
