"""Print one step's kernel sequence from a rocprofv3 kernel-trace CSV.  Anchor: the step's first kernel — the persistent
level-0 forward (k_level0_fwd) when the plan runs it, else k_adj_pack."""
import csv, sys
path = sys.argv[1]
thresh_us = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_level0_fwd' in r['Kernel_Name']]
if len(idx) < 3:
    idx = [i for i, r in enumerate(rows) if 'k_adj_pack' in r['Kernel_Name']]
a, b = idx[-3], idx[-2]
tot = 0
for r in rows[a:b]:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot += d
    if d >= thresh_us:
        print(f"{r['Kernel_Name'][:64]:64s} grid={r['Grid_Size_X']:>8s},{r['Grid_Size_Y']:>4s} wg={r['Workgroup_Size_X']:>4s} us={d:9.2f}")
span = (int(rows[b]['Start_Timestamp']) - int(rows[a]['Start_Timestamp'])) / 1e3
print(f"{b - a} launches, {tot:.1f} us of kernel time ({span:.1f} us between two steps' first kernels UNDER THE PROFILER, "
      f"which serialises graph replays; the unprofiled step time is bench.py's ms_per_step)")
