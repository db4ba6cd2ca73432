"""Prints the last kernels of a rocprofv3 kernel-trace CSV as a timeline (name, start offset, duration, gap)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tail = rows[-n:]
t0 = int(tail[0]["Start_Timestamp"]); prev_end = t0
for r in tail:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap {(s - prev_end) / 1e3:7.1f}  grid {r.get('Grid_Size_X', '?'):>8} wg {r.get('Workgroup_Size_X', '?'):>4}  {r['Kernel_Name'][:90]}")
    prev_end = e
