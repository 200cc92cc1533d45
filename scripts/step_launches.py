"""Every launch of ONE steady-state step from a rocprofv3 kernel trace, in launch order: duration, workgroups, idle gap in front.
usage: step_launches.py kernel_trace.csv [marker]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else "lidar_depth_kernel"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
prev_end = None
for r in rows[a:b]:
    k = re.sub(r"^void\s+", "", r["Kernel_Name"]).replace("creste::", "").split("(")[0]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    g = [int(r.get(f"Grid_Size_{c}", 1) or 1) for c in "XYZ"]; w = [int(r.get(f"Workgroup_Size_{c}", 1) or 1) for c in "XYZ"]
    wgs = (g[0] // max(w[0], 1)) * (g[1] // max(w[1], 1)) * (g[2] // max(w[2], 1))
    gap = 0 if prev_end is None else (s - prev_end) / 1e3
    prev_end = max(e, prev_end or 0)
    print(f"{k[:56]:56s} {(e - s) / 1e3:8.1f} us  wgs {wgs:6d} x {w[0] * w[1] * w[2]:4d}  gap {gap:6.1f}")
