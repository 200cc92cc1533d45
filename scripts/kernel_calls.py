"""List the launches of one kernel in the timed region of a rocprofv3 kernel trace (first f16 patch kernel onwards).
usage: kernel_calls.py trace_kernel_trace.csv name_substring [max]"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
first = next(i for i, r in enumerate(rows) if "conv_patch" in r["Kernel_Name"])
n = 0
for r in rows[first:]:
    if sys.argv[2] in r["Kernel_Name"]:
        print(f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:9.1f} us  grid {r.get('Grid_Size', '?')} wg {r.get('Workgroup_Size', '?')}")
        n += 1
        if n >= int(sys.argv[3]) if len(sys.argv) > 3 else 12: break
