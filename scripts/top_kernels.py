"""Print the top kernels of a rocprofv3 kernel_stats.csv. usage: top_kernels.py stats.csv [n]"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
    print(f"{r['Name'][:72]:72s} calls {r['Calls']:>5s}  avg {float(r['AverageNs']) / 1e3:9.1f} us  total {float(r['TotalDurationNs']) / 1e6:8.2f} ms")
