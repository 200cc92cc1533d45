"""Kernel table of ONE steady-state step from a rocprofv3 kernel trace: the launches between the last two occurrences
of a once-per-step kernel (default svf_kernel).  usage: last_step_stats.py kernel_trace.csv [marker] [top]"""
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else "svf_kernel"
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
tot = collections.defaultdict(float); cnt = collections.Counter()
for r in rows[a:b]:
    k = re.sub(r"^void\s+", "", r["Kernel_Name"]).replace("creste::", "")
    m = re.match(r"([\w:]+(?:<[^(]*>)?)", k)
    k = m.group(1) if m else k[:60]
    tot[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); cnt[k] += 1
T = sum(tot.values())
wall = int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])
print(f"one step: {b - a} launches, GPU kernel time {T / 1e6:.2f} ms, wall {wall / 1e6:.2f} ms\n\n| kernel | calls | ms | % | avg us |\n|---|---|---|---|---|")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:top]:
    print(f"| `{k[:70]}` | {cnt[k]} | {v / 1e6:.3f} | {100 * v / T:.1f} | {v / cnt[k] / 1e3:.1f} |")
if len(sys.argv) > 4:      # every launch of the kernels whose name contains argv[4], in launch order (us, grid)
    print()
    for r in rows[a:b]:
        if sys.argv[4] in r["Kernel_Name"]:
            k = re.sub(r"^void\s+", "", r["Kernel_Name"]).replace("creste::", "").split("(")[0]
            print(f"{k[:48]:48s} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:9.1f} us  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?'))}")
