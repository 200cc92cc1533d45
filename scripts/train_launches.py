"""Launches of the LAST training step in a rocprofv3 kernel trace (steps delimited by a once-per-step kernel), aggregated by
(kernel, workgroups): calls, total us, avg us, workgroups x threads, queue.  usage: train_launches.py kernel_trace.csv [marker] [top]"""
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else "multi_tensor_apply"
top = int(sys.argv[3]) if len(sys.argv) > 3 else 70
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
# the marker may fire several times per step (optimizer): steps = runs of markers separated by > 200 other launches
starts = [idx[0]]
for a, b in zip(idx, idx[1:]):
    if b - a > 200: starts.append(b)
a, b = starts[-2], starts[-1]
agg = collections.defaultdict(lambda: [0, 0.0]); qs = collections.defaultdict(set)
for r in rows[a:b]:
    k = re.sub(r"^void\s+", "", r["Kernel_Name"]).replace("creste::", "").split("(")[0]
    g = [int(r.get(f"Grid_Size_{c}", 1) or 1) for c in "XYZ"]; w = [int(r.get(f"Workgroup_Size_{c}", 1) or 1) for c in "XYZ"]
    wgs = (g[0] // max(w[0], 1)) * (g[1] // max(w[1], 1)) * (g[2] // max(w[2], 1))
    key = (k[:60], wgs, w[0] * w[1] * w[2])
    agg[key][0] += 1; agg[key][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    qs[key].add(r.get("Queue_Id", "?"))
T = sum(v[1] for v in agg.values())
print(f"one step: {b - a} launches, kernel time {T / 1e3:.2f} ms, wall {(int(rows[b]['Start_Timestamp']) - int(rows[a]['Start_Timestamp'])) / 1e6:.2f} ms")
for key, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{key[0]:60s} x{n:3d} {us:9.1f} us  avg {us / n:8.1f}  wgs {key[1]:6d} x {key[2]:4d}  q{','.join(sorted(qs[key]))}")
