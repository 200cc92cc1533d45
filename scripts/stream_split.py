"""Per-queue (= per HIP stream) kernel tables of ONE steady-state step from a rocprofv3 kernel trace: which stream is the critical path.
usage: stream_split.py kernel_trace.csv marker_kernel [top]"""
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2]; top = int(sys.argv[3]) if len(sys.argv) > 3 else 14
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
qcol = "Queue_Id" if "Queue_Id" in rows[0] else ("Stream_Id" if "Stream_Id" in rows[0] else None)
print("columns:", list(rows[0].keys()))
step = rows[a:b]
t0, t1 = int(step[0]["Start_Timestamp"]), int(step[-1]["End_Timestamp"])
print(f"one step: {len(step)} launches, wall {(t1 - t0) / 1e6:.2f} ms")
byq = collections.defaultdict(list)
for r in step:
    byq[r[qcol] if qcol else "?"].append(r)
for q, rs in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs)
    span = (int(rs[-1]["End_Timestamp"]) - int(rs[0]["Start_Timestamp"])) / 1e6
    print(f"\nqueue {q}: {len(rs)} launches, kernel time {busy / 1e6:.2f} ms, first start +{(int(rs[0]['Start_Timestamp']) - t0) / 1e6:.2f} ms, last end +{(int(rs[-1]['End_Timestamp']) - t0) / 1e6:.2f} ms (span {span:.2f})")
    tot = collections.defaultdict(float); cnt = collections.Counter()
    for r in rs:
        k = re.sub(r"^void\s+", "", r["Kernel_Name"]).replace("creste::", "")
        m = re.match(r"([\w:]+(?:<[^(]*>)?)", k); k = m.group(1) if m else k[:60]
        tot[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); cnt[k] += 1
    for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:top]:
        print(f"   {k[:64]:64s} {cnt[k]:4d} {v / 1e6:8.3f} ms")
