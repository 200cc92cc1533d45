"""Timeline analysis of a rocprofv3 --kernel-trace CSV of pipelined steps: for a window of the timed region, per kernel
family: calls, summed duration, and how much of the window has 0 / 1 / >= 2 kernels running; GEMM-active union.
usage: trace_analyze.py kernel_trace.csv [first_fraction last_fraction]"""
import csv, sys, re
f = sys.argv[1]
lo, hi = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (0.35, 0.6)
rows = list(csv.DictReader(open(f)))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
T0, T1 = rows[0]["s"], rows[-1]["e"]
a, b = T0 + lo * (T1 - T0), T0 + hi * (T1 - T0)
if lo >= 1:         # window = from the start of step `lo` to the start of step `hi` (steps = launches of lidar_depth_kernel)
    marks = [r["s"] for r in rows if "lidar_depth_kernel" in r["Kernel_Name"]]
    a, b = marks[int(lo)], marks[int(hi)]
    print(f"steps {int(lo)}..{int(hi)} of {len(marks)}: {(b - a) / (int(hi) - int(lo)) / 1e6:.3f} ms per step")
win = [r for r in rows if r["s"] >= a and r["s"] < b]
def fam(n):
    n = re.sub(r"^void ", "", n); n = re.sub(r"creste::", "", n)
    return re.sub(r"\(.*", "", n)[:48]
ev = []
for r in win:
    ev.append((r["s"], 1, r)); ev.append((r["e"], -1, r))
ev.sort(key=lambda x: (x[0], x[1]))
depth, last, hist = 0, ev[0][0], {}
gemm_depth, gemm_union, gemm_beside = 0, 0, 0
for t, d, r in ev:
    hist[min(depth, 3)] = hist.get(min(depth, 3), 0) + (t - last)
    if gemm_depth:
        gemm_union += t - last
        if depth > gemm_depth: gemm_beside += t - last
    last = t
    depth += d
    if "gemm" in r["Kernel_Name"]: gemm_depth += d
span = ev[-1][0] - ev[0][0]
print(f"window {span/1e6:.2f} ms, {len(win)} kernels; queues {sorted(set(r['Queue_Id'] for r in win))}")
print("  time with k kernels running:", {k: f"{v/1e6:.2f} ms ({100*v/span:.0f}%)" for k, v in sorted(hist.items())})
print(f"  GEMM active (union) {gemm_union/1e6:.2f} ms ({100*gemm_union/span:.0f}%), of which with another kernel beside it {gemm_beside/1e6:.2f} ms")
by = {}
for r in win:
    d = by.setdefault(fam(r["Kernel_Name"]), [0, 0]); d[0] += 1; d[1] += r["e"] - r["s"]
for k, (n, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"  {k:50s} {n:5d} calls {t/1e6:8.2f} ms  avg {t/n/1e3:8.1f} us")
