"""Per-step kernel table of the TIMED region of bench.py from a rocprofv3 kernel trace: everything launched after the
first f16x3 patch kernel (the BN calibration pass before it runs on the exact-fp32 engine).
usage: steps_only_stats.py trace_kernel_trace.csv n_steps"""
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1]))); n = float(sys.argv[2])
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = next(i for i, r in enumerate(rows) if "conv_patch" in r["Kernel_Name"])
# the step's first kernels (LiDAR projection, layout change) precede the first conv: back up to the previous gap
tot = collections.defaultdict(float); cnt = collections.Counter()
for r in rows[first:]:
    k = re.sub(r"^void\s+", "", r["Kernel_Name"]).replace("creste::", "")
    k = re.match(r"([\w:]+(?:<[^(]*>)?)", k).group(1)
    tot[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); cnt[k] += 1
T = sum(tot.values())
print(f"GPU kernel time per step (timed region only): {T / n / 1e6:.2f} ms\n\n| kernel | calls/step | ms/step | % |\n|---|---|---|---|")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:22]:
    print(f"| `{k}` | {cnt[k] / n:.1f} | {v / n / 1e6:.3f} | {100 * v / T:.1f} |")
