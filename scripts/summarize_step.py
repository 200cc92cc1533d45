"""Turn a rocprofv3 kernel_stats.csv of a training-step script into a per-step markdown table.
usage: summarize_step.py stats.csv n_steps "title" > summary.md"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1]))); n = float(sys.argv[2]); title = sys.argv[3]
def short(name):
    name = re.sub(r"^void\s+", "", name); name = name.replace("creste::", "")
    m = re.match(r"([\w:]+(?:<[^(]*>)?)", name)
    return (m.group(1) if m else name)[:70]
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"# rocprofv3 --kernel-trace --stats: {title}\n\nGPU kernel time per step: {tot / n / 1e6:.1f} ms ({int(n)} steps in the trace)\n")
print("| kernel | calls/step | ms/step | % | avg us |\n|---|---|---|---|---|")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:24]:
    t = float(r["TotalDurationNs"])
    print(f"| `{short(r['Name'])}` | {float(r['Calls']) / n:.1f} | {t / n / 1e6:.2f} | {100 * t / tot:.1f} | {float(r['AverageNs']) / 1e3:.1f} |")
