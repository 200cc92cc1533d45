#!/usr/bin/env python3
"""Turn gpurun_out/prof_<tag>/ (rocprofv3 kernel-trace stats + PMC passes of bench.py) into the tracked
files under profiles/: <tag>_kernel_stats.csv (verbatim rocprofv3 --stats summary),
<tag>_summary.md and roofline_counters.json (HBM traffic per launch, read by bench.py)."""
import csv
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)

shutil.copy(os.path.join(src, "trace", "trace_kernel_stats.csv"), os.path.join(dst, f"{tag}_kernel_stats.csv"))
bench = json.loads(open(os.path.join(src, "bench_under_trace.json")).readline())

stats = list(csv.DictReader(open(os.path.join(src, "trace", "trace_kernel_stats.csv"))))


def short(name):
    n = name.replace("void ", "").replace("creste::", "")
    return n.split("(")[0]


# PMC rows of ONE steady-state step only: the dispatches between the last two lidar_depth_kernel launches (a kernel name's
# launches elsewhere in the run -- the batch-2 BatchNorm calibration forward, warm-ups -- have other sizes and would skew a
# per-launch average: round 4's table showed the batch-16 gather at 325 MB for that reason)
pmc = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    per = defaultdict(lambda: [0.0, 0])
    f = os.path.join(src, f"pmc_{c}", "pmc_counter_collection.csv")
    if os.path.exists(f):
        disp = {}
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == c:
                d = disp.setdefault(int(row["Dispatch_Id"]), [row["Kernel_Name"], 0.0])
                d[1] += float(row["Counter_Value"])
        ids = sorted(disp)
        marks = [i for i in ids if "lidar_depth_kernel" in disp[i][0]]
        assert len(marks) >= 2, f"{c}: fewer than two steps in the PMC run -- no steady-state step to take (scripts/profile_gpu.sh)"
        lo, hi = marks[-2], marks[-1]
        for i in ids:
            if lo <= i < hi:
                k = short(disp[i][0])
                per[k][0] += disp[i][1]
                per[k][1] += 1
    pmc[c] = per

# rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB.  On gfx950 FETCH_SIZE counts 64 B per 128-B request of wide coalesced
# reads -> x2 (MI355X_MICROARCH.md section HBM); WRITE_SIZE is "uncalibrated" there: scripts/pmc_calib.sh measures both
# against known byte counts on the same box (1 GiB fill / copy, the gather's nontemporal stores over an empty plan) and the
# factors found replace the defaults.
ff, wf, wf_nt, calib = 2.0, 1.0, 1.0, None
cj = os.path.join(ROOT, "gpurun_out", "pmc_calib", "calib.json")
if os.path.exists(cj):
    calib = json.load(open(cj))
    f = calib.get("factors", {})
    if f.get("fetch_copy_1GiB"): ff = 1.0 / f["fetch_copy_1GiB"]
    if f.get("write_fill_1GiB"): wf = 1.0 / f["write_fill_1GiB"]
    if f.get("write_gather_nt_stores"): wf_nt = 1.0 / f["write_gather_nt_stores"]
counters = {}
for k in set(pmc["FETCH_SIZE"]) | set(pmc["WRITE_SIZE"]):
    fk, fn = pmc["FETCH_SIZE"].get(k, [0, 0])
    wk, wn = pmc["WRITE_SIZE"].get(k, [0, 0])
    n = max(fn, wn, 1)
    w = wf_nt if "splat_gather" in k else wf
    counters[k] = dict(launches=n, fetch_kib_raw=fk, write_kib_raw=wk,
                       hbm_bytes_per_launch=(ff * fk / max(fn, 1) + w * wk / max(wn, 1)) * 1024.0,
                       hbm_bytes_per_launch_uncorrected=(fk / max(fn, 1) + wk / max(wn, 1)) * 1024.0)
out = dict(counters)
out["_note"] = (f"per-launch averages over ONE step of bench.py --parts 1 (batch 16; `launches` = launches of the kernel in that step); hbm_bytes = ({ff:.3f} * FETCH_SIZE + "
                f"{wf:.3f} * WRITE_SIZE) * 1024 ({wf_nt:.3f} for the nontemporal stores of splat_gather8); factors "
                + ("calibrated on this box against known byte counts (scripts/pmc_calib.sh)" if calib else
                   "defaults: the gfx950 FETCH_SIZE x2 correction of the guide, WRITE_SIZE as reported"))
out["_source"] = f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py --parts 1, tag {tag}"
# the counters must be those of exactly the step scripts/step_table.sh tabulated: same kernels, same launch counts (VERDICT r05
# item 7: the r05 file carried the pack / calibration launches of a one-step run)
import re
import subprocess
st = os.path.join(ROOT, "gpurun_out", f"step_{tag}", "step_table.md")
if os.path.exists(st):
    table = {}
    for ln in open(st):
        m = re.match(r"\| `([^`]+)` \| (\d+) \|", ln)
        if m:
            table[m.group(1)] = int(m.group(2))
    fam = lambda k: re.match(r"([\w:]+(?:<[^(]*>)?)", k).group(1)[:70]
    got = {}
    for k, v in counters.items():
        got[fam(k)] = got.get(fam(k), 0) + v["launches"]
    # (the table lists the step's top 40 kernels; a kernel below its cut -- one tiny launch -- is only in the PMC pass)
    bad = {k: (table.get(k), got.get(k)) for k in table if table.get(k) != got.get(k) and not k.startswith(("at::", "__amd_rocclr"))}
    assert not bad, f"PMC step and step table disagree on launches (table, PMC): {bad}"
    out["_launch_check"] = f"launch counts of the {len(table)} kernel families of the step table equal gpurun_out/step_{tag}/step_table.md (profiles/{tag}_step_table.md)"
try:
    out["_commit"] = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
except Exception:
    out["_commit"] = None
bx = os.path.join(src, "box.txt")
out["_box"] = " / ".join(l.strip() for l in open(bx)) if os.path.exists(bx) else None
if calib:
    out["_calibration"] = calib
json.dump(out, open(os.path.join(dst, "roofline_counters.json"), "w"), indent=1, sort_keys=True)

tot = sum(float(r["TotalDurationNs"]) for r in stats)
lines = [f"# rocprofv3 --kernel-trace --stats summary ({tag})", "",
         f"command: `python bench.py --steps {bench['steps']} --warmup {bench['warmup']} --no-cpu-baseline` under rocprofv3 "
         f"(MI355X, 1 GPU); bench line under the profiler: {bench['value']} frames/s, {bench['ms_per_step']} ms/step, "
         f"dominant kernel {bench['roofline']['kernel']} at {bench['roofline']['achieved']} TFLOP/s "
         f"(HIP-event avg {bench['roofline']['avg_launch_ms']} ms/launch).", "",
         "| kernel | calls | total ms | avg us | % of GPU time | HBM MB/launch (PMC, corrected) |", "|---|---|---|---|---|---|"]
for r in stats[:24]:
    k = short(r["Name"])
    c = counters.get(k)
    hb = f"{c['hbm_bytes_per_launch'] / 1e6:.1f}" if c else ""
    lines.append(f"| `{k[:70]}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.2f} | "
                 f"{float(r['AverageNs']) / 1e3:.1f} | {100 * float(r['TotalDurationNs']) / tot:.1f} | {hb} |")
lines += ["", "Note: `bench.py` first gives the random-init network trained-network statistics with ONE training-mode pass at batch 2 "
          "(`synth.calibrate_bn_hip`) on the exact-fp32 engine; those launches are the extra `conv_igemm_f32_kernel` / "
          "`moments_kernel` / `bn_elementwise_kernel` rows above and are outside the timed steps. The f16x3 kernels (`conv_patch*`) "
          "only run inside the warm-up + timed steps, so their averages are per timed launch."]
open(os.path.join(dst, f"{tag}_summary.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:14]))
