#!/bin/bash
# kernel table of batch-1 inference (bf16x6): rocprofv3 kernel trace of scripts/latency_b1.py, per-step averages
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/lat_b1; mkdir -p $OUT
PREC=bf16x6 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python scripts/latency_b1.py 1 > $OUT/trace.log 2>&1
grep "B=1" $OUT/trace.log
python - <<'PY'
import csv, re, collections
rows = list(csv.DictReader(open("gpurun_out/lat_b1/trace/trace_kernel_stats.csv")))
steps = 76.0
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"GPU kernel time per step: {tot / steps / 1e6:.2f} ms")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:28]:
    n = re.sub(r"^void\s+", "", r["Name"]).replace("creste::", "").split("(")[0][:60]
    print(f"{n:62s} {float(r['Calls']) / steps:6.1f} calls {float(r['TotalDurationNs']) / steps / 1e6:7.3f} ms  avg {float(r['AverageNs']) / 1e3:7.1f} us")
PY
find $OUT -name "*.db" -delete; rm -f $OUT/trace/trace_kernel_trace.csv
