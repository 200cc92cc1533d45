#!/bin/bash
# FETCH_SIZE / WRITE_SIZE against known byte counts (scripts/pmc_calib.py) -> gpurun_out/pmc_calib/calib.json + a table
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_calib; mkdir -p $OUT
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --output-format csv -d $OUT/$C -o pmc -- python scripts/pmc_calib.py > $OUT/$C.log 2>&1
done
python - <<'PY'
import csv, glob, json, collections
out = "gpurun_out/pmc_calib"
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = []
    for f in glob.glob(f"{out}/{c}/**/pmc_counter_collection.csv", recursive=True):
        rows += [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == c]
    by = collections.OrderedDict()
    for r in rows:
        by[int(r["Dispatch_Id"])] = (r["Kernel_Name"], by.get(int(r["Dispatch_Id"]), ("", 0.0))[1] + float(r["Counter_Value"]))
    seq = [by[k] for k in sorted(by)]
    fill = [v for n, v in seq if "FillFunctor<float>" in n and v > 1e5][:4]
    copy = [v for n, v in seq if ("copy" in n.lower()) and v > 1e5][:4]
    gath = [v for n, v in seq if "splat_gather8" in n]
    res[c] = {"fill_KiB": sum(fill) / max(len(fill), 1), "copy_KiB": sum(copy) / max(len(copy), 1),
              "gather_empty_KiB": sum(gath[:4]) / 4 if len(gath) >= 8 else None, "gather_frustum_KiB": sum(gath[4:8]) / 4 if len(gath) >= 8 else None}
GiB = 1 << 20    # KiB in a GiB
bev_KiB = (16 * 256 * 256 * 97 * 4) / 1024.0
res["factors"] = {
    "fetch_copy_1GiB": res["FETCH_SIZE"]["copy_KiB"] / GiB, "write_fill_1GiB": res["WRITE_SIZE"]["fill_KiB"] / GiB,
    "write_copy_1GiB": res["WRITE_SIZE"]["copy_KiB"] / GiB,
    "write_gather_nt_stores": (res["WRITE_SIZE"]["gather_empty_KiB"] or 0) / bev_KiB,
}
json.dump(res, open(f"{out}/calib.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
find $OUT -name "*.db" -delete
