#!/bin/bash
# PMC passes (each counter set in its own rocprofv3 run, no tracing) of ONE command, summarised per kernel whose name
# contains $1:   scripts/pmc_kernel.sh <kernel-substring> <out-tag> -- <command...>
# -> gpurun_out/pmc_<tag>.txt  (per-launch averages; SQ counters are summed over SIMDs, TCC / GRBM over XCDs)
PAT=$1; TAG=$2; shift 3
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$TAG; mkdir -p $OUT
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU" \
           "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "GRBM_GUI_ACTIVE FETCH_SIZE" "GRBM_GUI_ACTIVE WRITE_SIZE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $SET --output-format csv -d $OUT/s$i -o pmc -- "$@" > $OUT/s$i.log 2>&1
done
python - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for f in glob.glob("$OUT/s*/pmc_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "$PAT" in r["Kernel_Name"]:
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("creste::", "")[:70]
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
with open("gpurun_out/pmc_$TAG.txt", "w") as o:
    for k, d in sorted(tot.items()):
        o.write(k + "\n")
        for c, v in sorted(d.items()):
            o.write(f"   {c:32s} {v / cnt[k][c]:.6g} per launch ({cnt[k][c]} launches)\n")
print(open("gpurun_out/pmc_$TAG.txt").read())
PY
find $OUT -name "*.db" -delete
