#!/bin/bash
# PMC profile of the weight-gradient micro-benchmark (GPU box). usage: pmc_wgrad.sh
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_wgrad; mkdir -p $OUT
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM" \
           "GRBM_GUI_ACTIVE FETCH_SIZE" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $SET | cut -d' ' -f2)
  rocprofv3 --pmc $SET --output-format csv -d $OUT/$tag -o pmc -- python scripts/wgrad_micro.py > $OUT/$tag.log 2>&1
done
python - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for f in glob.glob("$OUT/*/pmc_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "wgrad_f16" in r["Kernel_Name"]:
            k = r["Kernel_Name"][:40]
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k, d in tot.items():
    print(k)
    for c, v in sorted(d.items()): print(f"   {c:34s} {v / cnt[k][c]:.4g} per launch")
PY
