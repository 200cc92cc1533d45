#!/bin/bash
# PMC profile of the splat gather kernel (GPU box)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_splat; mkdir -p $OUT
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_VMEM SQ_INSTS_VALU" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS" \
           "GRBM_GUI_ACTIVE FETCH_SIZE" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  tag=$(echo $SET | cut -d' ' -f2)
  rocprofv3 --pmc $SET --output-format csv -d $OUT/$tag -o pmc -- python scripts/splat_micro.py ${1:-frustum} > $OUT/$tag.log 2>&1
done
python - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for f in glob.glob("$OUT/*/pmc_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "splat_gather" in r["Kernel_Name"]:
            k = r["Kernel_Name"][:40]
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k, d in tot.items():
    print(k)
    for c, v in sorted(d.items()): print(f"   {c:34s} {v / cnt[k][c]:.4g} per launch")
PY
