#!/bin/bash
# PMC profile of the conv micro-benchmark (GPU box). usage: pmc_conv.sh PREC
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
P=${1:-bf16x3}
OUT=gpurun_out/pmc_conv_$P; mkdir -p $OUT
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT" \
           "GRBM_GUI_ACTIVE FETCH_SIZE" "GRBM_GUI_ACTIVE WRITE_SIZE"; do
  tag=$(echo $SET | cut -d' ' -f1)
  rocprofv3 --pmc $SET --output-format csv -d $OUT/$tag -o pmc -- python scripts/conv_micro.py $P 496 496 3 152 304 16 3 > $OUT/$tag.log 2>&1
done
python - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("$OUT/*/pmc_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "conv_patch" in r["Kernel_Name"] or "conv_igemm" in r["Kernel_Name"]:
            tot[r["Kernel_Name"][:60]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in tot.items():
    print(k)
    for c, v in sorted(d.items()): print(f"   {c:34s} {v:.4g}")
PY
