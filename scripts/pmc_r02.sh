#!/bin/bash
# r02 PMC evidence (GPU box): (1) MFMA-busy / issue counters of the dominant conv kernel on its largest layer
# (496->496 3x3 at 152x304, batch 16, f16x3); (2) FETCH_SIZE / WRITE_SIZE of the BEV splat at batch 16 on the frustum
# distribution, separate passes.  Summaries -> gpurun_out/pmc_r02_{conv,splat}.txt (copied into profiles/ by hand).
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_r02; mkdir -p $OUT
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU" \
           "GRBM_GUI_ACTIVE FETCH_SIZE" "GRBM_GUI_ACTIVE WRITE_SIZE"; do
  tag=conv_$(echo $SET | cut -d' ' -f2)
  rocprofv3 --pmc $SET --output-format csv -d $OUT/$tag -o pmc -- python scripts/conv_micro.py f16x3 496 496 3 152 304 16 3 > $OUT/$tag.log 2>&1
done
for SET in "GRBM_GUI_ACTIVE FETCH_SIZE" "GRBM_GUI_ACTIVE WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_INSTS_VALU SQ_INSTS_LDS"; do
  tag=splat_$(echo $SET | cut -d' ' -f2)
  rocprofv3 --pmc $SET --output-format csv -d $OUT/$tag -o pmc -- python scripts/splat_micro.py frustum > $OUT/$tag.log 2>&1
done
python - <<PY
import csv, glob, collections
for what, pat in (("conv", "conv_patch3"), ("splat", "splat_")):
    tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
    for f in glob.glob("$OUT/%s_*/pmc_counter_collection.csv" % what):
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]:
                k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("creste::", "")[:60]
                tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
    with open("$OUT/../pmc_r02_%s.txt" % what, "w") as o:
        for k, d in sorted(tot.items()):
            o.write(k + "\n")
            for c, v in sorted(d.items()):
                o.write(f"   {c:28s} {v / cnt[k][c]:.6g} per launch ({cnt[k][c]} launches)\n")
    print(open("$OUT/../pmc_r02_%s.txt" % what).read())
PY
find $OUT -name "*.db" -delete
