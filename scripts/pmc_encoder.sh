#!/bin/bash
# MFMA-busy of the F(4x4) GEMM kernel and of the whole RGB-D encoder of one inference step (north_star: ">= 40 % MFMA on the
# encoder"): two rocprofv3 --pmc passes (no tracing) of `bench.py --parts 1` -> profiles-ready summary in gpurun_out/<tag>_pmc_encoder.txt
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_encoder; mkdir -p $OUT
CMD="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-irl --no-modes --no-host-fed --parts 1"
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $OUT/sq -o pmc -- $CMD > $OUT/sq.log 2>&1
timeout 900 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/grbm -o pmc -- $CMD > $OUT/grbm.log 2>&1
python scripts/pmc_encoder_summary.py $OUT gpurun_out/${TAG}_pmc_encoder.json > gpurun_out/${TAG}_pmc_encoder.txt
cat gpurun_out/${TAG}_pmc_encoder.txt
find $OUT -name "*.db" -delete
