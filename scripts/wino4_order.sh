#!/bin/bash
# A/B of the F(4x4) transform variants: CRESTE_W4_ORDER (bit 0 input / bit 1 output transform workgroup order),
# CRESTE_W4_F32V (transformed input as fp32, split inside the GEMM)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/w4order
for SHAPE in "496 496 152 304" "472 472 76 152" "432 432 38 76" "256 256 128 128" "256 128 256 256"; do
  for F in 0 1; do
    export CRESTE_W4_F32V=$F
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/w4order/t -o t -- python scripts/wino4_micro.py $SHAPE 16 10 2>&1 | grep "order="
    python scripts/top_kernels.py gpurun_out/w4order/t/t_kernel_stats.csv 4 | grep wino4
  done
done
rm -rf gpurun_out/w4order/t
