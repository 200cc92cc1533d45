#!/bin/bash
# A/B of the F(4x4) variants: CRESTE_W4_F32V = 0 pre-split bf16 pieces / 2 fp32 V, split at the head of each chunk / 1 fp32 V, streaming GEMM
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/w4order
for SHAPE in "496 496 152 304" "256 256 128 128" "256 128 256 256"; do
  for F in 0 2 1; do
    export CRESTE_W4_F32V=$F
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/w4order/t -o t -- python scripts/wino4_micro.py $SHAPE 16 10 2>&1 | grep "order="
    python scripts/top_kernels.py gpurun_out/w4order/t/t_kernel_stats.csv 4 | grep wino4
  done
done
rm -rf gpurun_out/w4order/t
