#!/bin/bash
# distillation / BEV-SSC train step with and without the BatchNorm statistics from the conv epilogues (CRESTE_CONV_STATS)
for i in 1 2; do for f in 1 0; do
  echo "== CRESTE_CONV_STATS=$f"
  CRESTE_CONV_STATS=$f python scripts/distill_step.py 8 bf16x6 2>&1 | grep -v amdgpu | tail -2
  CRESTE_CONV_STATS=$f python scripts/ssc_step.py 8 bf16x6 2>&1 | grep -v amdgpu | tail -2
done; done
