#!/bin/bash
# what the F(4x4) GEMM kernel's time is made of: timing experiments with parts of the loop removed (results are WRONG under
# CRESTE_W4_EXP != 0; needs a build with CRESTE_W4_EXPERIMENTS=1).  GEMM only (CRESTE_W4_ONLY=2 keeps V / M of the first call).
cd "$(dirname "$0")/.."
for ex in ${EXPS:-0 1 2 3 4 5}; do
  echo "== CRESTE_W4_EXP=$ex"
  CRESTE_W4_EXP=$ex ONLY=2 python scripts/coresidency_probe.py 2>&1 | grep "GEMM alone"
  CRESTE_W4_EXP=$ex ONLY=2 CH=256 python scripts/coresidency_probe.py 2>&1 | grep "GEMM alone"
done
