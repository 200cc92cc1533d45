#!/bin/bash
# A/B of the row-split wave layout of the F(4x4) GEMM (CRESTE_W4_RS) on one box
cd "$(dirname "$0")/.."
for rs in 0 1 0 1; do
  export CRESTE_W4_RS=$rs
  echo "== CRESTE_W4_RS=$rs"
  ONLY=2 python scripts/coresidency_probe.py 2>&1 | grep "GEMM alone"
  ONLY=2 CH=256 python scripts/coresidency_probe.py 2>&1 | grep "GEMM alone"
  python scripts/wino4_micro.py 496 496 152 304 2>&1 | grep -v amdgpu
  python scripts/wino4_micro.py 256 128 256 256 2>&1 | grep -v amdgpu
done
for i in 1 2; do
for rs in 0 1; do
  export CRESTE_W4_RS=$rs
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-irl --no-modes 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('bench rs=$rs', l['value'], l['ms_per_step'], l.get('ms_per_step_one_stream'), l.get('host_fed',{}).get('ms_per_step'), l.get('host_fed',{}).get('equals_resident'))
"
done
done
