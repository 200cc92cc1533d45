#!/bin/bash
# A/B of the small-footprint F(4x4) transforms (CRESTE_W4_SMALL) on one box: stand-alone layers, the two-stream conv chain, the step
cd "$(dirname "$0")/.."
for small in 0 3 1 2; do
  export CRESTE_W4_SMALL=$small
  echo "== CRESTE_W4_SMALL=$small"
  python scripts/wino4_micro.py 496 496 152 304 2>&1 | grep -v amdgpu
  python scripts/wino4_micro.py 256 128 256 256 2>&1 | grep -v amdgpu
  python scripts/wino4_micro.py 256 256 128 128 2>&1 | grep -v amdgpu
done
for small in 0 3; do
  export CRESTE_W4_SMALL=$small
  echo "== overlap micro CRESTE_W4_SMALL=$small"
  python scripts/wino_overlap_micro.py 496 496 152 304 2>&1 | grep -v amdgpu | head -4
done
for i in 1 2; do
for small in 0 3; do
  export CRESTE_W4_SMALL=$small
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-irl --no-modes 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('bench small=$small', l['value'], l['ms_per_step'], l.get('ms_per_step_one_stream'), l.get('host_fed',{}).get('ms_per_step'), l.get('host_fed',{}).get('equals_resident'))
"
done
done
