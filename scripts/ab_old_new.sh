#!/bin/bash
# A/B on one box: r04 python (no ordering of shared buffers) vs current
for i in 1 2; do
for v in _old .; do
  (cd $v && python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-irl --no-modes 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', l['value'], l['ms_per_step'], l.get('ms_per_step_one_stream'), l.get('host_fed',{}).get('ms_per_step'), l.get('host_fed',{}).get('equals_resident'))
")
done
done
