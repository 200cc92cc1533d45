#!/bin/bash
# the short bench N times on one box: timed (pipelined) vs one-stream vs host-fed ms per step
cd "$(dirname "$0")/.."
for i in $(seq 1 ${1:-4}); do
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-irl --no-modes 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('bench', l['value'], l['ms_per_step'], l.get('ms_per_step_one_stream'), l.get('host_fed',{}).get('ms_per_step'), l.get('host_fed',{}).get('equals_resident'))
"
done
