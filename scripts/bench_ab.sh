#!/bin/bash
# same-box A/B of the headline step under env settings given as arguments ("VAR=val VAR2=val" per arm)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for ARM in "$@"; do
  echo "== arm: $ARM"
  env $ARM python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-modes --no-irl --no-host-fed 2>&1 | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('  ms_per_step', d['ms_per_step'], 'frames/s', d['value'], 'roofline', d['roofline']['frac'], 'avg_launch_ms', d['roofline']['avg_launch_ms'], 'splat', d.get('roofline_splat', {}).get('frac'))"
done
