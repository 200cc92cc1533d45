#!/bin/bash
# kernel timeline of the pipelined inference steps: rocprofv3 --kernel-trace of bench.py (few steps, no extras)
cd "$(dirname "$0")/.."
R=$PWD
cd /tmp && export TMPDIR=/tmp
for small in ${SMALLS:-3 0}; do
  CRESTE_W4_SMALL=$small rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_step_s$small -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-irl --no-modes --no-host-fed 2>/dev/null | python -c "
import json,sys
for ln in sys.stdin:
    if ln.startswith('{'):
        l=json.loads(ln); print('bench small=$small', l['value'], l['ms_per_step'], l.get('ms_per_step_one_stream'))
"
done
