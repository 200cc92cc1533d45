#!/bin/bash
# kernel timeline of the two-stream conv chain (who overlaps whom): rocprofv3 --kernel-trace of wino_overlap_micro.py
cd "$(dirname "$0")/.."
cd /tmp && export TMPDIR=/tmp
R=$OLDPWD
for small in 3 0; do
  CRESTE_W4_SMALL=$small OVERLAP_QUICK=1 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_overlap_s$small -- python $R/scripts/wino_overlap_micro.py 496 496 152 304 16 4 2>&1 | grep -v amdgpu | tail -4
done
