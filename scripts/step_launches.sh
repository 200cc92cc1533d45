#!/bin/bash
# usage: step_launches.sh TAG [extra bench args]
TAG=${1:-x}; shift
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/launches_$TAG; mkdir -p $OUT
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o trace -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-modes --no-irl --no-host-fed "$@" > $OUT/trace.log 2>&1
python scripts/step_launches.py $OUT/trace/trace_kernel_trace.csv > $OUT/launches.txt
python scripts/last_step_stats.py $OUT/trace/trace_kernel_trace.csv lidar_depth_kernel 45 > $OUT/step_table.md
find $OUT -name "*.db" -delete; rm -f $OUT/trace/trace_kernel_trace.csv
head -3 $OUT/step_table.md
