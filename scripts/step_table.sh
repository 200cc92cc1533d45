#!/bin/bash
# kernel table of ONE steady-state inference step of bench.py (rocprofv3 kernel trace, scripts/last_step_stats.py)
# usage: step_table.sh TAG [marker-kernel]
TAG=${1:-r05}; MARK=${2:-lidar_depth_kernel}
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/step_$TAG; mkdir -p $OUT
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o trace -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-modes --no-irl --no-host-fed ${4:-} > $OUT/trace.log 2>&1
grep '^{' $OUT/trace.log | head -1 > $OUT/bench_under_trace.json
python scripts/last_step_stats.py $OUT/trace/trace_kernel_trace.csv $MARK 40 ${3:-} | tee $OUT/step_table.md
find $OUT -name "*.db" -delete; rm -f $OUT/trace/trace_kernel_trace.csv
