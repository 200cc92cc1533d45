#!/bin/bash
# usage: train_launches.sh distill|ssc
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
W=${1:-distill}; OUT=gpurun_out/launches_${W}; mkdir -p $OUT
case $W in distill) CMD="python scripts/distill_step.py 8 bf16x6";; ssc) CMD="python scripts/ssc_step.py 8 bf16x6";; esac
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
python scripts/train_launches.py $OUT/trace/trace_kernel_trace.csv multi_tensor_apply 80 > $OUT/launches.txt 2>&1
find $OUT -name "*.db" -delete; rm -f $OUT/trace/trace_kernel_trace.csv
tail -4 $OUT/trace.log; head -3 $OUT/launches.txt
