#!/bin/bash
# kernel-trace stats of one training-step script (GPU box). usage: step_profile.sh distill|ssc|irl
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
W=${1:-distill}; OUT=gpurun_out/prof_${W}_step; mkdir -p $OUT
case $W in distill) CMD="python scripts/distill_step.py 8 ${2:-bf16x6}";; ssc) CMD="python scripts/ssc_step.py 8 ${2:-bf16x6}";; irl) CMD="python scripts/irl_profile.py";; esac
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete
tail -3 $OUT/trace.log; find $OUT -name "*kernel_stats.csv"
