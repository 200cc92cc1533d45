#!/bin/bash
# rocprofv3 kernel tables of ONE steady-state IRL training step per variant, with and without the prefetched frozen half
# usage: irl_tables.sh TAG   -> gpurun_out/irl_TAG/{variant}_{pf0,pf1}.md
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/irl_$TAG; mkdir -p $OUT
for V in reference mdp256 cf512; do for PF in 0 1; do
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -o trace -- python scripts/irl_step.py $V $PF 6 > $OUT/${V}_pf$PF.log 2>&1
  { grep "ms / step" $OUT/${V}_pf$PF.log; python scripts/last_step_stats.py $OUT/tr/trace_kernel_trace.csv svf_kernel 28; } > $OUT/${V}_pf$PF.md
  rm -rf $OUT/tr
done; done
head -4 $OUT/*.md
