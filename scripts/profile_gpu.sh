#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats + separate PMC passes of the bench command.
# Outputs land in gpurun_out/prof_<tag>/ ; scripts/summarize_profiles.py turns them into profiles/.
set -u
TAG=${1:-r01}
STEPS=${2:-3}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
PREC=${3:-bf16x6}
EXTRA=${4:-}      # e.g. "--parts 1": one forward on one stream (stand-alone kernel durations)
CMD="python bench.py --steps $STEPS --warmup 2 --no-cpu-baseline --no-modes --no-irl --precision $PREC $EXTRA"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
grep '^{' $OUT/trace.log > $OUT/bench_under_trace.json
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 1200 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -o pmc -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-modes --no-irl --no-host-fed --parts 1 --precision $PREC > $OUT/pmc_$C.log 2>&1
done
# (>= 3 steps in the PMC runs: summarize_profiles.py takes the dispatches BETWEEN the last two lidar_depth_kernel launches = one steady-state step)
rocminfo 2>/dev/null | grep -i "uuid" | grep -m1 "GPU-" | tr -s " " > $OUT/box.txt; rocminfo 2>/dev/null | grep -m1 "Marketing Name.*MI" | tr -s " " >> $OUT/box.txt
find $OUT -type f | head -50
# keep the merge small: drop the big per-dispatch traces except the stats/counter CSVs
find $OUT -name "*.db" -delete
ls -la $OUT/trace $OUT/pmc_FETCH_SIZE 2>/dev/null | head -40
