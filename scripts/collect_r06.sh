#!/bin/bash
# after `gpurun -- bash scripts/r06_final.sh`: copy what gpurun_out/ holds into the tracked evidence files under profiles/
# (run in the container, from the repo root)
set -e
TAG=${1:-r06}
{ echo "# One steady-state inference step of bench.py --parts 1 (batch 16, bf16x6, one forward on one stream; round-6 final build: phase-convolution heads, fused distillation head, deep-prefetch 1x1 kernel, sliced squeeze-excite gate, narrow GEMM tiles on small maps): rocprofv3 --kernel-trace, launches between two lidar_depth kernels (scripts/step_table.sh)"; echo; cat gpurun_out/step_$TAG/step_table.md; } > profiles/${TAG}_step_table.md
python scripts/summarize_profiles.py $TAG | tail -2
cp gpurun_out/${TAG}_pmc_encoder.txt gpurun_out/${TAG}_pmc_encoder.json profiles/
cp gpurun_out/pmc_calib/calib.json profiles/${TAG}_pmc_calibration.json
cp gpurun_out/${TAG}_bench_line.json profiles/${TAG}_bench_line.json
for f in gpurun_out/irl_$TAG/*.md; do cp $f profiles/${TAG}_irl_$(basename $f); done
for W in distill ssc; do
  cp gpurun_out/prof_${W}_step/trace/trace_kernel_stats.csv profiles/${TAG}_${W}_step_kernel_stats.csv
  n=$(grep -c "^it[0-9]" gpurun_out/prof_${W}_step/trace.log || true); n=$((n + 1))
done
python scripts/summarize_step.py profiles/${TAG}_distill_step_kernel_stats.csv 5 "stage-1 distillation train step, batch 8 of 1216x608, bf16x6, round-6 final build -- scripts/distill_step.py 8 bf16x6" > profiles/${TAG}_distill_step_summary.md
python scripts/summarize_step.py profiles/${TAG}_ssc_step_kernel_stats.csv 5 "BEV-SSC train step, batch 8 of 1216x608, bf16x6, round-6 final build -- scripts/ssc_step.py 8 bf16x6" > profiles/${TAG}_ssc_step_summary.md
[ -f gpurun_out/${TAG}_gputest_final.log ] && cp gpurun_out/${TAG}_gputest_final.log profiles/${TAG}_gputest_head.log
git status --short profiles | head -30
