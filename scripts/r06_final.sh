#!/bin/bash
# the round's final evidence batch on one box: default bench line, kernel stats / PMC / step table / MFMA-busy, IRL step tables,
# distillation / BEV-SSC step kernel stats
cd "$GRAFT_REPO_ROOT"
python bench.py > gpurun_out/r06_bench_line.json 2> gpurun_out/r06_bench_line.err
bash scripts/r06_evidence.sh r06 > gpurun_out/r06_evidence.log 2>&1
bash scripts/irl_tables.sh r06 > gpurun_out/r06_irl_tables.log 2>&1
for W in distill ssc; do bash scripts/step_profile.sh $W bf16x6 > gpurun_out/r06_${W}_profile.log 2>&1; done
tail -c 400 gpurun_out/r06_bench_line.json; ls gpurun_out/prof_distill_step/trace gpurun_out/prof_ssc_step/trace
