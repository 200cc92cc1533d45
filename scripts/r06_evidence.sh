#!/bin/bash
# round-6 evidence batch (GPU box): PMC calibration, kernel stats + FETCH/WRITE passes, one-step table, MFMA-busy of the encoder
TAG=${1:-r06}
cd "$GRAFT_REPO_ROOT"
bash scripts/pmc_calib.sh > gpurun_out/${TAG}_pmc_calib.log 2>&1
bash scripts/profile_gpu.sh $TAG 5 bf16x6 "--parts 1" > gpurun_out/${TAG}_profile.log 2>&1
bash scripts/step_table.sh $TAG lidar_depth_kernel "" "--parts 1" > gpurun_out/${TAG}_step_table.log 2>&1
bash scripts/pmc_encoder.sh $TAG > gpurun_out/${TAG}_pmc_encoder.log 2>&1
tail -3 gpurun_out/${TAG}_pmc_calib.log; tail -3 gpurun_out/${TAG}_step_table.log; head -8 gpurun_out/${TAG}_pmc_encoder.txt
