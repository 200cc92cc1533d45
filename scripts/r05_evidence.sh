#!/bin/bash
# round-5 evidence batch (GPU box): PMC calibration, kernel stats + FETCH/WRITE passes, one-step table, MFMA-busy of the encoder
cd "$GRAFT_REPO_ROOT"
bash scripts/pmc_calib.sh > gpurun_out/r05_pmc_calib.log 2>&1
bash scripts/profile_gpu.sh r05 5 bf16x6 "--parts 1" > gpurun_out/r05_profile.log 2>&1
bash scripts/step_table.sh r05 lidar_depth_kernel "" "--parts 1" > gpurun_out/r05_step_table.log 2>&1
bash scripts/pmc_encoder.sh > gpurun_out/r05_pmc_encoder.log 2>&1
tail -3 gpurun_out/r05_pmc_calib.log; tail -3 gpurun_out/r05_step_table.log; head -8 gpurun_out/r05_pmc_encoder.txt
