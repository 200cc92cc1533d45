#!/bin/bash
# old kernel | the launcher's own choice | deep, 64-channel tiles | deep, 128-channel tiles
cd "$GRAFT_REPO_ROOT"
CRESTE_CONV1X1_DEEP=0 python scripts/conv1x1_deep_ab.py $1 2>&1 | grep -v amdgpu > gpurun_out/deep_ab_0.txt
python scripts/conv1x1_deep_ab.py $1 2>&1 | grep -v amdgpu > gpurun_out/deep_ab_1.txt
CRESTE_CONV1X1_DEEP=2 CRESTE_CONV1X1_DEEP_BN=64 python scripts/conv1x1_deep_ab.py $1 2>&1 | grep -v amdgpu > gpurun_out/deep_ab_2.txt
CRESTE_CONV1X1_DEEP=2 CRESTE_CONV1X1_DEEP_BN=128 python scripts/conv1x1_deep_ab.py $1 2>&1 | grep -v amdgpu > gpurun_out/deep_ab_3.txt
paste -d'|' gpurun_out/deep_ab_0.txt gpurun_out/deep_ab_1.txt gpurun_out/deep_ab_2.txt gpurun_out/deep_ab_3.txt | awk -F'|' '{split($1,a,": "); split($2,b,": "); split($3,c,": "); split($4,e,": "); split(a[2],a2," chk "); split(b[2],b2," chk "); split(c[2],c2," chk "); split(e[2],e2," chk "); ok = (a2[2]==b2[2] && a2[2]==c2[2] && a2[2]==e2[2]) ? "same bits" : "BITS DIFFER"; print a[1] ": " a2[1] " | " b2[1] " | " c2[1] " | " e2[1] " | " ok}' > gpurun_out/deep_ab.txt
cat gpurun_out/deep_ab.txt
