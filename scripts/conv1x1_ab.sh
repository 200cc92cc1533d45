echo "== new"; python scripts/conv1x1_ab.py 2>&1 | grep -v amdgpu
echo "== prev"; CRESTE_HIP_LIB=$PWD/creste_public_amd/lib/libcreste_hip_prev.so python scripts/conv1x1_ab.py 2>&1 | grep -v amdgpu
