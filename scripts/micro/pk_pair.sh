#!/bin/bash
# The two-kernel reproducer of the packed-fp32 corruption with one-file variant libraries built ON THE BOX (they are not part of the
# shipped tree): mbconv.hip with packed fp32 allowed, and the same with one stage of the fused kernel forced onto scalar v_fma_f32
# (CRESTE_MB_BISECT: 1 expand stage = the v_pk_fma_f32 with a broadcast operand, 2 depthwise stage, 3 both; 4 = the expand stage packed
# but with the broadcast pair in registers instead of op_sel)
cd "$GRAFT_REPO_ROOT"
R=${1:-40}
python scripts/micro/pk_pair.py $R 2>&1 | grep -v amdgpu
for B in ${2:-0 1 2 3 4}; do
  python -c "from creste_public_amd import build; build.build_variant('pk$B', ('mbconv.hip',), verbose=False, extra_flags=('-DCRESTE_MB_BISECT=$B',))" > /dev/null 2>&1
  echo "--- packed fp32 allowed in mbconv.hip, CRESTE_MB_BISECT=$B"
  CRESTE_HIP_LIB=$PWD/creste_public_amd/lib/libcreste_hip_pk$B.so python scripts/micro/pk_pair.py $R 2>&1 | grep -v amdgpu | head -4
  rm -f creste_public_amd/lib/libcreste_hip_pk$B.so; rm -rf creste_public_amd/lib/obj_pk$B
done
