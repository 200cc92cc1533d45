#!/bin/bash
# Round 4's packed-fp32 corruption, revisited after the allocator-race fix (profiles/r05_pk_hazard.md): the kernel-level bisect
# (every encoder stage on serially computed inputs, two halves on two streams) with the SHIPPED library (no packed-fp32 VALU
# anywhere) and with a variant whose mbconv.hip alone is compiled with packed fp32 allowed
# (python -m creste_public_amd.build --variant pk mbconv.hip).
cd "$(dirname "$0")/.."
R=${1:-6}
echo "=== shipped library (no packed-fp32 VALU)"
python scripts/concurrency_bisect.py $R 2>&1 | grep -v amdgpu | grep -E "DIFFERS|wrong [1-9]|pairs|differs in|^EffNet|whole encoder" 
echo "=== variant: packed fp32 allowed in mbconv.hip only"
CRESTE_HIP_LIB=$PWD/creste_public_amd/lib/libcreste_hip_pk.so python scripts/concurrency_bisect.py $R 2>&1 | grep -v amdgpu | grep -E "DIFFERS|wrong [1-9]|pairs|differs in|^EffNet|whole encoder"
