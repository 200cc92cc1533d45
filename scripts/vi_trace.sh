#!/bin/bash
# build creste_public_amd/lib/libcreste_trace.so = the library with csrc/value_iteration.hip compiled -DVI_TRACE (here, no GPU needed)
cd "$(dirname "$0")/.."
mkdir -p /tmp/vit
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -DVI_TRACE -x hip -c creste_public_amd/csrc/value_iteration.hip -o /tmp/vit/value_iteration.o || exit 1
OBJS=$(ls creste_public_amd/lib/obj/*.o | grep -v value_iteration)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/vit/value_iteration.o -o creste_public_amd/lib/libcreste_trace.so
