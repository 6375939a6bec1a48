#!/bin/bash
# build_variant.sh NAME "<extra flags>" : rebuild shade.hip with the flags, link against the other objects -> harp_amd/csrc/variants/libharp_NAME.so
set -e
cd "$(dirname "$0")/../.."
mkdir -p harp_amd/csrc/variants
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -DNDEBUG -I include"
SRC=${3:-shade}
EXTRA=""; [ "$SRC" = "shade_bwd" ] && EXTRA="-fno-slp-vectorize"
/opt/rocm/bin/hipcc -c $F $EXTRA $2 harp_amd/csrc/$SRC.hip -o harp_amd/csrc/variants/${SRC}_$1.o
OBJS=$(ls harp_amd/csrc/*.o | grep -v "/$SRC.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS harp_amd/csrc/variants/${SRC}_$1.o -o harp_amd/csrc/variants/libharp_$1.so
echo built $1
