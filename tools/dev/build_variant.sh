#!/bin/bash
# build_variant.sh NAME "<extra flags>" [SRC] [PATCH]: rebuild SRC.hip (default shade) with the flags, link against the other objects
#   -> harp_amd/csrc/variants/libharp_NAME.so (select it with HARP_LIB_PATH).
# PATCH (tools/dev/variants/*.patch) is applied to a scratch copy of harp_amd/csrc first: the timing-only ablation switches that produce
# WRONG results (CONV_NOSTAGE / CONV_NOFETCH / CONV_LINEAR_LDS, RASTER_ABLATE=<bits>) live in those patches, not in the product sources.
set -e
cd "$(dirname "$0")/../.."
mkdir -p harp_amd/csrc/variants
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -DNDEBUG -I include"
SRC=${3:-shade}
PATCH=$4
EXTRA=""; [ "$SRC" = "shade_bwd" ] && EXTRA="-fno-slp-vectorize -mllvm --amdgpu-sched-strategy=max-memory-clause"
[ "$SRC" = "raster" ] && EXTRA="-fno-slp-vectorize"
[ "$SRC" = "conv" ] && EXTRA="-mllvm --amdgpu-sched-strategy=max-memory-clause"
DIR=harp_amd/csrc
if [ -n "$PATCH" ]; then
  DIR=$(mktemp -d)/harp_amd/csrc; mkdir -p $DIR; cp harp_amd/csrc/*.hip harp_amd/csrc/*.h $DIR/
  (cd $DIR/../.. && patch -p0 < "$OLDPWD/$PATCH")
fi
/opt/rocm/bin/hipcc -c $F $EXTRA $2 $DIR/$SRC.hip -o harp_amd/csrc/variants/${SRC}_$1.o
OBJS=$(ls harp_amd/csrc/*.o | grep -v "/$SRC.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS harp_amd/csrc/variants/${SRC}_$1.o -o harp_amd/csrc/variants/libharp_$1.so
echo built $1
