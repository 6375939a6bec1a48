#!/bin/bash
# kernel_regs.sh FILE.hip [extra flags]: VGPRs / SGPRs / scratch / LDS / occupancy of every kernel in the file (compiler remarks)
cd "$(dirname "$0")/../.."
f=$1; shift
EXTRA=""; [ "$(basename $f)" = "shade_bwd.hip" ] && EXTRA="-fno-slp-vectorize -mllvm --amdgpu-sched-strategy=max-memory-clause"
[ "$(basename $f)" = "raster.hip" ] && EXTRA="-fno-slp-vectorize"
[ "$(basename $f)" = "conv.hip" ] && EXTRA="-mllvm --amdgpu-sched-strategy=max-memory-clause"
/opt/rocm/bin/hipcc -c --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -DNDEBUG -I include $EXTRA "$@" \
  -Rpass-analysis=kernel-resource-usage $f -o /tmp/kr_$$.o 2>&1 | grep -E "Function Name|VGPRs:|AGPRs|ScratchSize|Occupancy|LDS Size|SGPRs:" | \
  sed -e 's/.*remark: [^ ]* //' | paste - - - - - - - | sed -e 's/\[-Rpass-analysis=kernel-resource-usage\]//g'
rm -f /tmp/kr_$$.o
