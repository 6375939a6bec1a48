#!/bin/bash
# time one layer of conv_bench with every library variant under harp_amd/csrc/variants: tools/dev/conv_variants.sh "256,256,4" [N]
layer=${1:-256,256,4}; n=${2:-8}
for lib in harp_amd/csrc/libharp_hip.so harp_amd/csrc/variants/libharp_*.so; do
  echo "== $lib"
  HARP_LIB_PATH=$PWD/$lib CONV_LAYER=$layer python tools/dev/conv_bench.py $n 512 2>&1 | grep "^prec" | grep -v sum
done
