#!/bin/bash
# Developer A/B helper: build_variant.sh NAME -DFLAG...  ->  build_variants/NAME.so (conv/pairwise/dense rebuilt with the flags).
set -e
cd "$(dirname "$0")/../embeddingnet_amd/csrc"
name=$1; shift
out=../../build_variants; mkdir -p $out/obj_$name
for f in conv conv_planes pairwise dense; do
  /opt/rocm/bin/hipcc "$@" -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -c $f.hip -o $out/obj_$name/$f.o &
done
wait
others=$(ls *.o | grep -v -E "^(conv|conv_planes|pairwise|dense)\.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $out/$name.so $out/obj_$name/*.o $others
echo built $out/$name.so
