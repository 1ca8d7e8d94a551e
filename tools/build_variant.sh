#!/bin/bash
# Developer A/B helper: build_variant.sh NAME -DFLAG...  ->  build_variants/NAME.so (conv/pairwise/dense rebuilt with the flags).
# WITH_PLANES=1 also links the round-3 experiment tools/exp/conv_planes.hip (pre-split planes + LDS-DMA conv prototypes).
set -e
cd "$(dirname "$0")/../embeddingnet_amd/csrc"
name=$1; shift
out=../../build_variants; mkdir -p $out/obj_$name
rm -f $out/obj_$name/*.o
extra=""
if [ "${WITH_PLANES:-0}" = "1" ]; then extra="-DEMBNET_EXP_HOOKS=1"; fi
# any round-1/2 engine experiment switch selects the diagnostic engine header (tools/exp/gemm_engine_diag.h)
for a in "$@"; do
  case "$a" in -DEMBNET_ABLATE*|-DEMBNET_INTERLEAVE*|-DEMBNET_PIN*|-DEMBNET_SETPRIO*|-DEMBNET_SPLIT_DIST*|-DEMBNET_SPLIT_ABLATE*|-DEMBNET_LDS_STAGES*|-DEMBNET_PHASE_PRIO*|-DEMBNET_SPLIT_EARLY*|-DEMBNET_STAMPS*)
    extra="$extra -DEMBNET_DIAG_ENGINE=1";; esac
done
for f in conv pairwise dense; do
  /opt/rocm/bin/hipcc "$@" $extra -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -c $f.hip -o $out/obj_$name/$f.o &
done
if [ "${WITH_PLANES:-0}" = "1" ]; then
  /opt/rocm/bin/hipcc "$@" -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -c ../../tools/exp/conv_planes.hip -o $out/obj_$name/conv_planes.o &
fi
wait
others=$(ls *.o | grep -v -E "^(conv|pairwise|dense)\.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $out/$name.so $out/obj_$name/*.o $others
echo built $out/$name.so
