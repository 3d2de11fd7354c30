#!/bin/bash
# Builds A/B variants of libbirda_hip.so that differ in the compile-time switches of ONE translation unit (default: the GELU copy of
# the fused MBConv kernel, the one the BirdNET-v2.4-shaped model runs): tools/ab/libbirda_hip_<tag>.so for every "tag:-DFLAG=.. .."
#   tools/build_variants.sh "base:-DBH_MB_BUFLOAD=0 -DBH_MB_HOSTRCP=0 -DBH_MB_CONSTDIV=0" "buf:-DBH_MB_BUFLOAD=1 -DBH_MB_HOSTRCP=0 -DBH_MB_CONSTDIV=0" ...
# UNIT=kernels_frontend builds variants of another unit.  The other objects are the current build's (make first).
cd "$(dirname "$0")/../birda_amd/csrc" || exit 1
unit=${UNIT:-kernels_mbconv_gelu}
objs=$(ls _build/*.o | grep -v "_build/$unit.o")
mkdir -p ../../tools/ab _build_var
pids=()
for spec in "$@"; do
  tag=${spec%%:*}; flags=${spec#*:}
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -Wall -Wno-unused-result -Wno-inline-asm $flags -c $unit.hip -o _build_var/${unit}_$tag.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ab/libbirda_hip_$tag.so $objs _build_var/${unit}_$tag.o -lpthread -ldl && echo "built $tag" ) &
  pids+=($!)
  while [ $(jobs -r | wc -l) -ge ${JOBS:-3} ]; do sleep 2; done
done
wait
