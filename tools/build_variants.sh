#!/bin/bash
# Builds library variants for A/B runs: tools/build_variants.sh name1 "flags1" name2 "flags2" ... -> variants/<name>.so
# (at most four at a time: each build takes one core for about two minutes)
cd "$(dirname "$0")/../mitsuba_amd/csrc"
mkdir -p ../../variants
SRCS="mtsamd.hip host_scene.cpp host_util.cpp film_ldr.cpp spectrum.cpp mipmap.cpp image_io.cpp bvh.cpp bvh_device.hip"
FL="-std=c++17 -O3 -fPIC -ffp-contract=off -Wall -Wno-unused-function -Wno-unused-result --offload-arch=gfx950 -shared -x hip"
while [ $# -ge 2 ]; do
  n=$1; f=$2; shift 2
  ( /opt/rocm/bin/hipcc $FL $f $SRCS -o ../../variants/$n.so -lpthread -lz 2>&1 | grep -E "error|warning: v" | head -5; echo "built $n" ) &
done
wait
