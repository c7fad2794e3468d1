#!/bin/bash
# Builds variants/librt_NAME.so: the library with kernels.hip and api.cpp compiled with extra flags (tuning macros).
# usage: tools/build_variant.sh NAME [-DMACRO=VALUE ...]
set -e
name=$1; shift
cd "$(dirname "$0")/../raytracer_amd/csrc"
mkdir -p ../../variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wall -Wno-unused-function"
/opt/rocm/bin/hipcc $FLAGS "$@" -c kernels.hip -o ../../variants/kernels_$name.o
/opt/rocm/bin/hipcc $FLAGS "$@" -c api.cpp -o ../../variants/api_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../variants/librt_$name.so ../../variants/kernels_$name.o bvh_build.o ../../variants/api_$name.o scene_build.o gltf.o png_decode.o jpeg_decode.o -lpthread -lz -ldl
rm -f ../../variants/kernels_$name.o ../../variants/api_$name.o
echo built variants/librt_$name.so
