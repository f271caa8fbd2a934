#!/bin/bash
# tools/build_variant.sh <tag> [extra hipcc flags...]: a second build of libptmi355.so under
# project2-pathtracer_amd/build/variants/<tag>.so (git-ignored, travels to the GPU box); select it with PTMI355_LIB
# or "LIB=<tag> <bench args>" in tools/gpu_check.sh.  Same flags as the Makefile + the extras.
set -e
TAG=$1; shift
cd "$(dirname "$0")/../project2-pathtracer_amd"
mkdir -p build/variants
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math \
      -fno-slp-vectorize -w -shared csrc/pt_kernels.hip csrc/pt_scene.cpp "$@" -o build/variants/$TAG.so
echo built build/variants/$TAG.so
