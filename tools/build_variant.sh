#!/bin/bash
# tools/build_variant.sh <tag> [extra hipcc flags...]: a second build of libptmi355.so under
# project2-pathtracer_amd/build/variants/<tag>.so (git-ignored, travels to the GPU box); select it with PTMI355_LIB
# or "LIB=<tag> <bench args>" in tools/gpu_check.sh.  Same flags as the Makefile + the extras.
# VARIANT_SRC=<dir> builds from another copy of csrc/ (e.g. a git worktree of an older commit, for in-call A/Bs).
set -e
TAG=$1; shift
cd "$(dirname "$0")/../project2-pathtracer_amd"
SRC=${VARIANT_SRC:-csrc}
mkdir -p build/variants
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math \
      -fno-slp-vectorize -w -shared $SRC/pt_api.hip $SRC/pt_k_seg.hip $SRC/pt_k_queue.hip $SRC/pt_k_path.hip $SRC/pt_k_wide.hip \
      $SRC/pt_k_misc.hip $SRC/pt_scene.cpp $( [ -f $SRC/pt_build.cpp ] && echo $SRC/pt_build.cpp ) "$@" -o build/variants/$TAG.so
echo built build/variants/$TAG.so
