#!/bin/bash
# MESH: children visited near side first (skip links per direction octant; working tree) against the fixed depth-first order (prevm = the last commit)
set -u
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_mesh.py tests/test_gpu_round3.py -x -q -k "mesh" > gpurun_out/r03m9_tests.log 2>&1 || { echo "TESTS FAILED"; tail -40 gpurun_out/r03m9_tests.log; exit 1; }
tail -1 gpurun_out/r03m9_tests.log
echo "== mesh"; bash tools/ab_lib.sh r03m9_mesh 2 "--workload mesh --steps 20 --warmup 5" prevm - || exit 1
echo "== mesh5k"; bash tools/ab_lib.sh r03m9_mesh5k 2 "--workload mesh5k --steps 20 --warmup 5" prevm - || exit 1
echo "== per-bounce kernels (per-lane mesh_test)"; bash tools/ab_lib.sh r03m9_mesh_o0 1 "--workload mesh --steps 20 --warmup 5 --ordering 0" prevm - || exit 1
for sc in scenes/cornell_mesh.txt scenes/cornell_mesh5k.txt; do PTMI355_LIB=$(pwd)/project2-pathtracer_amd/build/variants/meshstats.so timeout -k 10 200 python3 tools/meshstats.py $sc > gpurun_out/r03m9_meshstats_$(basename $sc .txt).log || exit 1; cat gpurun_out/r03m9_meshstats_$(basename $sc .txt).log; done
