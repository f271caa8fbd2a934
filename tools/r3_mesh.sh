#!/bin/bash
# MESH: global_load instead of flat_load for the blob reads (working tree) against the last commit (prevm)
set -u
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_mesh.py tests/test_gpu_round3.py -x -q -k "mesh" > gpurun_out/r03m8_tests.log 2>&1 || { echo "TESTS FAILED"; tail -40 gpurun_out/r03m8_tests.log; exit 1; }
tail -1 gpurun_out/r03m8_tests.log
echo "== mesh"; bash tools/ab_lib.sh r03m8_mesh 2 "--workload mesh --steps 20 --warmup 5" prevm - || exit 1
echo "== mesh5k"; bash tools/ab_lib.sh r03m8_mesh5k 2 "--workload mesh5k --steps 20 --warmup 5" prevm - || exit 1
