#!/bin/bash
# in-call sweep: when a MESH turn runs (rays waiting) and when a thinned-out WALK gives way (lanes left)
set -u
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_mesh.py tests/test_gpu_round3.py -x -q -k "mesh" > gpurun_out/r03m3_tests.log 2>&1 || { echo "TESTS FAILED"; tail -40 gpurun_out/r03m3_tests.log; exit 1; }
tail -1 gpurun_out/r03m3_tests.log
echo "== mesh"; bash tools/ab_lib.sh r03m3_mesh 1 "--workload mesh --steps 20 --warmup 5" mt64_16 mt128_16 - mt128_32 mt128_40 mt192_32 || exit 1
echo "== mesh5k"; bash tools/ab_lib.sh r03m3_mesh5k 1 "--workload mesh5k --steps 20 --warmup 5" mt64_16 mt128_16 - mt128_32 mt128_40 mt192_32 || exit 1
for sc in scenes/cornell_mesh.txt scenes/cornell_mesh5k.txt; do PTMI355_LIB=$(pwd)/project2-pathtracer_amd/build/variants/meshstats.so timeout -k 10 200 python3 tools/meshstats.py $sc || exit 1; done
