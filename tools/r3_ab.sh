#!/bin/bash
# in-call A/B: k_path_q instantiated per queue capacity, the largest that keeps five blocks per CU (four with meshes) chosen at upload
# (working tree) against the fixed 138 records per wave (prev = the last commit)
set -u
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r03w_tests.log 2>&1 || { echo "TESTS FAILED"; tail -40 gpurun_out/r03w_tests.log; exit 1; }
tail -1 gpurun_out/r03w_tests.log
echo "== driver command"; bash tools/ab_lib.sh r03w_c3 3 "--steps 20 --warmup 5" prev - || exit 1
echo "== 192 steps"; bash tools/ab_lib.sh r03w_c3d 2 "" prev - || exit 1
echo "== configs[4]"; bash tools/ab_lib.sh r03w_c5 1 "--workload c5 --steps 20 --warmup 5" prev - || exit 1
echo "== configs[1]"; bash tools/ab_lib.sh r03w_c2 1 "--workload c2 --steps 20 --warmup 5" prev - || exit 1
echo "== mesh"; bash tools/ab_lib.sh r03w_mesh 2 "--workload mesh --steps 20 --warmup 5" prev - || exit 1
echo "== mesh5k"; bash tools/ab_lib.sh r03w_mesh5k 1 "--workload mesh5k --steps 20 --warmup 5" prev - || exit 1
echo "== direct light, whole paths"; bash tools/ab_lib.sh r03w_nee 1 "--steps 20 --warmup 5 --direct-light 1 --ordering 2" prev - || exit 1
