#!/bin/bash
# in-call A/B: whole-path contexts with launch groups of up to 96 M rays, evenly sized, and pools for one iteration only
# (working tree) against 32 M rays per group and greedy groups (prev = the last commit)
set -u
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r03q_tests.log 2>&1 || { echo "TESTS FAILED"; tail -40 gpurun_out/r03q_tests.log; exit 1; }
tail -1 gpurun_out/r03q_tests.log
echo "== configs[4]"; bash tools/ab_lib.sh r03q_c5 2 "--workload c5 --steps 20 --warmup 5" prev - || exit 1
echo "== configs[4], 64 steps"; bash tools/ab_lib.sh r03q_c5l 1 "--workload c5 --steps 64 --warmup 8" prev - || exit 1
echo "== driver command"; bash tools/ab_lib.sh r03q_c3 2 "--steps 20 --warmup 5" prev - || exit 1
echo "== 192 steps"; bash tools/ab_lib.sh r03q_c3d 2 "" prev - || exit 1
echo "== configs[3], 192 steps"; bash tools/ab_lib.sh r03q_c4d 1 "--workload c4" prev - || exit 1
echo "== mesh, 192 steps"; bash tools/ab_lib.sh r03q_meshd 1 "--workload mesh" prev - || exit 1
