#!/bin/bash
# the whole -m gpu suite + smoke on the tree as it is
set -u
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r03s_gpu_tests.log 2>&1 || { echo "GPU TESTS FAILED"; tail -60 gpurun_out/r03s_gpu_tests.log; exit 1; }
tail -2 gpurun_out/r03s_gpu_tests.log
timeout -k 10 300 python __graft_entry__.py smoke > gpurun_out/r03s_smoke.log 2>&1 || { echo "SMOKE FAILED"; tail -20 gpurun_out/r03s_smoke.log; exit 1; }
tail -1 gpurun_out/r03s_smoke.log
bash tools/ab_lib.sh r03s_nee 2 "--steps 20 --warmup 5 --direct-light 1 --ordering 2" - || exit 1
