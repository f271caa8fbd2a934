#!/bin/bash
# in-call A/B: k_path_w with five survivors' stacks (walk loop as committed) against the last commit (three)
set -u
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_round2.py tests/test_gpu_parity.py -x -q -k "wide or many_primitives or config4 or overflows or scene_scales or ray_records" > gpurun_out/r03x_tests.log 2>&1 || { echo "TESTS FAILED"; tail -40 gpurun_out/r03x_tests.log; exit 1; }
tail -1 gpurun_out/r03x_tests.log
echo "== configs[3]: last commit (3 stacks), working tree (5 stacks)"; bash tools/ab_lib.sh r03x_c4 3 "--workload c4 --steps 20 --warmup 5" prev - || exit 1
