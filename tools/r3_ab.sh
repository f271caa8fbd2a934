#!/bin/bash
# direct light on the whole-path kernel (k_path_q<NEE>): parity, then the bench line against the per-bounce kernels in one call
set -u
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_direct_light.py -x -q -m gpu > gpurun_out/r03r_nee_tests.log 2>&1 || { echo "NEE TESTS FAILED"; tail -60 gpurun_out/r03r_nee_tests.log; exit 1; }
tail -1 gpurun_out/r03r_nee_tests.log
timeout -k 10 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_round2.py tests/test_gpu_parity.py -x -q -k "wide or many_primitives or config4 or whole_path or queue_kernel or ray_pool or launch_variants" > gpurun_out/r03r_tests.log 2>&1 || { echo "TESTS FAILED"; tail -40 gpurun_out/r03r_tests.log; exit 1; }
tail -1 gpurun_out/r03r_tests.log
echo "== direct light, driver shape: per-bounce kernels (ordering 0), whole paths (ordering 2)"
bash tools/ab_lib.sh r03r_nee_o0 2 "--steps 20 --warmup 5 --direct-light 1 --ordering 0" - || exit 1
bash tools/ab_lib.sh r03r_nee_o2 2 "--steps 20 --warmup 5 --direct-light 1 --ordering 2" - || exit 1
echo "== configs[3] and the driver command on the tree as it is"
bash tools/ab_lib.sh r03r_c4 1 "--workload c4 --steps 20 --warmup 5" r3start - || exit 1
bash tools/ab_lib.sh r03r_c3 2 "--steps 20 --warmup 5" r3start - || exit 1
