#!/bin/bash
# in-call A/B: k_path_w payload records as one 64-byte line per ray (four 16-byte accesses) instead of fifteen field-major arrays
set -u
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_round2.py tests/test_gpu_parity.py -x -q -k "wide or many_primitives or config4 or overflows or scene_scales or ray_records" > gpurun_out/r03y_tests.log 2>&1 || { echo "TESTS FAILED"; tail -40 gpurun_out/r03y_tests.log; exit 1; }
tail -1 gpurun_out/r03y_tests.log
echo "== configs[3]: last commit, working tree"; bash tools/ab_lib.sh r03y_c4 3 "--workload c4 --steps 20 --warmup 5" prev - || exit 1
bash tools/pmc_bench.sh r03y_f "FETCH_SIZE" --workload c4 2>&1 | tail -1
bash tools/pmc_bench.sh r03y_w "WRITE_SIZE" --workload c4 2>&1 | tail -1
