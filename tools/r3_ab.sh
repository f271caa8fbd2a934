#!/bin/bash
# round-3 in-call A/B: configs[3] on the round-start library, the grid-walk commit and the working tree (+ parity of the many-primitive paths)
set -u
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_round3.py tests/test_gpu_round2.py tests/test_gpu_parity.py -x -q -k "wide or many_primitives or config4 or overflows or scene_scales" > gpurun_out/r03m_tests.log 2>&1 || { echo "TESTS FAILED"; tail -40 gpurun_out/r03m_tests.log; exit 1; }
tail -1 gpurun_out/r03m_tests.log
echo "== configs[3]"; bash tools/ab_lib.sh r03m_c4 2 "--workload c4 --steps 20 --warmup 5" r3start grid - || exit 1
PTMI355_LIB=$(pwd)/project2-pathtracer_amd/build/variants/stats.so timeout -k 10 300 python3 tools/wstats.py scenes/random256.txt > gpurun_out/r03m_wstats.log 2>&1; cat gpurun_out/r03m_wstats.log
bash tools/pmc_bench.sh r03m_a "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES" --workload c4 2>&1 | tail -1
