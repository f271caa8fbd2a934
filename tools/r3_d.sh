#!/bin/bash
# round-3 GPU call: phase clock of k_path_w (stats build) + counters of the configs[3] bench
set -u
OUT=gpurun_out; mkdir -p $OUT
PTMI355_LIB=$(pwd)/project2-pathtracer_amd/build/variants/stats.so timeout -k 10 300 python3 tools/wstats.py scenes/random256.txt > $OUT/r03d_wstats.log 2>&1; cat $OUT/r03d_wstats.log
bash tools/pmc_bench.sh r03d_a "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" --workload c4 2>&1 | tail -2
bash tools/pmc_bench.sh r03d_b "SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" --workload c4 2>&1 | tail -2
