#!/bin/bash
# round 4: counter survey of k_path_w on configs[3] (what do the waves wait for?)
set -u
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
(cd /tmp && timeout -k 10 120 rocprofv3 --list-avail > $GRAFT_REPO_ROOT/$OUT/r04e_counters.txt 2>&1; true)
grep -c . $OUT/r04e_counters.txt
bash tools/pmc_bench.sh r04e_a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" --workload c4 || exit 1
bash tools/pmc_bench.sh r04e_b "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM" --workload c4 || exit 1
bash tools/pmc_bench.sh r04e_c "SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_INSTS_BRANCH" --workload c4 || exit 1
