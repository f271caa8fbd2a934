#!/bin/bash
# counter survey: what bounds k_path_w when its vector instruction count falls and its time does not
set -u
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
(cd /tmp && rocprofv3 -L > $GRAFT_REPO_ROOT/$OUT/r03k_counters.txt 2>&1)
grep -c . $OUT/r03k_counters.txt
grep -o "SQC\?_[A-Z0-9_]*" $OUT/r03k_counters.txt | sort -u | tr '\n' ' ' | head -c 6000; echo
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_IFETCH SQ_BUSY_CYCLES"; do
  name=$(echo $set | cut -d' ' -f1)
  for v in r3start -; do
    if [ "$v" = "-" ]; then unset PTMI355_LIB; else export PTMI355_LIB=$(pwd)/project2-pathtracer_amd/build/variants/$v.so; fi
    bash tools/pmc_bench.sh r03k_${name}_$v "$set" --workload c4 2>&1 | tail -1
  done
done
