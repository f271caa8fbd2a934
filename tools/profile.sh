#!/bin/bash
# Profile bench.py on the GPU box with rocprofv3: one kernel-trace pass (+stats) and separate PMC
# passes (never combined with other trace domains).  Usage: tools/profile.sh <tag> [bench args...]
# Writes CSVs under gpurun_out/prof_<tag>/ ; summarise with tools/summarize_profile.py.
set -u
TAG=${1:-r01}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
# 32 warm-up steps + 2 passes (one untimed, one timed: --repeats 1) of 64 steps = 160 rendered steps in FULL launch
# groups (2 contexts x 32 iterations at 1080p), so the per-kernel averages are comparable with the default bench run
BENCH="python3 $ROOT/bench.py --steps 64 --warmup 32 --warm-passes 1 --repeats 1 --no-cpu-baseline --no-kernel-events $*"
cd /tmp
echo "== kernel trace"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- $BENCH > "$OUT/kt.log" 2>&1 || { echo "kernel trace failed"; tail -5 "$OUT/kt.log"; exit 1; }
[ -n "${KT_ONLY:-}" ] && { echo "kernel trace only"; exit 0; }
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $set | cut -d' ' -f1)
  echo "== pmc $set"
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d "$OUT/pmc_$name" -- $BENCH > "$OUT/pmc_$name.log" 2>&1 || { echo "pmc pass $name failed"; tail -5 "$OUT/pmc_$name.log"; }
done
find "$OUT" -name "*.csv" | head -40
du -sh "$OUT"
