#!/bin/bash
# VALU wave-instructions per step for a bench configuration: tools/valu_count.sh <tag> [bench args]
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/valu_$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY --output-format csv -d $OUT -- python3 $ROOT/bench.py --steps 32 --warmup 0 --no-cpu-baseline --no-kernel-events "$@" > $OUT/log.txt 2>&1
python3 - "$OUT" "$TAG" <<'PY'
import csv,glob,sys,collections
agg=collections.defaultdict(float)
for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_bounce" in r["Kernel_Name"]: agg[r["Counter_Name"]]+=float(r["Counter_Value"])
steps=32.0
print(sys.argv[2], {k: round(v/steps/1e6,2) for k,v in sorted(agg.items())}, "(millions per step)")
PY
