#!/bin/bash
# one extra PMC pass over bench.py: tools/pmc_pass.sh <tag> "<counters>" [bench args]
TAG=$1; CTRS=$2; shift; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/pmc_$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout -k 10 200 rocprofv3 --pmc $CTRS --output-format csv -d $OUT -- python3 $ROOT/bench.py --steps 32 --warmup 0 --no-cpu-baseline --no-kernel-events "$@" > $OUT/log.txt 2>&1
python3 - "$OUT" "$TAG" <<'PY'
import csv,glob,sys,collections
agg=collections.defaultdict(float)
for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_bounce" in r["Kernel_Name"]: agg[r["Counter_Name"]]+=float(r["Counter_Value"])
print(sys.argv[2], {k: round(v/32/1e6,3) for k,v in sorted(agg.items())}, "(millions per step)")
PY
