#!/bin/bash
# PMC totals over all k_bounce_* / k_fold dispatches of one bench run: tools/pmc_bench.sh <tag> "<counters>" [bench args]
TAG=$1; CTRS=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/pmcb_$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout -k 10 200 rocprofv3 --pmc $CTRS --output-format csv -d $OUT -- python3 $ROOT/bench.py --steps 20 --warmup 5 --warm-passes 1 --repeats 1 --no-cpu-baseline --no-kernel-events "$@" > $OUT/log.txt 2>&1
python3 - "$OUT" "$TAG" <<'PY'
import csv,glob,sys,collections
agg=collections.defaultdict(float); n=0
for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_bounce" in r["Kernel_Name"] or "k_path" in r["Kernel_Name"]: agg[r["Counter_Name"]]+=float(r["Counter_Value"])
print(sys.argv[2], {k: round(v/1e6,1) for k,v in sorted(agg.items())}, "(millions, all bounce dispatches of the run: 5 warm-up + 2 x 20 steps)")
PY
