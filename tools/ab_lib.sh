#!/bin/bash
# in-call A/B of library builds: tools/ab_lib.sh <tag> <rounds> "<bench args>" <variant|-> ...   ("-" = the shipped library)
TAG=$1; N=$2; ARGS=$3; shift 3
for i in $(seq 1 $N); do
  for v in "$@"; do
    if [ "$v" = "-" ]; then unset PTMI355_LIB; else export PTMI355_LIB=$(pwd)/project2-pathtracer_amd/build/variants/$v.so; fi
    timeout -k 10 300 python3 bench.py $ARGS --no-cpu-baseline > gpurun_out/${TAG}_${v}_$i.log 2>&1 || { echo "run failed"; tail -5 gpurun_out/${TAG}_${v}_$i.log; exit 1; }
    python3 - gpurun_out/${TAG}_${v}_$i.log "$v" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d=json.loads(l); print(sys.argv[2], "ms/step", d["ms_per_step"], "min", round(min(d["passes_ms"])/d["steps"],4), "spread", d.get("spread"))
PY
  done
done
