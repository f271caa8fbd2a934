#!/bin/bash
# in-call A/B of bench.py argument sets: tools/ab_args.sh <tag> <rounds> "<common args>" "label:extra args" ...  (alternating, one line per run)
TAG=$1; N=$2; COMMON=$3; shift 3
for i in $(seq 1 $N); do
  for spec in "$@"; do
    label=${spec%%:*}; extra=${spec#*:}
    timeout -k 10 300 python3 bench.py $COMMON $extra --no-cpu-baseline > gpurun_out/${TAG}_${label}_$i.log 2>&1 || { echo "run failed"; tail -5 gpurun_out/${TAG}_${label}_$i.log; exit 1; }
    python3 - gpurun_out/${TAG}_${label}_$i.log "$label" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d=json.loads(l); print(sys.argv[2], "ms/step", d["ms_per_step"], "min", round(min(d["passes_ms"])/d["steps"],4), "spread", d.get("spread"))
PY
  done
done
