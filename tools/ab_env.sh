#!/bin/bash
# in-call A/B of an environment switch: tools/ab_env.sh <tag> <VAR> "<bench args>" [rounds]  -> alternating runs without / with VAR=1
TAG=$1; VAR=$2; ARGS=$3; N=${4:-3}
for i in $(seq 1 $N); do
  for v in off on; do
    if [ $v = on ]; then export $VAR=${ON:-1}; else if [ -n "${OFF:-}" ]; then export $VAR=$OFF; else unset $VAR; fi; fi
    timeout -k 10 300 python3 bench.py $ARGS --no-cpu-baseline > gpurun_out/${TAG}_${v}_$i.log 2>&1 || { echo "run failed"; tail -5 gpurun_out/${TAG}_${v}_$i.log; exit 1; }
    python3 - gpurun_out/${TAG}_${v}_$i.log "$VAR=$v" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d=json.loads(l); print(sys.argv[2], "ms/step", d["ms_per_step"], "min", round(min(d["passes_ms"])/d["steps"],4) if "passes_ms" in d else None, "spread", d.get("spread"))
PY
  done
done
