#!/bin/bash
# round-3 sweep of k_path_w's free parameters on configs[3] (one call: boxes differ by several per cent)
set -u
mkdir -p gpurun_out
run() { label=$1; shift; timeout -k 10 300 python3 bench.py --workload c4 --steps 20 --warmup 5 --no-cpu-baseline "$@" > gpurun_out/r03n_$label.log 2>&1 || { echo "$label failed"; tail -5 gpurun_out/r03n_$label.log; return; }
  python3 - gpurun_out/r03n_$label.log "$label" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d=json.loads(l); print(sys.argv[2], "ms/step", d["ms_per_step"], "min", round(min(d["passes_ms"])/d["steps"],4))
PY
}
run base
run d2 --grid-density 2
run d3 --grid-density 3
run d6 --grid-density 6
run d8 --grid-density 8
run d12 --grid-density 12
run v1 --wide-variant 1
run v2 --wide-variant 2
run v3 --wide-variant 3
run se0 --static-eighths 0
run se8 --static-eighths 8
run base2
