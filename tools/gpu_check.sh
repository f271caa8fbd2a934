#!/bin/bash
# One GPU-box call: the -m gpu suite, smoke(), then a list of bench.py variants ("label:args" ...), each under its
# own timeout; every step is joined with && so that nothing runs after a failure or a hang.  Logs: gpurun_out/<tag>_*.
# usage: tools/gpu_check.sh <tag> [--no-tests] "label:bench args" ...
set -u
TAG=$1; shift
OUT=gpurun_out
mkdir -p $OUT
if [ "${1:-}" = "--no-tests" ]; then shift; else
  timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/${TAG}_tests.log 2>&1 || { echo "GPU TESTS FAILED"; tail -40 $OUT/${TAG}_tests.log; exit 1; }
  tail -3 $OUT/${TAG}_tests.log
  timeout -k 10 300 python __graft_entry__.py smoke > $OUT/${TAG}_smoke.log 2>&1 || { echo "SMOKE FAILED"; tail -20 $OUT/${TAG}_smoke.log; exit 1; }
  tail -1 $OUT/${TAG}_smoke.log
fi
for spec in "$@"; do
  label=${spec%%:*}; args=${spec#*:}
  lib=""
  case "$args" in LIB=*) lib=$(pwd)/project2-pathtracer_amd/build/variants/${args%% *}; lib=${lib/LIB=/}.so; args=${args#* };; esac
  PTMI355_LIB=$lib timeout -k 10 400 python3 bench.py $args > $OUT/${TAG}_bench_$label.log 2>&1 || { echo "bench $label FAILED"; tail -20 $OUT/${TAG}_bench_$label.log; exit 1; }
  python3 - "$OUT/${TAG}_bench_$label.log" "$label" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d=json.loads(l); r=d.get("roofline") or {}; ke=r.get("kernel_events") or {}
        print(sys.argv[2], "Mray/s", d["value"], "ms/step", d["ms_per_step"], "frac", r.get("frac"), "spread", d.get("spread"), "kernel_only_frac", ke.get("kernel_only_frac"), "avg_us", ke.get("avg_launch_us"))
PY
done
