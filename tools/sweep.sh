#!/bin/bash
# usage: tools/sweep.sh "<label>:<bench args>" ...   -> one line per variant
for spec in "$@"; do
  label=${spec%%:*}; args=${spec#*:}
  timeout -k 10 150 python bench.py --steps 100 --warmup 10 --no-cpu-baseline $args > gpurun_out/sweep_$label.log 2>&1
  python - "$label" <<'PY'
import json,sys
lab=sys.argv[1]
for l in open("gpurun_out/sweep_%s.log"%lab):
    if l.startswith("{"):
        d=json.loads(l); r=d.get("roofline") or {}
        print(lab, "Mray/s", d["value"], "ms/step", d["ms_per_step"], "frac", r.get("frac"), "avg_us", (r.get("kernel_events") or {}).get("avg_launch_us"))
        break
else:
    print(lab, "FAILED"); print(open("gpurun_out/sweep_%s.log"%lab).read()[-600:])
PY
done
