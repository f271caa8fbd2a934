#!/usr/bin/env python3
"""profiles/<tag>_summary.json (tools/summarize_profile.py) -> profiles/traffic_latest.json: HBM bytes and VALU
instructions per rendered step of the bounce launches, which bench.py scales to its own launch count.
usage: tools/traffic_from_profile.py <tag> <steps rendered in the profiled run (timed + warm-up)> [workload=c3] [ordering=2] [direct_light=0]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_source_id  # noqa: E402
tag, steps = sys.argv[1], int(sys.argv[2])
workload = sys.argv[3] if len(sys.argv) > 3 else "c3"
ordering = int(sys.argv[4]) if len(sys.argv) > 4 else 2
direct_light = int(sys.argv[5]) if len(sys.argv) > 5 else 0
d = json.load(open(os.path.join(ROOT, "profiles", tag + "_summary.json")))
tot = {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "SQ_INSTS_VALU": 0.0}
dispatches = 0
for name, o in d["pmc"].items():
    if not (name.startswith("k_bounce") or name.startswith("k_path")):
        continue
    for c in tot:
        if c in o:
            tot[c] += o[c]["mean_per_dispatch"] * o[c]["dispatches"]
    dispatches += o["FETCH_SIZE"]["dispatches"]
path = os.path.join(ROOT, "profiles", "traffic_latest.json")
rec = json.load(open(path)) if os.path.exists(path) else {}
rec[workload] = {
    "hbm_bytes_per_step": round((2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024 / steps),
    "hbm_bytes_per_step_uncorrected": round((tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024 / steps),
    "steps_profiled": steps, "dispatches": dispatches, "profile": tag, "kernel_source_id": kernel_source_id(),
    "ordering": ordering, "direct_light": direct_light,      # bench.py --ordering / --direct-light the profile was taken with
    "valu_wave_instructions_per_step": round(tot["SQ_INSTS_VALU"] / steps),
    "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of the bench.py command in tools/profile.sh (%s, bench "
           "defaults); bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per dispatch (gfx950 FETCH_SIZE counts 64 B per 128-B request on wide "
           "coalesced reads; WRITE_SIZE exact), summed over all k_bounce_* / k_path_q dispatches of the %d rendered steps and divided by %d; "
           "bench.py scales it to its own launch count (bytes per step x steps / launches)" % (tag, steps, steps)}
json.dump(rec, open(path, "w"), indent=1)
print({k: v for k, v in rec[workload].items() if k != "how"})
