#!/bin/bash
# round-3 final GPU call: the whole -m gpu suite, smoke, the bench lines of every workload, rocprofv3 profiles of configs[2] and configs[3]
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $OUT/r03z_gpu_tests.log 2>&1 || { echo "GPU TESTS FAILED"; tail -60 $OUT/r03z_gpu_tests.log; exit 1; }
tail -1 $OUT/r03z_gpu_tests.log
timeout -k 10 300 python __graft_entry__.py smoke > $OUT/r03z_smoke.log 2>&1 || { echo "SMOKE FAILED"; tail -20 $OUT/r03z_smoke.log; exit 1; }
tail -1 $OUT/r03z_smoke.log
bash tools/profile.sh r03z > $OUT/r03z_profile.log 2>&1; tail -2 $OUT/r03z_profile.log
python3 tools/summarize_profile.py $OUT/prof_r03z r03z > $OUT/r03z_summarize.log 2>&1; tail -2 $OUT/r03z_summarize.log
python3 tools/traffic_from_profile.py r03z 160 c3 | tail -1
bash tools/profile.sh r03z_c4 --workload c4 > $OUT/r03z_c4_profile.log 2>&1; tail -2 $OUT/r03z_c4_profile.log
python3 tools/summarize_profile.py $OUT/prof_r03z_c4 r03z_c4 > $OUT/r03z_c4_summarize.log 2>&1; tail -2 $OUT/r03z_c4_summarize.log
python3 tools/traffic_from_profile.py r03z_c4 160 c4 | tail -1
mkdir -p $OUT/profiles_out; cp profiles/r03z* profiles/traffic_latest.json $OUT/profiles_out/ 2>/dev/null
run() { label=$1; shift; timeout -k 10 400 python3 bench.py "$@" > $OUT/r03z_bench_$label.log 2>&1 || { echo "bench $label failed"; tail -20 $OUT/r03z_bench_$label.log; exit 1; }; }
run driver --gpus 1 --steps 20 --warmup 5
run default
run c2 --workload c2 --steps 20 --warmup 5 --no-cpu-baseline
run c4 --workload c4 --steps 20 --warmup 5 --no-cpu-baseline
run c4_192 --workload c4 --no-cpu-baseline
run c5 --workload c5 --steps 20 --warmup 5 --no-cpu-baseline
run mesh --workload mesh --steps 20 --warmup 5 --no-cpu-baseline
run mesh5k --workload mesh5k --steps 20 --warmup 5 --no-cpu-baseline
run nee --steps 20 --warmup 5 --no-cpu-baseline --direct-light 1
run nee_o2 --steps 20 --warmup 5 --no-cpu-baseline --direct-light 1 --ordering 2
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03z_bench_*.log")):
    for l in open(f):
        if l.startswith("{"):
            d=json.loads(l); r=d["roofline"]
            print(f.split("/")[-1], "ms/step", d["ms_per_step"], "Mray/s", d["value"], "frac", r["frac"], "spread", d["spread"], "hbm", (r.get("hbm_measured") or {}).get("frac_of_peak"), "valu", (r.get("valu_issue") or {}).get("frac"))
PY
timeout -k 10 300 python3 tools/shard_sim.py 1 ordering=2 > $OUT/r03z_shard_sim_c3.log 2>&1; tail -4 $OUT/r03z_shard_sim_c3.log
timeout -k 10 200 python3 tools/launch_curve.py 8 > $OUT/r03z_launch_curve_8way.log 2>&1; tail -3 $OUT/r03z_launch_curve_8way.log
timeout -k 10 200 python3 tools/launch_curve.py 1 > $OUT/r03z_launch_curve_1way.log 2>&1; tail -3 $OUT/r03z_launch_curve_1way.log
