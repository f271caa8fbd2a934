#!/bin/bash
# round-3 GPU call #3: k_path_w after the overflow fix, mesh / direct-light bench lines, shard simulations, c3 counters
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_round3.py tests/test_gpu_round2.py -x -q -k "wide or overflows" > $OUT/r03c_wide_tests.log 2>&1 || { echo "WIDE TESTS FAILED"; tail -60 $OUT/r03c_wide_tests.log; exit 1; }
tail -2 $OUT/r03c_wide_tests.log
run() { label=$1; shift; timeout -k 10 400 python3 bench.py "$@" > $OUT/r03c_bench_$label.log 2>&1 || { echo "bench $label failed"; tail -20 $OUT/r03c_bench_$label.log; exit 1; }; }
run c4 --workload c4 --steps 20 --warmup 5 --no-cpu-baseline
run c4_s2 --workload c4 --steps 20 --warmup 5 --no-cpu-baseline --streams 2
run c4_192 --workload c4 --no-cpu-baseline
run c4_cs6 --workload c4 --steps 20 --warmup 5 --no-cpu-baseline --cluster-size 6
run c4_cs4 --workload c4 --steps 20 --warmup 5 --no-cpu-baseline --cluster-size 4
run mesh --workload mesh --steps 20 --warmup 5 --no-cpu-baseline
run mesh_o0 --workload mesh --steps 20 --warmup 5 --no-cpu-baseline --ordering 0
run mesh5k --workload mesh5k --steps 20 --warmup 5 --no-cpu-baseline
run mesh5k_o0 --workload mesh5k --steps 20 --warmup 5 --no-cpu-baseline --ordering 0
run nee --steps 20 --warmup 5 --no-cpu-baseline --direct-light 1
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03c_bench_*.log")):
    for l in open(f):
        if l.startswith("{"):
            d=json.loads(l); print(f.split("/")[-1], "ms/step", d["ms_per_step"], "Mray/s", d["value"], "frac", d["roofline"]["frac"], "spread", d["spread"], "live/step", d["config"]["live_ray_bounces_per_step"])
PY
bash tools/build_variant.sh stats -DPT_CULL_STATS > $OUT/r03c_build.log 2>&1 || { tail -20 $OUT/r03c_build.log; exit 1; }
PTMI355_LIB=$(pwd)/project2-pathtracer_amd/build/variants/stats.so timeout -k 10 300 python3 tools/wstats.py scenes/random256.txt > $OUT/r03c_wstats.log 2>&1; cat $OUT/r03c_wstats.log
timeout -k 10 300 python3 tools/shard_sim.py 1 ordering=2 > $OUT/r03c_shard_sim_c3.log 2>&1; cat $OUT/r03c_shard_sim_c3.log
timeout -k 10 400 python3 tools/shard_sim.py 1 scene=scenes/random256.txt ordering=2 > $OUT/r03c_shard_sim_c4.log 2>&1; cat $OUT/r03c_shard_sim_c4.log
cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_r03c_shard8 -- python3 $GRAFT_REPO_ROOT/tools/shard_trace.py world=8 streams=1 steps=20 passes=6 ordering=2 > $GRAFT_REPO_ROOT/$OUT/r03c_shard8_trace.log 2>&1; cd $GRAFT_REPO_ROOT
python3 tools/trace_timeline.py $OUT/prof_r03c_shard8 > $OUT/r03c_shard8_timeline.log 2>&1; tail -12 $OUT/r03c_shard8_timeline.log
bash tools/profile.sh r03c > $OUT/r03c_profile.log 2>&1; tail -3 $OUT/r03c_profile.log
