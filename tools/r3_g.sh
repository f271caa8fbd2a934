#!/bin/bash
# round-3 GPU call: k_path_w (grid walk, inner walk loop) -- quick parity, bench, counters, stage statistics
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_round2.py tests/test_gpu_parity.py -x -q -k "wide or many_primitives or config4 or overflows or scene_scales" > $OUT/r03i_wide_tests.log 2>&1 || { echo "WIDE TESTS FAILED"; tail -60 $OUT/r03i_wide_tests.log; exit 1; }
tail -2 $OUT/r03i_wide_tests.log
run() { label=$1; shift; timeout -k 10 400 python3 bench.py "$@" > $OUT/r03i_bench_$label.log 2>&1 || { echo "bench $label failed"; tail -20 $OUT/r03i_bench_$label.log; exit 1; }; }
run c4 --workload c4 --steps 20 --warmup 5 --no-cpu-baseline
run c4_d2 --workload c4 --steps 20 --warmup 5 --no-cpu-baseline --grid-density 2
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03i_bench_*.log")):
    for l in open(f):
        if l.startswith("{"):
            d=json.loads(l); print(f.split("/")[-1], "ms/step", d["ms_per_step"], "Mray/s", d["value"], "frac", d["roofline"]["frac"], "spread", d["spread"], "live/step", d["config"]["live_ray_bounces_per_step"])
PY
PTMI355_LIB=$(pwd)/project2-pathtracer_amd/build/variants/stats.so timeout -k 10 300 python3 tools/wstats.py scenes/random256.txt > $OUT/r03i_wstats.log 2>&1; cat $OUT/r03i_wstats.log
bash tools/pmc_bench.sh r03i_a "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" --workload c4 2>&1 | tail -2
