#!/bin/bash
# round-3 GPU call #1: the new kernel's tests first (fail fast), then the whole suite, smoke, bench lines, VALU ceiling
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_round3.py -x -q -k "wide or turn" > $OUT/r03a_wide_tests.log 2>&1 || { echo "WIDE TESTS FAILED"; tail -60 $OUT/r03a_wide_tests.log; exit 1; }
tail -2 $OUT/r03a_wide_tests.log
timeout -k 10 400 python3 bench.py --workload c4 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/r03a_bench_c4_pathw.log 2>&1 || { echo "c4 bench failed"; tail -20 $OUT/r03a_bench_c4_pathw.log; exit 1; }
timeout -k 10 400 python3 bench.py --workload c4 --steps 20 --warmup 5 --no-cpu-baseline --ordering 0 > $OUT/r03a_bench_c4_stable.log 2>&1 || { echo "c4 stable bench failed"; exit 1; }
for v in 1 2; do timeout -k 10 400 python3 bench.py --workload c4 --steps 20 --warmup 5 --no-cpu-baseline --wide-variant $v > $OUT/r03a_bench_c4_pathw_v$v.log 2>&1 || { echo "c4 v$v failed"; tail -5 $OUT/r03a_bench_c4_pathw_v$v.log; exit 1; }; done
for cs in 4 6 12; do timeout -k 10 400 python3 bench.py --workload c4 --steps 20 --warmup 5 --no-cpu-baseline --cluster-size $cs > $OUT/r03a_bench_c4_pathw_cs$cs.log 2>&1 || { echo "c4 cs$cs failed"; exit 1; }; done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03a_bench_c4*.log")):
    for l in open(f):
        if l.startswith("{"):
            d=json.loads(l); print(f.split("/")[-1], "ms/step", d["ms_per_step"], "frac", d["roofline"]["frac"], "spread", d["spread"])
PY
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/r03a_tests.log 2>&1 || { echo "GPU TESTS FAILED"; tail -60 $OUT/r03a_tests.log; exit 1; }
tail -3 $OUT/r03a_tests.log
timeout -k 10 300 python __graft_entry__.py smoke > $OUT/r03a_smoke.log 2>&1 || { echo "SMOKE FAILED"; tail -20 $OUT/r03a_smoke.log; exit 1; }
tail -1 $OUT/r03a_smoke.log
timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r03a_bench_driver.log 2>&1 || { echo "driver bench failed"; tail -20 $OUT/r03a_bench_driver.log; exit 1; }
tail -1 $OUT/r03a_bench_driver.log | cut -c1-600
cd tools/ubench && hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I../../project2-pathtracer_amd/csrc valu_ceiling.hip -o /tmp/valu_ceiling && timeout -k 10 120 /tmp/valu_ceiling > ../../$OUT/r03a_valu_ceiling.log 2>&1; tail -4 ../../$OUT/r03a_valu_ceiling.log
