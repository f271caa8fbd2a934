#!/bin/bash
# round 4, first call on the generalised k_path_w: the many-primitive parity tests (narrow and wide ids), then in-call A/B
# of configs[3] against the round-3 library (build/variants/base.so) and the 1 024-primitive workload
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_many_primitives.py tests/test_gpu_round3.py tests/test_gpu_parity.py -m gpu -x -q -k "primitives or wide or config4" > $OUT/r04b_tests.log 2>&1 || { echo "TESTS FAILED"; tail -40 $OUT/r04b_tests.log; exit 1; }
tail -3 $OUT/r04b_tests.log
bash tools/ab_lib.sh r04b_c4 2 "--workload c4 --steps 20 --warmup 5" base - || exit 1
PTMI355_LIB= timeout -k 10 400 python3 bench.py --workload c1k --steps 20 --warmup 5 --no-cpu-baseline > $OUT/r04b_c1k.log 2>&1 || { echo "c1k failed"; tail -20 $OUT/r04b_c1k.log; exit 1; }
python3 - <<'PY'
import json
for l in open("gpurun_out/r04b_c1k.log"):
    if l.startswith("{"):
        d=json.loads(l); print("c1k ms/step", d["ms_per_step"], "live", d["config"]["live_ray_bounces_per_step"], d["config"]["kernel_shape"])
PY
PTMI355_LIB= timeout -k 10 400 python3 bench.py --workload c1k --steps 20 --warmup 5 --no-cpu-baseline --ordering 0 > $OUT/r04b_c1k_o0.log 2>&1 || { echo "c1k o0 failed"; tail -20 $OUT/r04b_c1k_o0.log; exit 1; }
grep -o '"ms_per_step": [0-9.]*' $OUT/r04b_c1k_o0.log
