#!/bin/bash
# round 4: camera groups on one cone (PT_FAN) -- parity, then in-call A/B against the same library built with -DPT_FAN=0
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_many_primitives.py tests/test_gpu_round3.py tests/test_gpu_parity.py -m gpu -x -q -k "primitives or wide or config4 or camera_groups" > $OUT/r04c_tests.log 2>&1 || { echo "TESTS FAILED"; tail -40 $OUT/r04c_tests.log; exit 1; }
tail -3 $OUT/r04c_tests.log
bash tools/ab_lib.sh r04c_c4 3 "--workload c4 --steps 20 --warmup 5" nofan - || exit 1
bash tools/ab_lib.sh r04c_c1k 2 "--workload c1k --steps 20 --warmup 5" nofan - || exit 1
bash tools/ab_lib.sh r04c_c4_192 1 "--workload c4" nofan - || exit 1
