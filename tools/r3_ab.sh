#!/bin/bash
# in-call A/B: k_fold clears only the plane entries that were written (working tree) against clearing all (prev = the last commit)
set -u
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_round2.py tests/test_gpu_parity.py -x -q -k "batch or fold or headline or lazy" > gpurun_out/r03f2_tests.log 2>&1 || { echo "TESTS FAILED"; tail -40 gpurun_out/r03f2_tests.log; exit 1; }
tail -1 gpurun_out/r03f2_tests.log
echo "== driver command"; bash tools/ab_lib.sh r03f2_c3 3 "--steps 20 --warmup 5" prev - || exit 1
echo "== configs[3]"; bash tools/ab_lib.sh r03f2_c4 2 "--workload c4 --steps 20 --warmup 5" prev - || exit 1
for v in prev -; do
  if [ "$v" = "-" ]; then unset PTMI355_LIB; else export PTMI355_LIB=$(pwd)/project2-pathtracer_amd/build/variants/$v.so; fi
  echo "== 8-way shard, $v"; timeout -k 10 200 python3 tools/shard_sim.py 1 ordering=2 worlds=8 2>&1 | grep "shards 8"
done
