#!/bin/bash
# round 4, final measurement set on the final tree: the -m gpu suite + smoke, the bench lines of every workload (the driver's
# command first), then the rocprofv3 profiles of every workload (tools/r4_profiles.sh) -- two gpurun calls: `final` and `profiles`
set -u
OUT=gpurun_out; mkdir -p $OUT
if [ "${1:-final}" = "profiles" ]; then
  bash tools/r4_profiles.sh c3 c4 c5 c1k mesh mesh5k || exit 1
  exit 0
fi
bash tools/gpu_check.sh r04z \
  "driver:--gpus 1 --steps 20 --warmup 5" \
  "c3_192:" \
  "c2:--workload c2 --no-cpu-baseline" \
  "c4:--workload c4 --steps 20 --warmup 5 --no-cpu-baseline" \
  "c4_192:--workload c4 --no-cpu-baseline" \
  "c1k:--workload c1k --steps 20 --warmup 5 --no-cpu-baseline" \
  "c5:--workload c5 --steps 20 --warmup 5 --no-cpu-baseline" \
  "mesh:--workload mesh --no-cpu-baseline" \
  "mesh5k:--workload mesh5k --no-cpu-baseline" \
  "nee:--direct-light 1 --no-cpu-baseline" \
  "nee_o2:--direct-light 1 --ordering 2 --no-cpu-baseline" || exit 1
