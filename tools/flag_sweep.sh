#!/bin/bash
# tools/flag_sweep.sh: bench every library build under project2-pathtracer_amd/build/variants/*.so
# (compiler-flag experiments built on the CPU box) next to the in-tree build; one line per build.
# Build a variant with e.g.
#   cd project2-pathtracer_amd && mkdir -p build/variants && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC \
#     -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -fno-slp-vectorize -w -shared \
#     csrc/pt_*.hip csrc/pt_scene.cpp <extra flags> -o build/variants/<tag>.so
# (build/ is git-ignored but travels to the GPU box with the snapshot); PTMI355_LIB selects the library.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for lib in "" $ROOT/project2-pathtracer_amd/build/variants/*.so; do
  for o in 1 0; do
    PTMI355_LIB=$lib timeout -k 10 200 python3 $ROOT/bench.py --ordering $o --steps 200 --warmup 20 --no-cpu-baseline --no-kernel-events > /tmp/fs.log 2>&1 || { echo "FAILED $lib"; tail -3 /tmp/fs.log; continue; }
    python3 - "$lib" $o <<'PY'
import json,sys
r=json.loads(open('/tmp/fs.log').read().strip().splitlines()[-1])
print("%-14s ordering=%s  %.4f ms/step  %.0f Mray/s" % ((sys.argv[1].split('/')[-1] or 'in-tree'), sys.argv[2], r['ms_per_step'], r['value']))
PY
  done
done
