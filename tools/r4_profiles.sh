#!/bin/bash
# round 4: rocprofv3 profiles (kernel trace + separate PMC passes: tools/profile.sh) of every bench workload on the final kernels.
# Summaries and profiles/traffic_latest.json are made from gpurun_out/prof_<tag>/ afterwards, in the build container:
#   for w in c3 c4 c5 c1k mesh mesh5k; do python tools/summarize_profile.py gpurun_out/prof_r04z_$w r04z_$w; python tools/traffic_from_profile.py r04z_$w 160 $w; done
set -u
for w in "$@"; do
  bash tools/profile.sh r04z_$w --workload $w > gpurun_out/r04z_profile_$w.log 2>&1 || { echo "profile $w failed"; tail -5 gpurun_out/r04z_profile_$w.log; exit 1; }
  tail -2 gpurun_out/r04z_profile_$w.log
done
