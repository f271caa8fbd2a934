#!/bin/bash
# rocprofv3 summaries (kernel trace + PMC passes) of workloads beside configs[2] / configs[3]: tools/r3_prof_mesh.sh [workload args label]...
set -u
OUT=gpurun_out; mkdir -p $OUT
prof() { tag=$1; shift
  bash tools/profile.sh r03z_$tag "$@" > $OUT/r03z_${tag}_profile.log 2>&1; tail -1 $OUT/r03z_${tag}_profile.log
  python3 tools/summarize_profile.py $OUT/prof_r03z_$tag r03z_$tag > $OUT/r03z_${tag}_summarize.log 2>&1; tail -2 $OUT/r03z_${tag}_summarize.log; }
prof mesh --workload mesh
prof mesh5k --workload mesh5k
prof nee --direct-light 1
prof nee_o2 --direct-light 1 --ordering 2
prof c5 --workload c5
mkdir -p $OUT/profiles_out; cp profiles/r03z_mesh* profiles/r03z_nee* profiles/r03z_c5_summary* $OUT/profiles_out/ 2>/dev/null; ls $OUT/profiles_out | grep -c summary
