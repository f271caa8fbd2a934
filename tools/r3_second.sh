#!/bin/bash
# round-3 GPU call #2: stage statistics + rocprof counters of k_path_w on configs[3]
set -u
OUT=gpurun_out; mkdir -p $OUT
bash tools/build_variant.sh stats -DPT_CULL_STATS > $OUT/r03b_build.log 2>&1 || { tail -20 $OUT/r03b_build.log; exit 1; }
for cs in 0 5; do PTMI355_LIB=$(pwd)/project2-pathtracer_amd/build/variants/stats.so timeout -k 10 300 python3 tools/wstats.py scenes/random256.txt cluster_size=$cs > $OUT/r03b_wstats_cs$cs.log 2>&1 || { tail -20 $OUT/r03b_wstats_cs$cs.log; exit 1; }; cat $OUT/r03b_wstats_cs$cs.log; done
bash tools/profile.sh r03b_c4 --workload c4 > $OUT/r03b_profile.log 2>&1; tail -5 $OUT/r03b_profile.log
python3 tools/summarize_profile.py gpurun_out/prof_r03b_c4 r03b_c4 > $OUT/r03b_summary.log 2>&1; tail -30 $OUT/r03b_summary.log
