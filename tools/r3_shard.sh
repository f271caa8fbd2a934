#!/bin/bash
# 8-way shard of configs[2] on one GPU (20 and 200 steps per call): blocks per CU, job size and static share of the whole-path kernel
set -u
mkdir -p gpurun_out
for kv in "" "blocks_per_cu=4" "blocks_per_cu=3" "chunk_rays=64" "chunk_rays=128" "path_static_eighths=0" "path_static_eighths=2" "path_static_eighths=6" "batch=64"; do
  echo "== $kv"
  timeout -k 10 200 python3 tools/shard_sim.py 1 ordering=2 worlds=8 $kv 2>&1 | grep "shards 8"
done
