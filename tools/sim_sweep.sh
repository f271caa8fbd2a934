set -e
for s in 1 2; do for b in 0 2 3 4; do
  echo "== streams $s blocks_per_cu $b"
  if [ $b = 0 ]; then timeout -k 10 120 python tools/shard_sim.py $s worlds=8; else timeout -k 10 120 python tools/shard_sim.py $s worlds=8 blocks_per_cu=$b; fi
done; done
