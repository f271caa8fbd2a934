set -u
OUT=gpurun_out; mkdir -p $OUT
for w in mesh mesh5k; do
  bash tools/profile.sh r03z_$w --workload $w > $OUT/r03z_${w}_profile.log 2>&1; tail -1 $OUT/r03z_${w}_profile.log
  python3 tools/summarize_profile.py $OUT/prof_r03z_$w r03z_$w > $OUT/r03z_${w}_summarize.log 2>&1; tail -2 $OUT/r03z_${w}_summarize.log
done
mkdir -p $OUT/profiles_out; cp profiles/r03z_mesh* $OUT/profiles_out/ 2>/dev/null; ls $OUT/profiles_out | grep mesh
