"""Exact-pass statistics of the culled nearest-hit (diagnostic): wave-level iterations of the per-lane
cube / sphere loops and their active lanes.  Needs a library built with -DPT_CULL_STATS:
  cd project2-pathtracer_amd && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off \
     (now: tools/build_variant.sh stats -DPT_CULL_STATS)
(select it with PTMI355_LIB=.../build/variants/stats.so).
Round-1 result on configs[2]: cube loop 1.43 iterations/group at 31 active lanes, sphere loop 0.96 at 6.1,
0.78 candidates per ray."""
import ctypes as C, importlib, os, sys, shutil
sys.path.insert(0, os.getcwd())
# select the -DPT_CULL_STATS build with PTMI355_LIB (tools/build_variant.sh stats -DPT_CULL_STATS)
pkg = importlib.import_module("project2-pathtracer_amd")
sf = pkg.SceneFile(sys.argv[1] if len(sys.argv) > 1 else "scenes/cornell_mirror.txt"); g, m, cam = sf.flatten(0)
tr = pkg.PathTracer(pkg.default_config(streams=1, max_depth=8, batch=1)); tr.upload(g, m, cam); tr.set_image(None)
tr.render(1, 2); tr.sync()
out = (C.c_ulonglong * 16)(); pkg.lib().pt_debug_cull_stats(out)
g0, bi, ba, si, sa, cand = out[0], out[1], out[2], out[3], out[4], out[5]
print("cluster-walk iterations/group %.2f at %.1f active lanes" % (out[6] / max(g0, 1), out[7] / max(out[6], 1)))
print("wave groups", g0, "box iters/group %.2f" % (bi / g0), "box active lanes/iter %.1f" % (ba / max(bi, 1)),
      "sphere iters/group %.2f" % (si / g0), "sphere active lanes/iter %.1f" % (sa / max(si, 1)), "candidates/ray %.2f" % (cand / (g0 * 64.0)))
