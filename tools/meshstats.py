"""Traversal statistics of the MESH primitives in the whole-path kernel (k_path_q<MESH>): how full the waves are in the two stages
of a MESH turn (WALK: lane = ray, one BVH node per trip; TRI: lane = one (ray, triangle) pair).  Needs a stats build:
  tools/build_variant.sh meshstats -DPT_CULL_STATS -DPT_MESH_STATS ;  PTMI355_LIB=.../build/variants/meshstats.so python3 tools/meshstats.py [scene]"""
import ctypes as C, importlib, os, sys
sys.path.insert(0, os.getcwd())
pkg = importlib.import_module("project2-pathtracer_amd")
scene = sys.argv[1] if len(sys.argv) > 1 else "scenes/cornell_mesh.txt"
sf = pkg.SceneFile(scene); g, m, cam = sf.flatten(0)
cam.resolution[0], cam.resolution[1] = 1920.0, 1080.0
tr = pkg.PathTracer(pkg.default_config(streams=1, max_depth=8, ordering=2)); tr.set_meshes(sf.meshes()); tr.upload(g, m, cam); tr.set_image(None)
tr.render(1, 20); tr.sync()
st = tr.stats()
out = (C.c_ulonglong * 16)(); pkg.lib().pt_debug_cull_stats.argtypes = [C.POINTER(C.c_ulonglong)]; pkg.lib().pt_debug_cull_stats(out)
s = [int(v) for v in out]
live = sum(int(st.live[k]) for k in range(8))
d = lambda a, b: a / max(b, 1)
print(scene, "live ray-bounces", live)
print("MESH turns per 64 live ray-bounces %.3f at %.1f lanes (rays that enter a mesh stage per live ray-bounce %.3f)" % (s[0] / (live / 64.0), d(s[1], s[0]), d(s[1], live)))
print("WALK: %.1f trips per turn at %.1f lanes (%.1f node visits per ray of a turn)" % (d(s[2], s[0]), d(s[3], s[2]), d(s[3], s[1])))
print("TRI: %.2f groups per turn at %.1f lanes (%.2f triangle tests per ray of a turn)" % (d(s[5], s[0]), d(s[4], s[5]), d(s[4], s[1])))
print("interrupted traversals that went back on the mesh stack: %.3f per ray of a turn" % d(s[6], s[1]))
