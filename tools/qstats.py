"""Stage statistics of the typed work-queue kernel (diagnostic).  Needs a library built with -DPT_CULL_STATS:
    tools/build_variant.sh stats -DPT_CULL_STATS
    PTMI355_LIB=$PWD/project2-pathtracer_amd/build/variants/stats.so python3 tools/qstats.py [scene] [depth] [ordering: 1 per bounce, 2 whole paths]
(ordering 2: "retests" = camera-ray groups, "candidates left" = shaded lanes at the last level)"""
import ctypes as C, importlib, os, sys
sys.path.insert(0, os.getcwd())
pkg = importlib.import_module("project2-pathtracer_amd")
sf = pkg.SceneFile(sys.argv[1] if len(sys.argv) > 1 else "scenes/cornell_mirror.txt"); g, m, cam = sf.flatten(0)
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ordering = int(sys.argv[3]) if len(sys.argv) > 3 else 1
tr = pkg.PathTracer(pkg.default_config(streams=1, max_depth=depth, ordering=ordering)); tr.upload(g, m, cam); tr.set_image(None)
tr.render(1, 16); tr.sync()
st = tr.stats()
live = sum(st.live[k] for k in range(depth))
out = (C.c_ulonglong * 16)(); pkg.lib().pt_debug_cull_stats.argtypes = [C.POINTER(C.c_ulonglong)]; pkg.lib().pt_debug_cull_stats(out)
o = list(out)
print("live ray-bounces", live, "= %.0f full groups" % (live / 64))
print("fresh groups %d (valid lanes/group %.1f), candidates/ray %.3f" % (o[8], o[9] / max(1, o[8]), o[5] / max(1, o[9])))
print("box groups %d (lanes/group %.1f)  sphere groups %d (lanes/group %.1f)" % (o[10], o[11] / max(1, o[10]), o[12], o[13] / max(1, o[12])))
print("tests/ray %.3f  test groups per fresh group %.3f" % ((o[11] + o[13]) / max(1, o[9]), (o[10] + o[12]) / max(1, o[8])))
print("shaded lanes %d (%.1f per test group)  re-queued %d  of which retests %d  lanes with candidates left after a test %d" % (o[14], o[14] / max(1, o[10] + o[12]), o[15], o[6], o[7]))
