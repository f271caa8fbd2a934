"""Stage statistics of the whole-path kernel for 33..256 primitives (k_path_w): groups per stage, their lane fill, and the
phase clock.  Needs a -DPT_CULL_STATS build:  tools/build_variant.sh stats -DPT_CULL_STATS ;
PTMI355_LIB=.../build/variants/stats.so python3 tools/wstats.py [scene] [key=value ...]"""
import ctypes as C, importlib, os, sys
sys.path.insert(0, os.getcwd())
pkg = importlib.import_module("project2-pathtracer_amd")
args = [a for a in sys.argv[1:] if "=" not in a]
kw = {k: int(v) for k, v in (a.split("=") for a in sys.argv[1:] if "=" in a)}
sf = pkg.SceneFile(args[0] if args else "scenes/random256.txt"); g, m, cam = sf.flatten(0)
tr = pkg.PathTracer(pkg.default_config(streams=1, max_depth=8, ordering=2, **kw)); tr.upload(g, m, cam); tr.set_image(None)
tr.render(1, 4); tr.sync()
st = tr.stats()
out = (C.c_ulonglong * 32)(); pkg.lib().pt_debug_wide_stats(out)
s = [int(v) for v in out]
live = sum(int(st.live[k]) for k in range(8))
per = live / 64.0
d = lambda a, b: a / max(b, 1)
print("live ray-bounces", live, "= %.0f groups of 64" % per)
print("FRESH groups per 64 ray-bounces %.3f at %.1f lanes; bound tests of big primitives per ray %.2f; rays not walked %d" % (s[0] / per, d(s[1], s[0]), d(s[25], live), s[24]))
print("WALK trips per FRESH group %.2f at %.1f lanes (cells per ray %.2f); non-empty cells per ray %.2f" % (d(s[2], s[0]), d(s[3], s[2]), d(s[3], live), d(s[4], live)))
print("CELLS chunks per 64 ray-bounces %.3f at %.1f lanes (references read per ray %.2f); new ones per ray %.2f" % (s[5] / per, d(s[6], s[5]), d(s[6], live), d(s[9], live)))
print("BOUNDS cube chunks per 64 ray-bounces %.3f at %.1f lanes; sphere chunks %.3f at %.1f lanes" % (s[10] / per, d(s[11], s[10]), s[12] / per, d(s[13], s[12])))
print("candidates per ray %.2f; overflowed or unwalked rays %d" % (d(s[14] , live) + 0.0, s[15]))
print("TEST cube groups per 64 ray-bounces %.3f at %.1f lanes; sphere groups %.3f at %.1f lanes; exact tests per ray %.2f" % (s[16] / per, d(s[17], s[16]), s[18] / per, d(s[19], s[18]), d(s[17] + s[19], live)))
print("SHADE cube-hit groups per 64 ray-bounces %.3f at %.1f lanes; sphere-hit groups %.3f at %.1f lanes" % (s[26] / per, d(s[27], s[26]), s[28] / per, d(s[29], s[28])))
print("done lanes per TEST group %.1f, shaded %.1f; requeued %.1f (of them after a win %.1f)" % (d(s[21], s[16] + s[18]), d(s[20], s[16] + s[18]), d(s[22], s[16] + s[18]), d(s[23], s[16] + s[18])))
ph = (C.c_ulonglong * 16)(); pkg.lib().pt_debug_phase_cycles(ph)
ph = [int(v) for v in ph][:10]; tot = float(sum(ph)) or 1.0
names = ["schedule", "fresh load + big primitives", "walk", "cells", "bounds", "select", "test load + exact test", "next candidate", "shade", "requeue"]
print("phase clock (share of the waves' cycles): " + ", ".join("%s %.1f%%" % (n, 100.0 * v / tot) for n, v in zip(names, ph)))
