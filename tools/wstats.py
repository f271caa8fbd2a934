"""Stage statistics of the whole-path kernel for 33..256 primitives (k_path_w): groups per stage and their lane fill.
Needs a -DPT_CULL_STATS build:  tools/build_variant.sh stats -DPT_CULL_STATS ;  PTMI355_LIB=.../build/variants/stats.so python3 tools/wstats.py [scene] [key=value ...]"""
import ctypes as C, importlib, os, sys
sys.path.insert(0, os.getcwd())
pkg = importlib.import_module("project2-pathtracer_amd")
args = [a for a in sys.argv[1:] if "=" not in a]
kw = {k: int(v) for k, v in (a.split("=") for a in sys.argv[1:] if "=" in a)}
sf = pkg.SceneFile(args[0] if args else "scenes/random256.txt"); g, m, cam = sf.flatten(0)
tr = pkg.PathTracer(pkg.default_config(max_depth=8, ordering=2, **kw)); tr.upload(g, m, cam); tr.set_image(None)
tr.render(1, 4); tr.sync()
st = tr.stats()
out = (C.c_ulonglong * 16)(); pkg.lib().pt_debug_cull_stats(out)
s = [int(v) for v in out]
live = sum(int(st.live[k]) for k in range(8))
per = live / 64.0
print("live ray-bounces", live, "= %.0f groups of 64" % per)
print("FRESH groups per 64 ray-bounces %.3f at %.1f lanes" % (s[0] / per, s[1] / max(s[0], 1)))
print("PAIR chunks per 64 ray-bounces %.3f at %.1f lanes; member tests per ray %.2f; member-loop trips per chunk (max over lanes) %.2f" % (s[2] / per, s[3] / max(s[2], 1), s[13] / max(live, 1), s[15] / max(s[2], 1)))
print("candidates per ray %.2f; overflow rays %d" % (s[10] / max(live, 1), s[11]))
print("TEST cube groups per 64 ray-bounces %.3f at %.1f lanes; sphere groups %.3f at %.1f lanes; exact tests per ray %.2f" % (s[4] / per, s[5] / max(s[4], 1), s[6] / per, s[7] / max(s[6], 1), (s[5] + s[7]) / max(live, 1)))
print("done lanes per TEST group %.1f, shaded %.1f; requeued %.1f (of them after a win %.1f)" % (s[9] / max(s[4] + s[6], 1), s[8] / max(s[4] + s[6], 1), s[12] / max(s[4] + s[6], 1), s[14] / max(s[4] + s[6], 1)))
