import ctypes as C, importlib, os, sys, shutil
sys.path.insert(0, os.getcwd())
shutil.copy("project2-pathtracer_amd/libptmi355_stats.so", "project2-pathtracer_amd/libptmi355.so")
pkg = importlib.import_module("project2-pathtracer_amd")
sf = pkg.SceneFile("scenes/cornell_mirror.txt"); g, m, cam = sf.flatten(0)
tr = pkg.PathTracer(pkg.default_config(max_depth=8, batch=1)); tr.upload(g, m, cam); tr.set_image(None)
tr.render(1, 2); tr.sync()
out = (C.c_ulonglong * 8)(); pkg.lib().pt_debug_cull_stats(out)
g0, bi, ba, si, sa, cand = out[0], out[1], out[2], out[3], out[4], out[5]
print("wave groups", g0, "box iters/group %.2f" % (bi / g0), "box active lanes/iter %.1f" % (ba / max(bi, 1)),
      "sphere iters/group %.2f" % (si / g0), "sphere active lanes/iter %.1f" % (sa / max(si, 1)), "candidates/ray %.2f" % (cand / (g0 * 64.0)))
