"""Per-rank cost of the row-sharded render as a function of the shard count, measured on ONE GPU:
the context renders only the rows one rank of an N-GPU job would own (row_stride = N).  N x the per-step
time against the 1-GPU step time is the compute-side strong-scaling efficiency (exchange excluded)."""
import importlib, sys, time, os
sys.path.insert(0, os.getcwd())
import torch
pkg = importlib.import_module("project2-pathtracer_amd")
sf = pkg.SceneFile("scenes/cornell_mirror.txt")
g, m, cam = sf.flatten(0)
for stride in (1, 2, 4, 8):
    for batch in (0, 32):
        tr = pkg.PathTracer(pkg.default_config(max_depth=8, ordering=1, row_offset=0, row_stride=stride, batch=batch))
        tr.upload(g, m, cam); tr.set_image(None)
        tr.render(1, 40); tr.sync()
        for steps in (20, 200):
            t0 = time.perf_counter(); tr.render(41, steps); tr.sync(); dt = time.perf_counter() - t0
            print("row_stride %d batch %s steps %3d: %.4f ms/step  (x%d = %.4f; efficiency vs 0.2932: %.2f)" % (stride, batch or "auto", steps, dt / steps * 1e3, stride, dt / steps * 1e3 * stride, 0.2932 / (dt / steps * 1e3 * stride)))
        tr.close()
