"""Per-rank cost of the row-sharded render as a function of the shard count, measured on ONE GPU:
the context renders only the rows one rank of an N-GPU job would own (row_stride = N).  N x the per-step
time against the 1-GPU step time OF THE SAME RUN is the compute-side strong-scaling efficiency (exchange
excluded).  usage: python tools/shard_sim.py   (from the repo root, on the GPU box)"""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("project2-pathtracer_amd")
sf = pkg.SceneFile(os.path.join(ROOT, "scenes", "cornell_mirror.txt"))
g, m, cam = sf.flatten(0)
base = {}
for stride in (1, 2, 4, 8):
    tr = pkg.PathTracer(pkg.default_config(max_depth=8, ordering=1, row_offset=0, row_stride=stride))
    tr.upload(g, m, cam)
    tr.set_image(None)
    tr.render(1, 40)
    tr.sync()
    for steps in (20, 200):
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter()
            tr.render(41 + rep * steps, steps)
            tr.sync()
            best = min(best, (time.perf_counter() - t0) / steps * 1e3)
        if stride == 1:
            base[steps] = best
        print("shards %d, %3d steps per call: %.4f ms/step per rank, x%d = %.4f ms, efficiency %.2f"
              % (stride, steps, best, stride, best * stride, base[steps] / (best * stride)))
    tr.close()
