"""One rank of an N-way row-sharded job on this GPU (as tools/shard_sim.py), a few K-step passes: run under
   rocprofv3 --kernel-trace to see the launch timeline of a small shard.
usage: python tools/shard_trace.py [world=8] [streams=2] [steps=20] [passes=6] [key=value ...]"""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("project2-pathtracer_amd")
KW = dict(a.split("=") for a in sys.argv[1:])
world, S, steps, passes = (int(KW.pop(k, d)) for k, d in (("world", 8), ("streams", 2), ("steps", 20), ("passes", 6)))
SCENE = KW.pop("scene", "scenes/cornell_mirror.txt")
KW = {k: int(v) for k, v in KW.items()}
sf = pkg.SceneFile(os.path.join(ROOT, SCENE))
g, m, cam = sf.flatten(0)
tr = pkg.PathTracer(pkg.default_config(**dict(dict(max_depth=8, ordering=1, row_offset=0, row_stride=world, streams=S), **KW)))
tr.upload(g, m, cam)
tr.set_image(None)
tr.render(1, 5)
tr.sync()
for p in range(passes):
    t0 = time.perf_counter()
    tr.render(6, steps)
    tr.sync()
    print("pass %d: %.4f ms (%.4f ms/step per rank)" % (p, (time.perf_counter() - t0) * 1e3, (time.perf_counter() - t0) * 1e3 / steps), flush=True)
    time.sleep(0.002)
tr.close()
